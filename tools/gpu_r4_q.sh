# Round 4: the row form of the quadrature update (lgh_qrows.hpp) against the point form: the parity tests that exercise
# it, then the per-kernel timing of bench.py for both forms on one box (C2 headline + the 64^3 Taylor-Green leg).
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r4_q; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "qupdate or fused or hydro_mult or kernel_switches or problem7 or energies" 2>&1 | tail -15) > $O/tests.log 2>&1
for F in 1 0; do
  LGH_Q_FORM=$F timeout 200 python bench.py --no-cpu-baseline --legs tg --steps 10 --warmup 3 2>/dev/null | grep '^{' > $O/bench_form$F.json
done
python - <<'PY' > $O/summary.txt 2>&1
import json
for f in (1, 0):
    d = json.loads(open('gpurun_out/r4_q/bench_form%d.json' % f).read())
    q = [v for k, v in d['kernels'].items() if k.startswith('qpoint')][0]
    tq = [v for k, v in d['legs']['tg']['kernels'].items() if k.startswith('qpoint')][0]
    print('form', f, 'c2 ms/step', d['ms_per_step'], 'value', d['value'], 'qupdate us', q['mean_us'], '| tg ms/step', d['legs']['tg']['ms_per_step'], 'value', d['legs']['tg']['value'], 'qupdate us', tq['mean_us'])
PY
cat $O/tests.log $O/summary.txt
