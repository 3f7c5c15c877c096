cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r4_stress; rm -rf $O; mkdir -p $O
(timeout 2200 python -m pytest tests -q -m gpu -x 2>&1 | tail -8) > $O/tests.log 2>&1
for K in 0 1; do
  LGH_STORE_STRESS=$K timeout 400 python bench.py --no-cpu-baseline --legs tg,c3 --steps 20 --warmup 5 2>/dev/null | grep '^{' > $O/bench_store$K.json
done
python - <<'PY' > $O/summary.txt 2>&1
import json
for f in (0, 1):
    d = json.loads(open('gpurun_out/r4_stress/bench_store%d.json' % f).read())
    q = [v for k, v in d['kernels'].items() if k.startswith(('qpoint', 'qrows'))][0]
    tg = d['legs']['tg']; tq = [v for k, v in tg['kernels'].items() if k.startswith(('qpoint', 'qrows'))][0]
    c3 = d['legs']['c3']; cq = [v for k, v in c3['kernels'].items() if k.startswith(('qpoint', 'qrows'))][0]
    print('store %d: c2 %.3f ms/step value %.1f qupdate %.1f us | tg %.2f ms/step value %.1f qupdate %.1f us | c3 %.2f value %.1f qupdate %.1f | e %.12e' % (
        f, d['ms_per_step'], d['value'], q['mean_us'], tg['ms_per_step'], tg['value'], tq['mean_us'], c3['ms_per_step'], c3['value'], cq['mean_us'], d['config']['e_norm']))
PY
cat $O/tests.log $O/summary.txt
