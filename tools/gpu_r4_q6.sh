# Round 4 experiment: Taylor-Green (no viscosity) instantiation of the row-form update at four wavefronts per SIMD
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r4_q6; rm -rf $O; mkdir -p $O
(timeout 600 python -m pytest tests/test_gpu_configs.py tests/test_gpu_kernels.py -x -q -k "config3 or qupdate or hydro_mult" 2>&1 | tail -3) > $O/tests.log 2>&1
timeout 300 python bench.py --no-cpu-baseline --legs tg --steps 10 --warmup 3 2>/dev/null | grep '^{' > $O/bench.json
python - <<'PY' > $O/summary.txt 2>&1
import json
d = json.loads(open('gpurun_out/r4_q6/bench.json').read())
q = [v for k, v in d['kernels'].items() if k.startswith(('qpoint', 'qrows'))][0]
tq = [v for k, v in d['legs']['tg']['kernels'].items() if k.startswith(('qpoint', 'qrows'))][0]
print('c2 %.3f ms/step value %.1f qupdate %.1f us | tg %.2f ms/step value %.1f qupdate %.1f us' % (d['ms_per_step'], d['value'], q['mean_us'], d['legs']['tg']['ms_per_step'], d['legs']['tg']['value'], tq['mean_us']))
PY
cat $O/tests.log $O/summary.txt
