mkdir -p gpurun_out/var
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/var/bench.json 2> gpurun_out/var/bench.err
python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/var/bench.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['fom']['seconds'])
for k,v in d['kernels'].items(): print('   ',k, round(v['mean_us'],1), v['launches'], round(v['GBs']))
PY
