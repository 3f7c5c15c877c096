mkdir -p gpurun_out/var
for g in 1 1.25 1.5 2 3 5; do
LGH_VCG_GRIDX=$g timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/var/bench_c.json 2> gpurun_out/var/bench_c.err
python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/var/bench_c.json') if l.startswith('{')][-1])
print("gridx=$g", round(d['value'],1), round(d['ms_per_step'],2), [ (k[:12], round(v['mean_us'],1)) for k,v in d['kernels'].items()][:2])
PY
done
