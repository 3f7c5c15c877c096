mkdir -p gpurun_out/var
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for cfg in "2 1" "2 0" "1 0"; do
set -- $cfg
LGH_VCG_VARIANT=$1 LGH_VCG_DYN=$2 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/var/bench_c.json 2> gpurun_out/var/bench_c.err
python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/var/bench_c.json') if l.startswith('{')][-1])
print("variant/dyn=$cfg", round(d['value'],1), round(d['ms_per_step'],2), [ (k[:12], round(v['mean_us'],1)) for k,v in d['kernels'].items()][:2])
PY
done
mkdir -p gpurun_out/trace
LGH_VCG_TRACE=gpurun_out/trace/k1.txt timeout 300 ./laghos_amd/laghos -p 1 -m data/cube01_hex.mesh -rs 4 -ok 3 -ot 2 -ms 3 -pa > gpurun_out/trace/log.txt 2>&1
