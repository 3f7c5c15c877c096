cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out/r2g
APP="./laghos_amd/laghos -p 1 -m data/cube01_hex.mesh -rs 4 -ok 3 -ot 2 -ms 3 -pa"
for nw in 5 -1; do
  tag=nw$nw
  LGH_PCG_CLOCK=1 LGH_PCG_NODE_WEIGHT=$nw LGH_PCG_TRACE=gpurun_out/r2g/trace_$tag.txt timeout 120 $APP > gpurun_out/r2g/app_$tag.log 2>&1; echo "$tag rc=$?"
  python tools/pcg_trace_summary.py gpurun_out/r2g/trace_$tag.txt 10 > gpurun_out/r2g/summary_$tag.txt 2>&1
  tail -5 gpurun_out/r2g/summary_$tag.txt | cut -c1-420
  grep "|e|" gpurun_out/r2g/app_$tag.log | tail -1; grep "shader clock" gpurun_out/r2g/app_$tag.log | tail -2
done
LGH_PCG=0 LGH_VCG_TRACE=gpurun_out/r2g/vcg_trace.txt timeout 120 $APP > gpurun_out/r2g/app_vcg.log 2>&1; echo "vcg rc=$?"
python - <<'PY'
import numpy as np
d=np.loadtxt('gpurun_out/r2g/vcg_trace.txt',dtype=np.int64)
wall=(d[:,3]-d[:,1]).astype(float)*0.01
clk=d[:,4].astype(float)
print("vcg K1: wall med %.1f us, shader clock med %.0f MHz"%(np.median(wall), np.median(clk/wall)))
PY
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/r2g/bench.json 2> gpurun_out/r2g/bench.err; echo "bench rc=$?"
python -c "
import json
d=json.loads(open('gpurun_out/r2g/bench.json').read().strip().splitlines()[-1])
print('bench', d['value'], d['ms_per_step'], d['config']['e_norm'])"
