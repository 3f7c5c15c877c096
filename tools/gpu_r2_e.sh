cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out/r2e
APP="./laghos_amd/laghos -p 1 -m data/cube01_hex.mesh -rs 4 -ok 3 -ot 2 -ms 3 -pa"
for g in 512 256; do
  tag=g$g
  LGH_PCG_GRID=$g LGH_PCG_TRACE=gpurun_out/r2e/trace_$tag.txt timeout 120 $APP > gpurun_out/r2e/app_$tag.log 2>&1; echo "$tag rc=$?"
  python tools/pcg_trace_summary.py gpurun_out/r2e/trace_$tag.txt 10 > gpurun_out/r2e/summary_$tag.txt 2>&1
  tail -6 gpurun_out/r2e/summary_$tag.txt | cut -c1-420
done
LGH_PCG=0 LGH_VCG_TRACE=gpurun_out/r2e/vcg_trace.txt timeout 120 $APP > gpurun_out/r2e/app_vcg.log 2>&1; echo "vcg rc=$?"
python - <<'PY'
import numpy as np
d=np.loadtxt('gpurun_out/r2e/vcg_trace.txt',dtype=np.int64)
t=d[:,1:4].astype(float)*0.01
t-=t[:,0].min()
b=d[:,0]
for lo,hi in ((0,256),(256,512)):
    m=(b>=lo)&(b<hi)
    print("vcg K1 bid",lo,hi,"start med %.2f loop end min/med/max %.1f %.1f %.1f end max %.1f"%(np.median(t[m,0]),t[m,1].min(),np.median(t[m,1]),t[m,1].max(),t[m,2].max()))
PY
