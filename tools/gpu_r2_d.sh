cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out/r2d
APP="./laghos_amd/laghos -p 1 -m data/cube01_hex.mesh -rs 4 -ok 3 -ot 2 -ms 3 -pa"
for bp in 0 1; do for pr in 0 1 2; do
  tag=bp${bp}_pr${pr}
  LGH_PCG_BPIPE=$bp LGH_PCG_PRIO=$pr LGH_PCG_TRACE=gpurun_out/r2d/trace_$tag.txt timeout 120 $APP > gpurun_out/r2d/app_$tag.log 2>&1; echo "$tag rc=$?"
  python tools/pcg_trace_summary.py gpurun_out/r2d/trace_$tag.txt 10 > gpurun_out/r2d/summary_$tag.txt 2>&1
  tail -5 gpurun_out/r2d/summary_$tag.txt | cut -c1-400
  grep -i "energy\|FOM\|step\b" gpurun_out/r2d/app_$tag.log | tail -3
done; done
for cfg in "1 0" "1 2" "0 2"; do set -- $cfg
  LGH_PCG_BPIPE=$1 LGH_PCG_PRIO=$2 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/r2d/bench_$1_$2.json 2> gpurun_out/r2d/bench_$1_$2.err; echo "bench bp=$1 pr=$2 rc=$?"
  python -c "
import json
d=json.loads(open('gpurun_out/r2d/bench_$1_$2.json').read().strip().splitlines()[-1])
print('bp=$1 pr=$2', d['value'], d['ms_per_step'], d['config']['e_norm'])"
done
