cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out/r2f
APP="./laghos_amd/laghos -p 1 -m data/cube01_hex.mesh -rs 4 -ok 3 -ot 2 -ms 3 -pa"
for w in 0 1; do
  tag=wide$w
  LGH_PCG_WIDE=$w LGH_PCG_TRACE=gpurun_out/r2f/trace_$tag.txt timeout 120 $APP > gpurun_out/r2f/app_$tag.log 2>&1; echo "$tag rc=$?"
  python tools/pcg_trace_summary.py gpurun_out/r2f/trace_$tag.txt 10 > gpurun_out/r2f/summary_$tag.txt 2>&1
  tail -6 gpurun_out/r2f/summary_$tag.txt | cut -c1-420
  grep "|e|" gpurun_out/r2f/app_$tag.log | tail -1
  LGH_PCG_WIDE=$w timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/r2f/bench_$tag.json 2> gpurun_out/r2f/bench_$tag.err; echo "bench rc=$?"
  python -c "
import json
d=json.loads(open('gpurun_out/r2f/bench_$tag.json').read().strip().splitlines()[-1])
print('$tag', d['value'], d['ms_per_step'], d['config']['e_norm'])"
done
