cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out/r2c
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2c/pytest.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/r2c/pytest.log
APP="./laghos_amd/laghos -p 1 -m data/cube01_hex.mesh -rs 4 -ok 3 -ot 2 -ms 3 -pa"
for wt in 0 1; do
  LGH_PCG_WT=$wt LGH_PCG_TRACE=gpurun_out/r2c/trace_wt$wt.txt timeout 120 $APP > gpurun_out/r2c/app_wt$wt.log 2>&1; echo "rc=$?"
  python tools/pcg_trace_summary.py gpurun_out/r2c/trace_wt$wt.txt 8 > gpurun_out/r2c/summary_wt$wt.txt 2>&1
  tail -8 gpurun_out/r2c/summary_wt$wt.txt
  LGH_PCG_WT=$wt timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/r2c/bench_wt$wt.json 2> gpurun_out/r2c/bench_wt$wt.err; echo "bench wt=$wt rc=$?"
  python -c "
import json
d=json.loads(open('gpurun_out/r2c/bench_wt$wt.json').read().strip().splitlines()[-1])
print('WT=$wt', d['value'], d['ms_per_step'], d['config']['e_norm'])"
done
