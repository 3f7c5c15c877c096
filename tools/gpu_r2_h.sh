cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out/r2h
APP="./laghos_amd/laghos -p 1 -m data/cube01_hex.mesh -rs 4 -ok 3 -ot 2 -ms 3 -pa"
for w in 1 0; do
  tag=wide$w
  LGH_PCG_CLOCK=1 LGH_PCG_WIDE=$w LGH_PCG_TRACE=gpurun_out/r2h/trace_$tag.txt timeout 120 $APP > gpurun_out/r2h/app_$tag.log 2>&1; echo "$tag rc=$?"
  python tools/pcg_trace_summary.py gpurun_out/r2h/trace_$tag.txt 10 > gpurun_out/r2h/summary_$tag.txt 2>&1
  tail -5 gpurun_out/r2h/summary_$tag.txt | cut -c1-420
  grep "|e|" gpurun_out/r2h/app_$tag.log | tail -1; grep "shader clock" gpurun_out/r2h/app_$tag.log | tail -1
done
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/r2h/bench.json 2> gpurun_out/r2h/bench.err; echo "bench rc=$?"
python -c "
import json
d=json.loads(open('gpurun_out/r2h/bench.json').read().strip().splitlines()[-1])
print('bench', d['value'], d['ms_per_step'], d['config']['e_norm'])"
timeout 600 python -m pytest tests -m gpu -x -q -k "cg_h1 or hydro_mult or checks_table or bit_identical or full_size" > gpurun_out/r2h/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r2h/pytest.log
