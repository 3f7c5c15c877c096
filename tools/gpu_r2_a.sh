cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out/r2a
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/r2a/pytest.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/r2a/pytest.log
for v in 1 0; do
  LGH_PCG=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/r2a/bench_pcg$v.json 2> gpurun_out/r2a/bench_pcg$v.err; echo "bench pcg=$v rc=$?"
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r2a/bench_pcg$v.json").read().strip().splitlines()[-1])
    print("PCG=$v", d["value"], d["ms_per_step"], d["config"]["e_norm"])
except Exception as e:
    print("parse fail", e)
PY
done
LGH_PCG=1 LGH_OVERLAP=0 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/r2a/bench_pcg1_noov.json 2>&1; tail -c 400 gpurun_out/r2a/bench_pcg1_noov.json
