#!/bin/bash
# Compact (rank-1) mass data: parity tests, then bench legs with and without it.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "mass or k1 or switches or cg or hydro" > gpurun_out/rank1_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/rank1_tests.log
tail -5 gpurun_out/rank1_tests.log
for r in 1 0; do
  LGH_MASS_RANK1=$r timeout 600 python bench.py --legs c3,c5 > gpurun_out/rank1_bench_$r.json 2> gpurun_out/rank1_bench_$r.err
  echo "rank1=$r rc=$?"
  python - <<PY
import json
d = json.loads(open("gpurun_out/rank1_bench_$r.json").read().strip().splitlines()[-1])
print("C2", d["value"], d["ms_per_step"], d["roofline"])
for k, v in d.get("legs", {}).items():
    print(k, v.get("value"), v.get("ms_per_step"))
print({k: v for k, v in d.get("kernels", {}).items()} if "kernels" in d else "")
PY
done
