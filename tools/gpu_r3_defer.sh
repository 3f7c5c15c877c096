#!/bin/bash
# Deferred fold of the exact (d, A d) accumulators: parity tests, then slab K1 at C2 with and without it, trace.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "slab or k1 or switches or mass_data" > gpurun_out/defer_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/defer_tests.log
tail -4 gpurun_out/defer_tests.log
run() {
  tag=$1; shift
  env "$@" timeout 300 python bench.py --legs none --no-cpu-baseline > gpurun_out/defer_$tag.json 2> gpurun_out/defer_$tag.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/defer_$tag.json").read().strip().splitlines()[-1])
print("$tag", round(d["value"],1), round(d["ms_per_step"],3), {k.split(" ")[0]: round(v["mean_us"],1) for k, v in d["kernels"].items()})
PY
}
run plane LGH_VCG_VARIANT=2
run slab LGH_VCG_VARIANT=4
run slab_nodefer LGH_VCG_VARIANT=4 LGH_SLAB_DEFER=0
run slab_w1 LGH_VCG_VARIANT=4 LGH_SLAB_WPS=1
O=gpurun_out/r3_trace2; mkdir -p $O
LGH_VCG_VARIANT=4 LGH_VCG_TRACE=$O/slab_defer.trace timeout 300 python bench.py --steps 3 --warmup 1 --legs none --no-cpu-baseline > $O/x.json 2> $O/x.err
python tools/k1_trace_summary.py $O/slab_defer.trace
