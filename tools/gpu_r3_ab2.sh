#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
run() {
  tag=$1; shift
  env "$@" timeout 300 python bench.py --legs c3 --no-cpu-baseline > gpurun_out/ab_$tag.json 2> gpurun_out/ab_$tag.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/ab_$tag.json").read().strip().splitlines()[-1])
print("$tag C2", round(d["value"],1), round(d["ms_per_step"],3), {k.split(" ")[0]: round(v["mean_us"],1) for k, v in d["kernels"].items()})
for k, v in d.get("legs", {}).items(): print("$tag", k, round(v.get("value", 0),1), round(v.get("ms_per_step", 0),2), {kk.split(" ")[0]: round(vv["mean_us"],1) for kk, vv in v.get("kernels", {}).items()})
PY
}
run nowait LGH_VCG_VARIANT=4
run wait LGH_VCG_VARIANT=4 LGH_SLAB_STORE_WAIT=1
run wait_w1 LGH_VCG_VARIANT=4 LGH_SLAB_STORE_WAIT=1 LGH_SLAB_WPS=1
run nowait_w1 LGH_VCG_VARIANT=4 LGH_SLAB_WPS=1
