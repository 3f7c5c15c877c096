#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
run() {
  tag=$1; shift
  env "$@" timeout 300 python bench.py --steps 4 --warmup 1 --legs c3 --no-cpu-baseline > gpurun_out/sw_$tag.json 2> gpurun_out/sw_$tag.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/sw_$tag.json").read().strip().splitlines()[-1])
for k, v in d.get("legs", {}).items(): print("$tag", k, round(v.get("value", 0),1), round(v.get("ms_per_step", 0),2), {kk.split(" ")[0]: round(vv["mean_us"],1) for kk, vv in v.get("kernels", {}).items() if "vcg" in kk})
PY
}
run wait1
run wait0 LGH_SLAB_STORE_WAIT=0
run wait1b
run wait0_w2 LGH_SLAB_STORE_WAIT=0 LGH_SLAB_WPS=2
run wait1_w2 LGH_SLAB_WPS=2
