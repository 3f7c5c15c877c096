#!/bin/bash
# Compact mass data: slab against plane K1 at C2 (32^3), slab switches.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
run() {
  tag=$1; shift
  env "$@" timeout 300 python bench.py --legs none > gpurun_out/r1b_$tag.json 2> gpurun_out/r1b_$tag.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/r1b_$tag.json").read().strip().splitlines()[-1])
print("$tag", round(d["value"],1), round(d["ms_per_step"],3), {k.split(" ")[0]: round(v["mean_us"],1) for k, v in d["kernels"].items()})
PY
}
run plane LGH_VCG_VARIANT=2
run slab LGH_VCG_VARIANT=4
run slab_w1 LGH_VCG_VARIANT=4 LGH_SLAB_WPS=1
run slab_static LGH_VCG_VARIANT=4 LGH_SLAB_DYN=0
run slab_w1_static LGH_VCG_VARIANT=4 LGH_SLAB_WPS=1 LGH_SLAB_DYN=0
run slab_stored LGH_VCG_VARIANT=4 LGH_MASS_RANK1=0
