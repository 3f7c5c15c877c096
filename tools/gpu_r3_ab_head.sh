#!/bin/bash
# same-box A/B: working tree library against the library of the last commit (tools/bin/head/, built by hand)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
LEGS=${LEGS:-c3}
run() {
  tag=$1; shift
  env "$@" timeout 300 python bench.py --legs $LEGS --no-cpu-baseline > gpurun_out/ab_$tag.json 2> gpurun_out/ab_$tag.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/ab_$tag.json").read().strip().splitlines()[-1])
print("$tag C2", round(d["value"],1), round(d["ms_per_step"],3), {k.split(" ")[0]: round(v["mean_us"],1) for k, v in d["kernels"].items()})
for k, v in d.get("legs", {}).items(): print("$tag", k, round(v.get("value", 0),1), round(v.get("ms_per_step", 0),2), {kk.split(" ")[0]: round(vv["mean_us"],1) for kk, vv in v.get("kernels", {}).items()})
PY
}
cp laghos_amd/liblaghos_hip.so /tmp/work.so
run work1
cp tools/bin/head/liblaghos_hip.so laghos_amd/liblaghos_hip.so
run head1
cp /tmp/work.so laghos_amd/liblaghos_hip.so
run work2
cp tools/bin/head/liblaghos_hip.so laghos_amd/liblaghos_hip.so
run head2
cp /tmp/work.so laghos_amd/liblaghos_hip.so
