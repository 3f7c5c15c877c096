#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3_trace2; mkdir -p $O
LGH_VCG_VARIANT=2 LGH_VCG_TRACE=$O/plane_c2.trace timeout 300 python bench.py --steps 3 --warmup 1 --legs none --no-cpu-baseline > $O/plane_c2.json 2> $O/plane_c2.err
python tools/k1_trace_summary.py $O/plane_c2.trace
head -3 $O/plane_c2.trace
