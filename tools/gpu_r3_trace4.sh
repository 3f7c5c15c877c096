#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3_trace2; mkdir -p $O
LGH_VCG_TRACE_PHASES=1 LGH_VCG_TRACE=$O/slab_w1_ph.trace timeout 300 python bench.py --steps 3 --warmup 1 --legs none --no-cpu-baseline > $O/x.json 2> $O/x.err
python tools/k1_trace_summary.py $O/slab_w1_ph.trace mfma
