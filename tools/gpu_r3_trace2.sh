#!/bin/bash
# Slab K1 at C2 with compact mass data: wall-clock stamps and per-phase shader cycles (wave 0 of every workgroup).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3_trace2; mkdir -p $O
for w in 2 1; do
LGH_VCG_VARIANT=4 LGH_SLAB_WPS=$w LGH_VCG_TRACE=$O/slab_c2_w$w.trace timeout 300 python bench.py --steps 3 --warmup 1 --legs none --no-cpu-baseline > $O/slab_c2.json 2> $O/slab_c2.err
echo "== slab wps=$w light trace"; python tools/k1_trace_summary.py $O/slab_c2_w$w.trace mfma
LGH_VCG_VARIANT=4 LGH_SLAB_WPS=$w LGH_VCG_TRACE_PHASES=1 LGH_VCG_TRACE=$O/slab_c2_w${w}_ph.trace timeout 300 python bench.py --steps 3 --warmup 1 --legs none --no-cpu-baseline > $O/slab_c2.json 2> $O/slab_c2.err
echo "== slab wps=$w phase trace"; python tools/k1_trace_summary.py $O/slab_c2_w${w}_ph.trace mfma
done
tail -3 $O/slab_c2.err
