# where the GPU idles: gaps between consecutive kernels of a short bench run (rocprofv3 kernel trace; the trace itself
# widens the gaps: K1 <-> K2 4.7 us here, ~2.5 us untraced; every host look costs 20-35 us until the next launch)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/gaps; rm -rf $O; mkdir -p $O
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr -o b -- python bench.py --steps 5 --warmup 2 --no-legs --no-cpu-baseline > $O/b.json 2> $O/b.err
python tools/gap_summary.py $O/tr > $O/gaps.txt; cat $O/gaps.txt
f=$(find $O/tr -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats.csv && head -25 $O/kernel_stats.csv | cut -c1-200
find $O -name "*kernel_trace.csv" -delete
