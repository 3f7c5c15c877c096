# config 5 (Q5Q4 triple point, 65 536 zones) on one GPU: kernel statistics of a short run of the C++ driver
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/c5; rm -rf $O; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr -o b -- ./laghos_amd/laghos -p 3 -m data/box01_hex.mesh -rs 4 -ok 5 -ot 4 -pa -ms 3 > $O/run.log 2>&1
tail -12 $O/run.log
f=$(find $O/tr -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats.csv && head -16 $O/kernel_stats.csv | cut -c1-170
python tools/gap_summary.py $O/tr | head -12
find $O -name "*kernel_trace.csv" -delete
