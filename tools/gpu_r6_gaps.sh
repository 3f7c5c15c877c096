# kernel trace of the plain headline run: per-kernel stats + idle gaps (steady state)
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r6_gaps
rm -rf $O; mkdir -p $O
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python bench.py --no-legs --no-cpu-baseline --no-roofline --steps 30 --warmup 5 --detail $O/detail.json > $O/bench.json 2> $O/bench.err
f=$(find $O/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats.csv
python tools/gap_summary.py $O/stats > $O/gaps.txt 2>&1
python tools/step_timeline.py $O/stats > $O/timeline.txt 2>&1
find $O -name "*kernel_trace.csv" -delete
rm -rf $O/stats
head -40 $O/timeline.txt
