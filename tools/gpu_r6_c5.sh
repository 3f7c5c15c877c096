# config 5 (Q5Q4 triple point, 65 536 zones) as bench.py's c5 leg runs it: where a step goes on the GPU's timeline
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r6_c5
rm -rf $O; mkdir -p $O
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o c5 -- python tools/run_sim.py 1 3 -p 3 -m data/box01_hex.mesh -rs 4 -ok 5 -ot 4 > $O/run.log 2>&1
f=$(find $O/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats.csv
python tools/step_timeline.py $O/stats > $O/timeline.txt 2>&1
find $O -name "*kernel_trace.csv" -delete
rm -rf $O/stats
grep "ms per step" $O/run.log
head -40 $O/timeline.txt
