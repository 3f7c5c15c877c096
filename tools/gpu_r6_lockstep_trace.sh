# kernel traces of the N-rank path on one rank, one communicator: energy CG in lockstep vs after the velocity CG
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r6_lockstep_trace
rm -rf $O; mkdir -p $O
for mode in 1 0; do
LGH_FORCE_MULTI=1 LGH_COMM2=0 LGH_ENERGY_LOCKSTEP=$mode timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/s$mode -o run -- python tools/run_sim.py 5 30 -m data/cube01_hex.mesh -rs 4 -p 1 -ok 3 -ot 2 > $O/run$mode.txt 2> $O/err$mode.txt
f=$(find $O/s$mode -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_$mode.csv
python tools/gap_summary.py $O/s$mode > $O/gaps_$mode.txt 2>&1
python tools/step_timeline.py $O/s$mode > $O/timeline_$mode.txt 2>&1
rm -rf $O/s$mode
cat $O/run$mode.txt; head -30 $O/timeline_$mode.txt
done
