# round-3 profile set: the plain bench line, the same command under rocprofv3 --kernel-trace --stats, FETCH/WRITE passes at
# 32^3 and 64^3 (-> profiles/r3_pmc_traffic.json), memory-side / SQ counters of K1 (slab and plane), per-wavefront trace of
# the slab K1, the micro-benchmarks.  Everything lands under gpurun_out/prof_r3/ and is copied into profiles/ by hand.
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/prof_r3
rm -rf $O; mkdir -p $O
timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python bench.py --no-legs --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
f=$(find $O/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/bench_kernel_stats.csv
python tools/gap_summary.py $O/stats > $O/gaps.txt 2>&1
# FETCH / WRITE (separate passes, --kernel-trace only)
bash tools/gpu_r3_pmc_traffic.sh > $O/pmc_traffic.log 2>&1
cp gpurun_out/r3_pmc_traffic/*.txt gpurun_out/r3_pmc_traffic/r3_pmc_traffic.json $O/ 2>/dev/null
# K1 counters
VARIANTS="4 2" bash tools/gpu_r3_pmc.sh > $O/k1_pmc.log 2>&1
cp gpurun_out/r3_pmc/v4_summary.txt $O/k1_slab_pmc.txt; cp gpurun_out/r3_pmc/v2_summary.txt $O/k1_plane_pmc.txt
# per-wavefront trace of the slab K1 (C2)
LGH_VCG_TRACE=$O/slab_waves.trace timeout 300 python bench.py --steps 3 --warmup 1 --no-legs --no-cpu-baseline > /dev/null 2> $O/trace.err
python tools/k1_trace_summary.py $O/slab_waves.trace > $O/k1_slab_trace.txt 2>&1
LGH_VCG_TRACE_PHASES=1 LGH_VCG_TRACE=$O/slab_phases.trace timeout 300 python bench.py --steps 3 --warmup 1 --no-legs --no-cpu-baseline > /dev/null 2>> $O/trace.err
python tools/k1_trace_summary.py $O/slab_phases.trace mfma >> $O/k1_slab_trace.txt 2>&1
# micro-benchmarks
for b in ubench_k1_floor ubench_f64 ubench_gather; do [ -x tools/bin/$b ] && timeout 120 tools/bin/$b > $O/$b.txt 2>&1; done
timeout 60 tools/bin/ubench_k1_floor 64 >> $O/ubench_k1_floor.txt 2>&1
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete
tail -c 400 $O/bench.json; head -8 $O/bench_kernel_stats.csv | cut -c1-160; cat $O/k1_slab_trace.txt | head -30; tail -3 $O/pmc_traffic.log
