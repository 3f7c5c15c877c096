# Copies what tools/gpu_final_r6.sh left under gpurun_out/final_r6 into profiles/ under the names the documents cite.
cd /root/repo
O=gpurun_out/final_r6
grep '^{' $O/bench.json | tail -1 > profiles/r6_bench.json
grep '^{' $O/bench_under_rocprof.json | tail -1 > profiles/r6_bench_under_rocprofv3.json
cp $O/bench_kernel_stats.csv profiles/r6_bench_kernel_stats.csv
cp $O/gaps.txt profiles/r6_gpu_idle_gaps.txt
cp $O/r6_pmc_traffic.json profiles/r6_pmc_traffic.json
for w in c2 c3 tg; do for k in FETCH_SIZE WRITE_SIZE; do cp $O/${w}_$k.txt profiles/r6_pmc_${w}_$k.txt; done; done
{ tail -3 $O/pytest_gpu.log; tail -2 $O/smoke.log; } > profiles/r6_pytest_gpu_summary.txt
cp $O/q_pmc.txt profiles/r6_sq_counters.txt
cp $O/bench_detail.json profiles/r6_bench_detail.json
cp $O/step_timeline.txt profiles/r6_step_timeline.txt
