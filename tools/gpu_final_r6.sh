# Final build of round 6: PMC traffic passes (c2 / c3 / tg) -> profiles/r6_pmc_traffic.json (written in the box's copy of
# the repo so that the bench line that follows quotes it, and into gpurun_out for the merge back), the full GPU test
# suite, smoke(), the plain bench line, the same command under rocprofv3 --kernel-trace --stats (kernel summary, idle gaps).
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/final_r6
rm -rf $O; mkdir -p $O
bash tools/gpu_pmc_traffic.sh > $O/pmc_traffic.log 2>&1
cp gpurun_out/pmc_traffic/*.txt $O/ 2>/dev/null
cp profiles/r6_pmc_traffic.json $O/ 2>/dev/null
# SQ counters of the quadrature update on the final build (two passes, --kernel-trace only): instructions per wavefront, LDS conflicts
APP="./laghos_amd/laghos -p 1 -m data/cube01_hex.mesh -rs 4 -ok 3 -ot 2 -ms 3 -pa"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/qa -o a --output-format csv -- $APP > $O/qa.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVES -d $O/qb -o b --output-format csv -- $APP > $O/qb.log 2>&1
for P in qa qb; do python tools/pmc_summary.py $O/$P qrows_kernel vcg_update_p_k vcg_apply_slab346 >> $O/q_pmc.txt; done
find $O/qa $O/qb -name "*.csv" -delete
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 1200 python bench.py --detail $O/bench_detail.json > $O/bench.json 2> $O/bench.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python bench.py --no-legs --no-cpu-baseline --detail $O/bench_under_rocprof_detail.json > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
f=$(find $O/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/bench_kernel_stats.csv
python tools/gap_summary.py $O/stats > $O/gaps.txt 2>&1
# (the per-step timeline wants a trace of timed steps only: a run of its own without the per-kernel sampling steps behind them)
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O/stats2 -o steps -- python bench.py --no-legs --no-cpu-baseline --no-roofline --steps 30 --warmup 5 --detail $O/steps_detail.json > $O/steps.json 2> $O/steps.err
python tools/step_timeline.py $O/stats2 > $O/step_timeline.txt 2>&1
rm -rf $O/stats2
find $O -name "*kernel_trace.csv" -delete
rm -rf $O/stats
python - <<PY
import json
d=json.loads([l for l in open("$O/bench.json").read().splitlines() if l.startswith("{")][-1])
r=d["roofline"]
print(d["value"], d["ms_per_step"], r["kernel"].split(" ")[0], r["frac"], r["traffic"], r["mean_launch_us"])
for k,v in d["legs"].items(): print(k, round(v.get("value",0),1), round(v.get("ms_per_step",0),2), v.get("error"))
print("parity", d.get("parity",{}).get("pass"))
PY
