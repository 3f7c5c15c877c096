# final build of round 3: full GPU test suite, smoke(), the plain bench line and the same command under rocprofv3
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/final_r3
rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python bench.py --no-legs --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
f=$(find $O/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/bench_kernel_stats.csv
python tools/gap_summary.py $O/stats > $O/gaps.txt 2>&1
find $O -name "*kernel_trace.csv" -delete
python - <<PY
import json
d=json.loads([l for l in open("$O/bench.json").read().splitlines() if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"]["mean_launch_us"])
for k,v in d["legs"].items(): print(k, round(v["value"],1), round(v["ms_per_step"],2), v.get("traffic"))
PY
