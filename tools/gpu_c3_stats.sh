# 64^3 Sedov Q3Q2 (the mesh of BASELINE configs 2-3 on one GPU): kernel statistics of a short bench run
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/c3; rm -rf $O; mkdir -p $O
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr -o b -- python bench.py --workload c3 --steps 5 --warmup 2 --no-legs --no-cpu-baseline > $O/b.json 2> $O/b.err
f=$(find $O/tr -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats.csv && head -12 $O/kernel_stats.csv | cut -c1-140
find $O -name "*kernel_trace.csv" -delete
python - <<PY
import json
d=json.loads([l for l in open("$O/b.json").read().splitlines() if l.startswith("{")][-1]); print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"])
PY
