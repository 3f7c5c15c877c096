# BASELINE config 2 (64^3 Taylor-Green Q3Q2): kernel statistics and idle gaps of a short bench run
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/tg; rm -rf $O; mkdir -p $O
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr -o b -- python bench.py --workload tg --steps 5 --warmup 2 --no-legs --no-cpu-baseline > $O/b.json 2> $O/b.err
python tools/gap_summary.py $O/tr > $O/gaps.txt; head -12 $O/gaps.txt
f=$(find $O/tr -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats.csv && head -22 $O/kernel_stats.csv | cut -c1-150
find $O -name "*kernel_trace.csv" -delete
python - <<PY
import json
d=json.loads([l for l in open("$O/b.json").read().splitlines() if l.startswith("{")][-1]); print(d["value"], d["ms_per_step"])
PY
