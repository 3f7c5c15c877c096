cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r2v; mkdir -p $O
for m in 0 1; do
LGH_FORCE_MULTI=$m timeout 600 python bench.py --steps 20 --warmup 5 --no-legs --no-cpu-baseline > $O/bench_m$m.json 2> $O/bench_m$m.err; echo "m=$m rc=$?"
python - <<P
import json
d=json.loads([l for l in open("$O/bench_m$m.json") if l.startswith("{")][-1])
print("multi=$m", d["value"], d["ms_per_step"])
P
done
LGH_FORCE_MULTI=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -o m1 -- python bench.py --steps 5 --warmup 2 --no-legs --no-cpu-baseline > $O/m1_prof.json 2> $O/m1_prof.err
head -30 $O/st/m1_kernel_stats.csv | cut -c1-160
find $O -name "*kernel_trace.csv" -delete
