cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r2y; mkdir -p $O
for t in -1 1e-30 1e-14 1e-8; do
LGH_Q_TINY_GRAD=$t timeout 600 python bench.py --steps 20 --warmup 5 --no-legs --no-cpu-baseline > $O/bench_t$t.json 2> $O/bench_t$t.err; echo "t=$t rc=$?"
python - <<P
import json
d=json.loads([l for l in open("$O/bench_t$t.json") if l.startswith("{")][-1])
k=[v for n,v in d["kernels"].items() if n.startswith("qpoint")][0]
print("tiny=$t", round(d["value"],1), round(d["ms_per_step"],3), "qupdate us", round(k["mean_us"],1), repr(d["config"]["e_norm"]))
P
done
