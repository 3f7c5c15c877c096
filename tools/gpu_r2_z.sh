cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r2z; mkdir -p $O
for w in 0 1; do
LGH_K1_WIDE=$w timeout 600 python bench.py --steps 20 --warmup 5 --no-legs --no-cpu-baseline > $O/bench_w$w.json 2> $O/bench_w$w.err; echo "w=$w rc=$?"
python - <<P
import json
d=json.loads([l for l in open("$O/bench_w$w.json") if l.startswith("{")][-1])
k=[v for n,v in d["kernels"].items() if n.startswith("vcg_apply")][0]
print("wide=$w", round(d["value"],1), round(d["ms_per_step"],3), "K1 us", round(k["mean_us"],1), repr(d["config"]["e_norm"]))
P
done
for w in 0 1; do
LGH_K1_WIDE=$w timeout 600 python bench.py --workload c3 --steps 4 --warmup 2 --no-legs --no-cpu-baseline > $O/c3_w$w.json 2> $O/c3_w$w.err; echo "w=$w rc=$?"
python - <<P
import json
d=json.loads([l for l in open("$O/c3_w$w.json") if l.startswith("{")][-1])
k=[v for n,v in d["kernels"].items() if n.startswith("vcg_apply")][0]
print("c3 wide=$w", round(d["value"],1), round(d["ms_per_step"],3), "K1 us", round(k["mean_us"],1))
P
done
