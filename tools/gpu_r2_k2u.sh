cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r2k2u; mkdir -p $O
for g in 4 6 8 3; do
LGH_K2_GRID=$g timeout 600 python bench.py --steps 20 --warmup 5 --no-legs --no-cpu-baseline > $O/b_$g.json 2> $O/b_$g.err; echo "grid=$g rc=$?"
python - <<P
import json
d=json.loads([l for l in open("$O/b_$g.json") if l.startswith("{")][-1])
k1=[v for n,v in d["kernels"].items() if n.startswith("vcg_apply")][0]
k2=[v for n,v in d["kernels"].items() if n.startswith("vcg_update")][0]
print("U=1 grid=$g", round(d["value"],1), round(d["ms_per_step"],3), "K1", round(k1["mean_us"],1), "K2", round(k2["mean_us"],1), repr(d["config"]["e_norm"]))
P
done
for cfg in "1 4" "2 2" "1 8"; do
set -- $cfg
LGH_K2_U=$1 LGH_K2_GRID=$2 timeout 600 python bench.py --workload c3 --steps 4 --warmup 2 --no-legs --no-cpu-baseline > $O/c3_$1_$2.json 2> $O/c3_$1_$2.err; echo "c3 U=$1 grid=$2 rc=$?"
python - <<P
import json
d=json.loads([l for l in open("$O/c3_$1_$2.json") if l.startswith("{")][-1])
k1=[v for n,v in d["kernels"].items() if n.startswith("vcg_apply")][0]
k2=[v for n,v in d["kernels"].items() if n.startswith("vcg_update")][0]
print("c3 U=$1 grid=$2", round(d["value"],1), round(d["ms_per_step"],3), "K1", round(k1["mean_us"],1), "K2", round(k2["mean_us"],1))
P
done
