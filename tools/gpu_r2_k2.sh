cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r2k2; mkdir -p $O
for cfg in "1 2" "0 2" "1 1" "1 3" "1 4"; do
set -- $cfg
LGH_K2_SKIP=$1 LGH_K2_GRID=$2 timeout 600 python bench.py --steps 20 --warmup 5 --no-legs --no-cpu-baseline > $O/b_$1_$2.json 2> $O/b_$1_$2.err; echo "skip=$1 grid=$2 rc=$?"
python - <<P
import json
d=json.loads([l for l in open("$O/b_$1_$2.json") if l.startswith("{")][-1])
k1=[v for n,v in d["kernels"].items() if n.startswith("vcg_apply")][0]
k2=[v for n,v in d["kernels"].items() if n.startswith("vcg_update")][0]
print("skip=$1 grid=$2", round(d["value"],1), round(d["ms_per_step"],3), "K1", round(k1["mean_us"],1), "K2", round(k2["mean_us"],1), repr(d["config"]["e_norm"]))
P
done
