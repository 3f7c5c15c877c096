cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/whatif; mkdir -p $O
for w in 8 4 2 1 0; do
LGH_WHATIF_SLOTS=$w timeout 300 python bench.py --steps 2 --warmup 1 --no-legs --no-cpu-baseline > $O/b_$w.json 2> $O/b_$w.err; echo "w=$w rc=$?"
python - <<P
import json
d=json.loads([l for l in open("$O/b_$w.json") if l.startswith("{")][-1])
for n,v in d["kernels"].items():
    if n.startswith("vcg"): print("slots=$w", n[:16], round(v["mean_us"],1), v["launches"])
P
done
