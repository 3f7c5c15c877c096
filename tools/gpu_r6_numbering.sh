# Round 6, first box: what the hot path costs under the numbering the reference's API would hand over (verdict item 1a):
# the headline leg next to `c2mfem` (MFEM-like numbering) and `c2perm` (random nodes and zones), one box.
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r6_numbering
rm -rf $O; mkdir -p $O
timeout 900 python bench.py --legs c2mfem,c2perm --no-cpu-baseline --detail $O/bench_detail.json > $O/bench.json 2> $O/bench.err
tail -c 3000 $O/bench.err
python - <<PY
import json
d=json.load(open("$O/bench_detail.json"))
print("c2", d["value"], d["ms_per_step"])
for k,v in d["legs"].items(): print(k, v.get("value"), v.get("ms_per_step"), v.get("k_us"), v.get("vcg_layout"), v.get("error"))
PY
