# Round 6: the library's own zone order / node numbering (lgh_order.hip).  Parity on the permuted meshes in both orders,
# the kernel-level hooks, then the numbering legs again (one box: headline, c2mfem, c2perm - and the same with LGH_ORDER=0).
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r6_order
rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_general_numbering.py tests/test_gpu_k1.py tests/test_gpu_k2.py -x -q > $O/pytest.log 2>&1; tail -15 $O/pytest.log
timeout 900 python bench.py --legs c2mfem,c2perm --no-cpu-baseline --detail $O/bench_detail.json > $O/bench.json 2> $O/bench.err
LGH_ORDER=0 timeout 900 python bench.py --legs c2mfem,c2perm --no-cpu-baseline --detail $O/bench_detail_order0.json > $O/bench_order0.json 2> $O/bench_order0.err
tail -c 1500 $O/bench.err
python - <<PY
import json
for f in ("bench_detail.json", "bench_detail_order0.json"):
    d=json.load(open("$O/"+f))
    print(f, "c2", d["value"], d["ms_per_step"])
    for k,v in d["legs"].items(): print("  ", k, v.get("value"), v.get("ms_per_step"), v.get("k_us"), v.get("vcg_layout"), v.get("error"))
PY
