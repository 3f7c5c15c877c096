# round 5: merged E-vector layout of the slab K1 - parity first (K1 / K2 one-launch tests, switches, multi-rank emulation),
# then the bench line with the merged layout and with LGH_SLAB_MERGE=0 on the same box
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r5_merge
rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_k1.py tests/test_gpu_k2.py -q -x > $O/pytest_k1k2.log 2>&1; tail -3 $O/pytest_k1k2.log
timeout 1500 python -m pytest tests -q -m gpu -x --deselect tests/test_gpu_k1.py --deselect tests/test_gpu_k2.py > $O/pytest_rest.log 2>&1; tail -3 $O/pytest_rest.log
timeout 600 python bench.py --no-cpu-baseline --legs c3,c2multi --detail $O/bench_merge_detail.json > $O/bench_merge.json 2> $O/bench_merge.err
LGH_SLAB_MERGE=0 timeout 600 python bench.py --no-cpu-baseline --legs c3,c2multi --detail $O/bench_nomerge_detail.json > $O/bench_nomerge.json 2> $O/bench_nomerge.err
python - <<PY
import json
for n in ("merge","nomerge"):
    d=json.load(open("$O/bench_%s_detail.json"%n))
    print(n, round(d["value"],1), round(d["ms_per_step"],3))
    for k,v in d["kernels"].items(): print("   ", k.split(" ")[0], round(v["mean_us"],2), v["launches"])
    for k,v in d.get("legs",{}).items():
        print("  leg",k, round(v.get("value",0),1), round(v.get("ms_per_step",0),3), v.get("error"))
        for kk,vv in v.get("kernels",{}).items(): print("      ", kk.split(" ")[0], round(vv["mean_us"],2))
PY
