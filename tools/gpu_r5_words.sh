# round 5: exact all-reduce of the (r, z) accumulator words on several ranks + the two-table ELL of K2:
# the multi-rank suites (emulated ranks, cross-process transport), the kernel-level K1 / K2 tests, then the bench line
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r5_words
rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_k1.py tests/test_gpu_k2.py tests/test_gpu_multiproc.py -q -x > $O/pytest_a.log 2>&1; tail -3 $O/pytest_a.log
timeout 1500 python -m pytest tests/test_gpu_pipeline.py -q -x -k "multi_rank or slab or k2 or force_multi" > $O/pytest_b.log 2>&1; tail -3 $O/pytest_b.log
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_configs.py -q -x > $O/pytest_c.log 2>&1; tail -3 $O/pytest_c.log
timeout 600 python bench.py --no-cpu-baseline --legs c3,c2multi --detail $O/bench_detail.json > $O/bench.json 2> $O/bench.err
python - <<PY
import json
d=json.load(open("$O/bench_detail.json"))
print(round(d["value"],1), round(d["ms_per_step"],3))
for k,v in d["kernels"].items(): print("   ", k.split(" ")[0], round(v["mean_us"],2), v["launches"], round(v["frac"],3), v.get("moves",""))
for k,v in d.get("legs",{}).items():
    print("  leg",k, round(v.get("value",0),1), round(v.get("ms_per_step",0),3), v.get("error"), v.get("ms_per_step_minus_single_rank_path"))
    for kk,vv in v.get("kernels",{}).items(): print("      ", kk.split(" ")[0], round(vv["mean_us"],2), round(vv["frac"],3))
    if "comm" in v: print("      comm", v["comm"])
PY
