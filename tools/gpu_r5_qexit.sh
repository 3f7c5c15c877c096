# round 5: the quadrature update without its ticketed dt fold (per-wavefront atomic min) - parity of everything that uses
# the update, then timing against the build before (same box: the leg values)
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r5_qexit
rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_pipeline.py tests/test_gpu_configs.py -q -x > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 600 python bench.py --no-cpu-baseline --legs c3,tg,c2dev --detail $O/bench_detail.json > $O/bench.json 2> $O/bench.err
python - <<PY
import json
d=json.load(open("$O/bench_detail.json"))
print(round(d["value"],1), round(d["ms_per_step"],3))
for k,v in d["kernels"].items(): print("   ", k.split(" ")[0], round(v["mean_us"],2), v["launches"], round(v["frac"],3))
for k,v in d.get("legs",{}).items():
    print("  leg",k, round(v.get("value",0),1), round(v.get("ms_per_step",0),3), v.get("error"))
    for kk,vv in v.get("kernels",{}).items(): print("      ", kk.split(" ")[0], round(vv["mean_us"],2), round(vv["frac"],3))
PY
LGH_Q_TRACE=$O/q_trace_c2.txt timeout 300 python bench.py --no-cpu-baseline --no-legs --no-roofline --steps 10 --warmup 3 > /dev/null 2> $O/qtrace.err
python tools/q_trace_summary.py $O/q_trace_c2.txt > $O/q_trace_c2_summary.txt 2>&1; cat $O/q_trace_c2_summary.txt
rm -f $O/q_trace_c2.txt
