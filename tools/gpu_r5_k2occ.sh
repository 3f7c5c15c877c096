# round 5: K2 with six wavefronts per SIMD (<= 80 registers), node ranges per CU; QUpdate stage trace
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r5_k2occ
rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_k2.py -q -x > $O/pytest_k2.log 2>&1; tail -2 $O/pytest_k2.log
run() { # name, env...
  n=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-legs --steps 10 --warmup 3 --detail $O/$n.json > /dev/null 2> $O/$n.err
  python - <<PY
import json
d=json.load(open("$O/$n.json"))
k={kk.split(" ")[0]:round(v["mean_us"],2) for kk,v in d["kernels"].items()}
print("$n", round(d["value"],1), round(d["ms_per_step"],3), k.get("vcg_update_p_k"), k.get("vcg_apply_slab346"), k.get("qrows_kernel"))
PY
}
run occ4 LGH_K2_OCC=4
run occ_default LGH_K2_OCC=1
run occ6 LGH_K2_OCC=6
run occ_default_grid3 LGH_K2_GRID=3
run occ_default_grid6 LGH_K2_GRID=6
run occ_default_grid8 LGH_K2_GRID=8
run occ_default_grid12 LGH_K2_GRID=12
run occ6_grid6 LGH_K2_OCC=6 LGH_K2_GRID=6
run occ6_grid9 LGH_K2_OCC=6 LGH_K2_GRID=9
# 64^3
run3() { n=$1; shift
  env "$@" timeout 300 python bench.py --workload c3 --no-cpu-baseline --no-legs --steps 4 --warmup 2 --detail $O/$n.json > /dev/null 2> $O/$n.err
  python - <<PY
import json
d=json.load(open("$O/$n.json"))
k={kk.split(" ")[0]:round(v["mean_us"],2) for kk,v in d["kernels"].items()}
print("$n", round(d["value"],1), round(d["ms_per_step"],3), k.get("vcg_update_p_k"), k.get("vcg_apply_slab346"), k.get("qrows_kernel"))
PY
}
run3 c3_occ4 LGH_K2_OCC=4
run3 c3_occ_default LGH_K2_OCC=1
run3 c3_occ6 LGH_K2_OCC=6
# QUpdate stage trace (call 40 of a C2 run = the 8th step)
LGH_Q_TRACE=$O/q_trace_c2.txt timeout 300 python bench.py --no-cpu-baseline --no-legs --no-roofline --steps 10 --warmup 3 > /dev/null 2> $O/qtrace.err
python tools/q_trace_summary.py $O/q_trace_c2.txt > $O/q_trace_c2_summary.txt 2>&1; cat $O/q_trace_c2_summary.txt
rm -f $O/q_trace_c2.txt
