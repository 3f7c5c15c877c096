# round 5: one Jac0inv per zone in the row-form update (LGH_JAC0_COMPACT=0: point values) - parity (incl. a curved initial mesh), A/B
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r5_jac0
rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_general_numbering.py tests/test_gpu_configs.py -q -x > $O/pytest.log 2>&1; tail -3 $O/pytest.log
run() { w=$1; n=$2; shift; shift
  env "$@" timeout 300 python bench.py --workload $w --no-cpu-baseline --no-legs --steps 10 --warmup 3 --detail $O/$n.json > /dev/null 2> $O/$n.err
  python - <<PY
import json
d=json.load(open("$O/$n.json"))
k={kk.split(" ")[0]:(round(v["mean_us"],2), round(v["frac"],3)) for kk,v in d["kernels"].items()}
print("$n", round(d["value"],1), round(d["ms_per_step"],3), "Q", k.get("qrows_kernel"), "K2", k.get("vcg_update_p_k"), "K1", k.get("vcg_apply_slab346"))
PY
}
run c2 c2_compact LGH_JAC0_COMPACT=1
run c2 c2_points LGH_JAC0_COMPACT=0
run c2 c2_compact_occ4 LGH_JAC0_COMPACT=1 LGH_Q_OCC4=1
run c2 c2_compact_again LGH_JAC0_COMPACT=1
run tg tg_compact LGH_JAC0_COMPACT=1
run tg tg_points LGH_JAC0_COMPACT=0
run tg tg_compact_occ3 LGH_JAC0_COMPACT=1 LGH_Q_OCC4=0
run c3 c3_compact LGH_JAC0_COMPACT=1
run c3 c3_points LGH_JAC0_COMPACT=0
run c3 c3_compact_occ4 LGH_JAC0_COMPACT=1 LGH_Q_OCC4=1
LGH_Q_TRACE=$O/q_trace_c2.txt LGH_Q_TRACE_CALL=41 timeout 300 python bench.py --no-cpu-baseline --no-legs --no-roofline --steps 10 --warmup 3 > /dev/null 2> $O/qtrace.err
python tools/q_trace_summary.py $O/q_trace_c2.txt > $O/q_trace_c2_summary.txt 2>&1; cat $O/q_trace_c2_summary.txt
rm -f $O/q_trace_c2.txt
