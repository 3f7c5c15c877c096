# round 5: map formula again, now that the update no longer streams Jac0inv per point
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r5_mapformula2
rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_general_numbering.py -q -x -k "kernel_switches or qupdate or fused or stress or permuted or numbering or curved" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
run() { w=$1; n=$2; shift; shift
  env "$@" timeout 300 python bench.py --workload $w --no-cpu-baseline --no-legs --steps 10 --warmup 3 --detail $O/$n.json > /dev/null 2> $O/$n.err
  python - <<PY
import json
d=json.load(open("$O/$n.json"))
k={kk.split(" ")[0]:round(v["mean_us"],2) for kk,v in d["kernels"].items()}
print("$n", round(d["value"],1), round(d["ms_per_step"],3), "Q", k.get("qrows_kernel"), "K2", k.get("vcg_update_p_k"), "K1", k.get("vcg_apply_slab346"))
PY
}
run c2 c2_formula LGH_MAP_FORMULA=1
run c2 c2_map LGH_MAP_FORMULA=0
run c2 c2_formula_again LGH_MAP_FORMULA=1
run c2 c2_map_again LGH_MAP_FORMULA=0
run tg tg_formula LGH_MAP_FORMULA=1
run tg tg_map LGH_MAP_FORMULA=0
run c3 c3_formula LGH_MAP_FORMULA=1
run c3 c3_map LGH_MAP_FORMULA=0
