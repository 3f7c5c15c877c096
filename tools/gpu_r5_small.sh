# round 5: the small kernels around the solves - RK stage combinations in pairs, F^T v hand-over in one launch, vectorised compare
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r5_small
rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_pipeline.py -q -x > $O/pytest.log 2>&1; tail -3 $O/pytest.log
run() { w=$1; n=$2; shift; shift
  env "$@" timeout 300 python bench.py --workload $w --no-cpu-baseline --no-legs --steps 10 --warmup 3 --detail $O/$n.json > /dev/null 2> $O/$n.err
  python - <<PY
import json
d=json.load(open("$O/$n.json"))
k={kk.split(" ")[0]:round(v["mean_us"],2) for kk,v in d["kernels"].items()}
print("$n", round(d["value"],1), round(d["ms_per_step"],3), "Q", k.get("qrows_kernel"), "K2", k.get("vcg_update_p_k"), "K1", k.get("vcg_apply_slab346"))
PY
}
run c2 c2
run c2 c2_again
run tg tg
run c3 c3
