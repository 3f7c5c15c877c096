# round 5: rows of four doubles as two half-arrays of pairs in the update's LDS (no bank conflicts on the 16-byte row reads)
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r5_lds
rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_configs.py tests/test_gpu_general_numbering.py -q -x > $O/pytest.log 2>&1; tail -3 $O/pytest.log
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
APP="./laghos_amd/laghos -p 1 -m data/cube01_hex.mesh -rs 4 -ok 3 -ot 2 -ms 3 -pa"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES SQ_ACTIVE_INST_LDS -d $O/qa -o a --output-format csv -- $APP > $O/qa.log 2>&1
python tools/pmc_summary.py $O/qa qrows_kernel
find $O/qa -name "*.csv" -delete
