# round 5: elements per workgroup of the row-form update (LGH_Q_EPW), C2 and 64^3 Taylor-Green, one box
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r5_epw
rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "qupdate or fused or stress or kernel_switches" > $O/pytest1.log 2>&1; tail -2 $O/pytest1.log
LGH_Q_EPW=2 timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_configs.py -q -x -k "qupdate or fused or stress or config2 or config3 or config4" > $O/pytest2.log 2>&1; tail -2 $O/pytest2.log
LGH_Q_EPW=3 timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "qupdate or fused or stress" > $O/pytest3.log 2>&1; tail -2 $O/pytest3.log
run() { w=$1; n=$2; shift; shift
  env "$@" timeout 300 python bench.py --workload $w --no-cpu-baseline --no-legs --steps 8 --warmup 3 --detail $O/$n.json > /dev/null 2> $O/$n.err
  python - <<PY
import json
d=json.load(open("$O/$n.json"))
k={kk.split(" ")[0]:round(v["mean_us"],2) for kk,v in d["kernels"].items()}
print("$n", round(d["value"],1), round(d["ms_per_step"],3), "Q", k.get("qrows_kernel"), "K2", k.get("vcg_update_p_k"), "K1", k.get("vcg_apply_slab346"))
PY
}
for e in 1 2 3 4; do run c2 c2_epw$e LGH_Q_EPW=$e; done
run c2 c2_epw1_again LGH_Q_EPW=1
for e in 1 2 3; do run tg tg_epw$e LGH_Q_EPW=$e; done
for e in 1 2 3; do run tg tg_occ3_epw$e LGH_Q_EPW=$e LGH_Q_OCC4=0; done
for e in 1 2; do run c3 c3_epw$e LGH_Q_EPW=$e; done
