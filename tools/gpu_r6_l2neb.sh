# zones per workgroup of the L2 Kronecker kernel: the default (two or three rounds of rows per stage) against LGH_L2_NEB=1 (one row per thread)
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r6_l2neb
rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_pipeline.py -x -q -k "l2 or L2 or energy or readme or config" > $O/pytest_default.log 2>&1; tail -2 $O/pytest_default.log
LGH_L2_NEB=1 timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_pipeline.py -x -q -k "l2 or L2 or energy or readme or config" > $O/pytest_neb1.log 2>&1; tail -2 $O/pytest_neb1.log
for i in 1 2; do
  for V in - 1; do
    if [ "$V" = "-" ]; then unset LGH_L2_NEB; else export LGH_L2_NEB=$V; fi
    timeout 900 python bench.py --legs c5 --no-cpu-baseline --detail $O/d_${V}_$i.json > /dev/null 2>> $O/err
  done
done
python - <<PY
import json
for V in ("-", "1"):
    for i in (1,2):
        d=json.load(open("$O/d_%s_%d.json"%(V,i)))
        k=d["kernels"]; g=d["legs"]["c5"]
        print("LGH_L2_NEB=%s"%V, "c2", round(d["value"],1), round(d["ms_per_step"],3), {n.split(" ")[0]: round(v["mean_us"],1) for n,v in k.items() if "l2" in n},
              "c5", round(g["value"],1), round(g["ms_per_step"],2), {n.split(" ")[0]: round(v["mean_us"],1) for n,v in g["kernels"].items() if "l2" in n})
PY
