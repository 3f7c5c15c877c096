# zones per workgroup of the Kronecker K1 (vcg_apply_kron) at D = 6 (config 5) and D = 5 (Q4Q3): default against LGH_KRON_NEB=1
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r6_kronneb
rm -rf $O; mkdir -p $O
LGH_KRON_NEB=1 timeout 900 python -m pytest tests/test_gpu_k1.py tests/test_gpu_kernels.py -x -q -k "kron or forms or Q5Q4 or Q4Q3 or q4q3 or q5q4 or switch" > $O/pytest_neb1.log 2>&1; tail -2 $O/pytest_neb1.log
for i in 1 2; do
  for V in - 1; do
    if [ "$V" = "-" ]; then unset LGH_KRON_NEB; else export LGH_KRON_NEB=$V; fi
    timeout 900 python bench.py --legs c5 --no-cpu-baseline --detail $O/d_${V}_$i.json > /dev/null 2>> $O/err
    echo "LGH_KRON_NEB=$V q4q3:" $(timeout 300 python tools/run_sim.py 2 6 -p 1 -m data/cube01_hex.mesh -rs 3 -ok 4 -ot 3 2>/dev/null | grep "ms per step")
  done
done
python - <<PY
import json
for V in ("-", "1"):
    for i in (1,2):
        d=json.load(open("$O/d_%s_%d.json"%(V,i)))
        g=d["legs"]["c5"]
        print("LGH_KRON_NEB=%s"%V, "c5", round(g["value"],1), round(g["ms_per_step"],2), g["k_us"])
PY
