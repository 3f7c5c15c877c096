cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out/r2t
timeout 900 python -m pytest tests -m gpu -q -x -k "Q4Q3 or Q5Q4 or config5" > gpurun_out/r2t/pytest.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/r2t/pytest.log
LGH_VCG_VARIANT=1 timeout 900 python -m pytest tests -m gpu -q -x -k "Q4Q3" > gpurun_out/r2t/pytest_v1.log 2>&1; echo "pytest v1 rc=$?"
tail -3 gpurun_out/r2t/pytest_v1.log
APP="./laghos_amd/laghos -p 3 -m data/box01_hex.mesh -rs 4 -ok 5 -ot 4 -ms 2 -pa -f"
for k in 2 0; do
LGH_VCG_VARIANT=$k timeout 600 $APP > gpurun_out/r2t/c5_v$k.log 2>&1; echo "rc=$?"
grep -i "CG (L2)\|CG (H1)\|UpdateQuadData\|major kernels\|step " gpurun_out/r2t/c5_v$k.log | tail -9
done
APP="./laghos_amd/laghos -p 1 -m data/cube01_hex.mesh -rs 4 -ok 4 -ot 3 -ms 5 -pa -f"
for k in 2 1 0; do
LGH_VCG_VARIANT=$k timeout 600 $APP > gpurun_out/r2t/q4q3_v$k.log 2>&1; echo "rc=$?"
grep -i "CG (L2)\|CG (H1)\|UpdateQuadData\|major kernels\|step " gpurun_out/r2t/q4q3_v$k.log | tail -9
done
