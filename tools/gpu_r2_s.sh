cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out/r2s
timeout 900 python -m pytest tests -m gpu -q -x -k "Q4Q3 or Q5Q4 or config5" > gpurun_out/r2s/pytest.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/r2s/pytest.log
APP="./laghos_amd/laghos -p 3 -m data/box01_hex.mesh -rs 4 -ok 5 -ot 4 -ms 2 -pa -f"
for k in 1 0; do
LGH_L2_PLANE=$k timeout 600 $APP > gpurun_out/r2s/c5_plane$k.log 2>&1; echo "rc=$?"
grep -i "CG (L2)\|CG (H1)\|Forces\|UpdateQuadData\|major kernels\|FOM\|step " gpurun_out/r2s/c5_plane$k.log | tail -12
done
