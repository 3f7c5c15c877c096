cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out/r2t
APP="./laghos_amd/laghos -p 3 -m data/box01_hex.mesh -rs 4 -ok 5 -ot 4 -ms 2 -pa -f"
for k in 2 3 4; do
LGH_VCG_VARIANT=$k timeout 600 $APP > gpurun_out/r2t/c5_v$k.log 2>&1; echo "c5 v$k rc=$?"
grep -i "CG (H1) total" gpurun_out/r2t/c5_v$k.log | tail -9
done
APP="./laghos_amd/laghos -p 1 -m data/cube01_hex.mesh -rs 4 -ok 4 -ot 3 -ms 5 -pa -f"
for k in 2 1 3 4; do
LGH_VCG_VARIANT=$k timeout 600 $APP > gpurun_out/r2t/q4q3_v$k.log 2>&1; echo "q4 v$k rc=$?"
grep -i "CG (H1) total\|step  " gpurun_out/r2t/q4q3_v$k.log | tail -2
done
