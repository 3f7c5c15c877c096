cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/q3; mkdir -p $O
APP="./laghos_amd/laghos -p 3 -m data/box01_hex.mesh -rs 4 -ok 5 -ot 4 -ms 2 -pa -f"
for nf in 1 6 3; do
LGH_Q_NF=$nf timeout 600 $APP > $O/c5_$nf.log 2>&1; echo "c5 nf=$nf rc=$?"
grep -i "UpdateQuadData total\|^step\|Repeat" $O/c5_$nf.log | tail -2
done
APP="./laghos_amd/laghos -p 1 -m data/cube01_hex.mesh -rs 4 -ok 4 -ot 3 -ms 5 -pa -f"
for nf in 3 6; do
LGH_Q_NF=$nf timeout 600 $APP > $O/q4_$nf.log 2>&1; echo "q4q3 nf=$nf rc=$?"
grep -i "UpdateQuadData total\|^step" $O/q4_$nf.log | tail -2
done
LGH_Q_NF=6 timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_configs.py -m gpu -q -x -k "Q4Q3 or Q5Q4 or config5" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
