# largest single-GPU runs through the driver: 128^3 zones (-rs 6), peak VRAM polled from rocm-smi
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/big; mkdir -p $O
( while true; do rocm-smi --showmemuse --showmeminfo vram 2>/dev/null | grep -i "used" | head -1; sleep 2; done ) > $O/vram.log 2>&1 &
SMI=$!
timeout 900 ./laghos_amd/laghos -p 1 -m data/cube01_hex.mesh -rs 6 -ok 3 -ot 2 -ms 4 -pa -f > $O/rs6.log 2>&1; echo "rs6 rc=$?"
kill $SMI
grep -i "zones\|dofs\|^step\|CG (\|Forces\|UpdateQuadData\|Major kernels\|^|" $O/rs6.log | tail -24
sort -t: -k3 -n $O/vram.log | tail -1
