# largest single-GPU runs through the driver: 128^3 zones (-rs 6) and 152^3 zones, peak VRAM polled from rocm-smi
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/big; mkdir -p $O
for case in rs6 n152; do
( while true; do rocm-smi --showmeminfo vram 2>/dev/null | grep -i "used" | head -1; sleep 2; done ) > $O/vram_$case.log 2>&1 &
SMI=$!
if [ $case = rs6 ]; then ARGS="-p 1 -m data/cube01_hex.mesh -rs 6 -ok 3 -ot 2 -ms 4 -pa -f"; else ARGS="-p 1 -dim 3 -nx 152 -ny 152 -nz 152 -rs 0 -ok 3 -ot 2 -ms 3 -pa -f"; fi
T0=$SECONDS; timeout 900 ./laghos_amd/laghos $ARGS > $O/$case.log 2>&1; echo "$case rc=$? real $((SECONDS-T0)) s" | tee -a $O/$case.log
kill $SMI
echo "max VRAM used (bytes): $(grep -o '[0-9]\{9,\}' $O/vram_$case.log | sort -n | tail -1)" >> $O/$case.log
grep -i "zones\|dofs\|CG (\|UpdateQuadData total\|^|      1\|real\|VRAM" $O/$case.log | tail -12
done
