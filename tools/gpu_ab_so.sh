# A/B of two builds on one box: liblaghos_hip.so (new) vs liblaghos_hip_old.so (reference build), C2 and 64^3
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/ab; mkdir -p $O
cp laghos_amd/liblaghos_hip.so /tmp/new.so
run() {
python bench.py $2 --no-legs --no-cpu-baseline > $O/$1.json 2> $O/$1.err
python - <<P
import json
d=json.loads([l for l in open("$O/$1.json") if l.startswith("{")][-1])
k1=[v for n,v in d["kernels"].items() if n.startswith("vcg_apply")][0]
k2=[v for n,v in d["kernels"].items() if n.startswith("vcg_update")][0]
q=[v for n,v in d["kernels"].items() if n.startswith("qpoint")][0]
print("$1", round(d["value"],1), round(d["ms_per_step"],3), "K1", round(k1["mean_us"],1), "K2", round(k2["mean_us"],1), "Q", round(q["mean_us"],1))
P
}
for rep in 1 2; do
cp /tmp/new.so laghos_amd/liblaghos_hip.so; run new_c2_$rep "--steps 20 --warmup 5"
cp laghos_amd/liblaghos_hip_old.so laghos_amd/liblaghos_hip.so; run old_c2_$rep "--steps 20 --warmup 5"
done
cp /tmp/new.so laghos_amd/liblaghos_hip.so; run new_c3 "--workload c3 --steps 4 --warmup 2"
cp laghos_amd/liblaghos_hip_old.so laghos_amd/liblaghos_hip.so; run old_c3 "--workload c3 --steps 4 --warmup 2"
cp /tmp/new.so laghos_amd/liblaghos_hip.so; run new_c3b "--workload c3 --steps 4 --warmup 2"
cp /tmp/new.so laghos_amd/liblaghos_hip.so
