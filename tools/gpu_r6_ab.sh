# same-box A/B of one environment switch on the headline: usage  gpu_r6_ab.sh VAR VALUE_A VALUE_B  (value "-" = unset)
cd /root/repo
export TMPDIR=/tmp
VAR=$1; A=$2; B=$3
O=gpurun_out/r6_ab_$VAR
rm -rf $O; mkdir -p $O
for i in 1 2 3; do
  for V in $A $B; do
    if [ "$V" = "-" ]; then unset $VAR; else export $VAR=$V; fi
    timeout 600 python bench.py --no-legs --no-cpu-baseline --detail $O/d_${V}_$i.json > /dev/null 2>> $O/err
  done
done
python - <<PY
import json
for V in ("$A", "$B"):
    for i in (1,2,3):
        d=json.load(open("$O/d_%s_%d.json"%(V,i)))
        k=d["kernels"]
        print("$VAR=%s"%V, round(d["value"],1), round(d["ms_per_step"],3), {n.split(" ")[0].split("<")[0]: round(v["mean_us"],1) for n,v in k.items() if "force" not in n})
PY
