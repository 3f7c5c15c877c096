# slab-form K1: bench + light trace for the settings in $CASES ("name:ENV=.. ENV=..")
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r3_slab2; mkdir -p $O
summ() {
python - <<P
import json
try:
    d=json.loads([l for l in open("$O/$1.json") if l.startswith("{")][-1])
    k1=[v for n,v in d["kernels"].items() if n.startswith("vcg_apply")][0]
    k2=[v for n,v in d["kernels"].items() if n.startswith("vcg_update")][0]
    print("$1", round(d["value"],1), round(d["ms_per_step"],3), "K1", round(k1["mean_us"],1), "K2", round(k2["mean_us"],1), "e_norm", d["config"]["e_norm"])
except Exception as ex:
    print("$1 FAILED", ex)
P
}
IFS=';'
for CASE in $CASES; do
  NAME="${CASE%%:*}"; ENVS="${CASE#*:}"
  IFS=' '
  env $ENVS LGH_VCG_VARIANT=4 python bench.py --steps 20 --warmup 5 --no-legs --no-cpu-baseline > $O/$NAME.json 2> $O/$NAME.err; summ $NAME
  if [ -n "$C3" ]; then env $ENVS LGH_VCG_VARIANT=4 python bench.py --workload c3 --steps 4 --warmup 2 --no-legs --no-cpu-baseline > $O/${NAME}_c3.json 2> $O/${NAME}_c3.err; summ ${NAME}_c3; fi
  env $ENVS LGH_VCG_VARIANT=4 LGH_VCG_TRACE=$O/$NAME.trace python bench.py --steps 3 --warmup 1 --no-legs --no-cpu-baseline --no-roofline > $O/tr.json 2> $O/tr.err
  python tools/k1_trace_summary.py $O/$NAME.trace | head -4
  IFS=';'
done
