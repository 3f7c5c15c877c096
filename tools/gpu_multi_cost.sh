# cost of the multi-rank sequencing on one GPU: LGH_FORCE_MULTI=1 runs it over RCCL on a communicator of size 1
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/multi; mkdir -p $O
for m in 0 1 0 1; do
LGH_FORCE_MULTI=$m timeout 600 python bench.py --steps 20 --warmup 5 --no-legs --no-cpu-baseline > $O/b_$m.json 2> $O/b_$m.err; echo "m=$m rc=$?"
python - <<P
import json
d=json.loads([l for l in open("$O/b_$m.json") if l.startswith("{")][-1])
print("multi=$m", round(d["value"],1), round(d["ms_per_step"],3), {k[:12]: round(v["mean_us"],1) for k,v in d["kernels"].items()}, repr(d["config"]["e_norm"]))
P
done
