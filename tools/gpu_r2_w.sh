cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r2w; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -x -k "multi or rank or halo or rccl or forms_agree" > $O/pytest.log 2>&1; echo "pytest rc=$?"
tail -5 $O/pytest.log
for m in 0 1; do
LGH_FORCE_MULTI=$m timeout 600 python bench.py --steps 20 --warmup 5 --no-legs --no-cpu-baseline > $O/bench_m$m.json 2> $O/bench_m$m.err; echo "m=$m rc=$?"
python - <<P
import json
d=json.loads([l for l in open("$O/bench_m$m.json") if l.startswith("{")][-1])
print("multi=$m", d["value"], d["ms_per_step"], d["config"]["e_norm"])
P
done
