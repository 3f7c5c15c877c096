cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/q; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_configs.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
for v in 1 2; do
timeout 600 python bench.py --steps 20 --warmup 5 --no-legs --no-cpu-baseline > $O/b_$v.json 2> $O/b_$v.err; echo "rc=$?"
python - <<P
import json
d=json.loads([l for l in open("$O/b_$v.json") if l.startswith("{")][-1])
k1=[v for n,v in d["kernels"].items() if n.startswith("vcg_apply")][0]
q=[v for n,v in d["kernels"].items() if n.startswith("qpoint")][0]
print(round(d["value"],1), round(d["ms_per_step"],3), "K1", round(k1["mean_us"],1), "Q", round(q["mean_us"],1), repr(d["config"]["e_norm"]))
P
done
