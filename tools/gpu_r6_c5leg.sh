cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r6_c5leg
rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_configs.py -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 900 python bench.py --legs c5,tg --no-cpu-baseline --detail $O/detail.json > $O/bench.json 2> $O/err
python - <<PY
import json
d=json.load(open("$O/detail.json"))
print("c2", d["value"], d["ms_per_step"])
for k,v in d["legs"].items(): print(k, v.get("value"), v.get("ms_per_step"), v.get("k_us"), v.get("error"))
PY
