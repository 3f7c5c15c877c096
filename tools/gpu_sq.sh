cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/sq; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_pipeline.py -m gpu -q -x -k "not readme_run9" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
APP="./laghos_amd/laghos -p 3 -m data/box01_hex.mesh -rs 4 -ok 5 -ot 4 -ms 2 -pa -f"
timeout 600 $APP > $O/c5.log 2>&1; echo "c5 rc=$?"
grep -i "UpdateQuadData total" $O/c5.log
for v in 1 2; do
timeout 600 python bench.py --steps 20 --warmup 5 --no-legs --no-cpu-baseline > $O/b.json 2> $O/b.err
python - <<P
import json
d=json.loads([l for l in open("$O/b.json") if l.startswith("{")][-1])
q=[v for n,v in d["kernels"].items() if n.startswith("qpoint")][0]
print(round(d["value"],1), round(d["ms_per_step"],3), "Q", round(q["mean_us"],1), repr(d["config"]["e_norm"]))
P
done
