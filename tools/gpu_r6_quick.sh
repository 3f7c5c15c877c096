# quick check point: the CG-related GPU tests + the headline without legs (several samples)
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r6_quick
rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_k1.py tests/test_gpu_k2.py tests/test_gpu_kernels.py tests/test_gpu_general_numbering.py -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
for i in 1 2 3; do timeout 600 python bench.py --no-legs --no-cpu-baseline --detail $O/bench_detail_$i.json > $O/bench_$i.json 2>> $O/bench.err; done
python - <<PY
import json
for i in (1,2,3):
    d=json.load(open("$O/bench_detail_%d.json"%i))
    k=d["kernels"]
    print(d["value"], d["ms_per_step"], {n.split(" ")[0].split("<")[0]: round(v["mean_us"],1) for n,v in k.items()})
PY
