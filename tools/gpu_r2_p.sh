cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out/r2r
timeout 1800 python -m pytest tests -m gpu -q -x > gpurun_out/r2r/pytest.log 2>&1; echo "pytest rc=$?"
tail -8 gpurun_out/r2r/pytest.log
for k in 1 0; do
LGH_FUSED_F1=$k timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-legs > gpurun_out/r2r/bench_f1_$k.json 2> gpurun_out/r2r/err$k; echo "bench rc=$?"
python -c "
import json
d=json.loads(open('gpurun_out/r2r/bench_f1_$k.json').read().strip().splitlines()[-1])
print('F1=$k', round(d['value'],1), round(d['ms_per_step'],3), d['config']['e_norm'], {k.split()[0]:round(v['mean_us'],1) for k,v in d['kernels'].items()})"
done
