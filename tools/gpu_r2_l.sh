cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out/r2l
timeout 900 python -m pytest tests -m gpu -q -x -k "qupdate or checks_table or readme_run4 or config2 or config3" > gpurun_out/r2l/pytest.log 2>&1; echo "pytest rc=$?"
tail -25 gpurun_out/r2l/pytest.log
for tg in 1e-30 -1; do
LGH_Q_TINY_GRAD=$tg timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-legs > gpurun_out/r2l/bench_$tg.json 2> gpurun_out/r2l/bench_$tg.err; echo "bench rc=$?"
python -c "
import json
d=json.loads(open('gpurun_out/r2l/bench_$tg.json').read().strip().splitlines()[-1])
print('tiny_grad=$tg', d['value'], d['ms_per_step'], d['config']['e_norm'], d['kernels']['qpoint_kernel (fused QUpdate)'])"
done
