cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out/r2m
timeout 1200 python -m pytest tests -m gpu -q -x -k "cg_h1 or hydro_mult or checks_table or bit_identical or full_size or config or readme_run4 or q3q2" > gpurun_out/r2m/pytest.log 2>&1; echo "pytest rc=$?"
tail -8 gpurun_out/r2m/pytest.log
for k in 1 0; do
LGH_K2P=$k timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-legs > gpurun_out/r2m/bench_k2p$k.json 2> gpurun_out/r2m/bench_k2p$k.err; echo "bench rc=$?"
python -c "
import json
d=json.loads(open('gpurun_out/r2m/bench_k2p$k.json').read().strip().splitlines()[-1])
print('K2P=$k', d['value'], d['ms_per_step'], d['config']['e_norm'], {k.split()[0]:round(v['mean_us'],1) for k,v in d['kernels'].items()})"
done
