cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out/r2o
for k in 1 0; do
LGH_K2P=$k timeout 900 python bench.py --workload c3 --steps 5 --warmup 2 --no-cpu-baseline --no-legs > gpurun_out/r2o/bench_c3_k2p$k.json 2> gpurun_out/r2o/err$k; echo "bench rc=$?"
python -c "
import json
d=json.loads(open('gpurun_out/r2o/bench_c3_k2p$k.json').read().strip().splitlines()[-1])
print('c3 K2P=$k', round(d['value'],1), round(d['ms_per_step'],2), d['config']['e_norm'], {k.split()[0]:round(v['mean_us'],1) for k,v in d['kernels'].items()})"
done
