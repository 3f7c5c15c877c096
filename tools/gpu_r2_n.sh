cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out/r2n
for cfg in "0 0" "1 0" "2 0" "0 1" "0 2" "1 1" "2 2"; do set -- $cfg
LGH_K2_STORE=$1 LGH_K1_STORE=$2 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-legs > gpurun_out/r2n/bench_$1_$2.json 2> gpurun_out/r2n/bench_$1_$2.err; echo "bench rc=$?"
python -c "
import json
d=json.loads(open('gpurun_out/r2n/bench_$1_$2.json').read().strip().splitlines()[-1])
print('K2ST=$1 K1ST=$2', round(d['value'],1), round(d['ms_per_step'],3), d['config']['e_norm'], {k.split()[0]:round(v['mean_us'],1) for k,v in d['kernels'].items()})"
done
