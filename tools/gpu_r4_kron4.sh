cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r4_kron4; rm -rf $O; mkdir -p $O
(timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_configs.py -x -q 2>&1 | tail -5) > $O/tests.log 2>&1
for K in 1 0; do
  LGH_MASS_KRON=$K timeout 400 python bench.py --no-cpu-baseline --legs c5,tg --steps 20 --warmup 5 2>/dev/null | grep '^{' > $O/bench_kron$K.json
done
python - <<'PY' > $O/summary.txt 2>&1
import json
for f in (1, 0):
    d = json.loads(open('gpurun_out/r4_kron4/bench_kron%d.json' % f).read())
    k1 = [v for k, v in d['kernels'].items() if k.startswith('vcg_apply')][0]
    l2 = [v for k, v in d['kernels'].items() if k.startswith('mass_apply_l2')][0]
    c5 = d['legs']['c5']; c5l2 = [v for k, v in c5['kernels'].items() if k.startswith('mass_apply_l2')][0]; c5k1 = [v for k, v in c5['kernels'].items() if k.startswith('vcg_apply')][0]
    tg = d['legs']['tg']
    print('kron %d: c2 %.3f ms/step value %.1f K1 %.1f us L2apply %.1f us | c5 %.1f ms/step value %.1f K1 %.1f us L2apply %.1f us | tg %.2f ms/step value %.1f' % (
        f, d['ms_per_step'], d['value'], k1['mean_us'], l2['mean_us'], c5['ms_per_step'], c5['value'], c5k1['mean_us'], c5l2['mean_us'], tg['ms_per_step'], tg['value']))
PY
cat $O/tests.log $O/summary.txt
