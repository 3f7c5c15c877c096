cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r4_kron5; rm -rf $O; mkdir -p $O
(timeout 2200 python -m pytest tests -q -m gpu -x 2>&1 | tail -8) > $O/tests.log 2>&1
timeout 400 python bench.py --no-cpu-baseline --legs c5,c3 --steps 20 --warmup 5 2>/dev/null | grep '^{' > $O/bench.json
python - <<'PY' > $O/summary.txt 2>&1
import json
d = json.loads(open('gpurun_out/r4_kron5/bench.json').read())
k1 = [v for k, v in d['kernels'].items() if k.startswith('vcg_apply')][0]
c5 = d['legs']['c5']; c5l2 = [v for k, v in c5['kernels'].items() if k.startswith('mass_apply_l2')][0]; c5k1 = [(k, v) for k, v in c5['kernels'].items() if k.startswith('vcg_apply')][0]
print('c2 %.3f ms/step value %.1f K1 %.1f us | c5 %.1f ms/step value %.1f K1 %s %.1f us L2apply %.1f us | c3 %.2f value %.1f' % (
    d['ms_per_step'], d['value'], k1['mean_us'], c5['ms_per_step'], c5['value'], c5k1[0].split(" ")[0], c5k1[1]['mean_us'], c5l2['mean_us'], d['legs']['c3']['ms_per_step'], d['legs']['c3']['value']))
for k, v in c5['kernels'].items(): print('  c5', k, round(v['mean_us'], 1), v['launches'])
PY
cat $O/tests.log $O/summary.txt
