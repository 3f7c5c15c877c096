# Round 4: the Kronecker form of the slab K1 (compact mass data on a tensor-product rule): parity, then timing against
# the contraction through the quadrature points (LGH_MASS_KRON=0) on one box.
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r4_kron; rm -rf $O; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_k1.py tests/test_gpu_kernels.py -x -q -k "k1 or slab or mass_data or lockstep or cg_h1 or hydro_mult" 2>&1 | tail -15) > $O/tests.log 2>&1
for K in 1 0; do
  LGH_MASS_KRON=$K timeout 300 python bench.py --no-cpu-baseline --legs c3,tg --steps 20 --warmup 5 2>/dev/null | grep '^{' > $O/bench_kron$K.json
done
python - <<'PY' > $O/summary.txt 2>&1
import json
for f in (1, 0):
    d = json.loads(open('gpurun_out/r4_kron/bench_kron%d.json' % f).read())
    k1 = [v for k, v in d['kernels'].items() if k.startswith('vcg_apply')][0]
    k2 = [v for k, v in d['kernels'].items() if k.startswith('vcg_update')][0]
    c3 = d['legs']['c3']; c3k1 = [v for k, v in c3['kernels'].items() if k.startswith('vcg_apply')][0]
    tg = d['legs']['tg']
    print('kron %d: c2 %.3f ms/step value %.1f K1 %.1f us (%.0f GB/s) K2 %.1f us | c3 %.2f ms/step value %.1f K1 %.1f us (%.0f GB/s) | tg %.2f ms/step value %.1f | e_norm %.12e' % (
        f, d['ms_per_step'], d['value'], k1['mean_us'], k1['GBs'], k2['mean_us'], c3['ms_per_step'], c3['value'], c3k1['mean_us'], c3k1['GBs'], tg['ms_per_step'], tg['value'], d['config']['e_norm']))
PY
cat $O/tests.log $O/summary.txt
