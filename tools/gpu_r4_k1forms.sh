# Round 4: K1 forms at C2 on one box: slab (Kronecker), generic Kronecker (LDS stages), plane, slab through the quadrature points
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r4_k1forms; rm -rf $O; mkdir -p $O
run() { env "$@" timeout 300 python bench.py --no-cpu-baseline --no-legs --steps 10 --warmup 3 2>/dev/null | grep '^{' > $O/bench_$TAG.json; }
TAG=slab_kron run LGH_X=1
TAG=generic_kron run LGH_VCG_VARIANT=5
TAG=plane run LGH_VCG_VARIANT=2
TAG=slab_qp run LGH_MASS_KRON=0
python - <<'PY' > $O/summary.txt 2>&1
import json
for f in ("slab_kron", "generic_kron", "plane", "slab_qp"):
    d = json.loads(open('gpurun_out/r4_k1forms/bench_%s.json' % f).read())
    k1 = [(k, v) for k, v in d['kernels'].items() if k.startswith('vcg_apply')][0]
    k2 = [v for k, v in d['kernels'].items() if k.startswith('vcg_update')][0]
    print('%-13s %-20s K1 %.1f us K2 %.1f us | %.3f ms/step value %.1f' % (f, k1[0].split(' ')[0], k1[1]['mean_us'], k2['mean_us'], d['ms_per_step'], d['value']))
PY
cat $O/summary.txt
