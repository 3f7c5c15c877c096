# Round 4: Kronecker-form slab K1, one against two wavefronts per SIMD (LGH_SLAB_WPS) and store-wait, C2 and 64^3
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r4_kron2; rm -rf $O; mkdir -p $O
run() { env "$@" timeout 300 python bench.py --no-cpu-baseline --legs c3 --steps 20 --warmup 5 2>/dev/null | grep '^{' > $O/bench_$TAG.json; }
TAG=wps1 run LGH_X=1
TAG=wps2 run LGH_SLAB_WPS=2
TAG=wps2sw run LGH_SLAB_WPS=2 LGH_SLAB_STORE_WAIT=1
TAG=wps1nodyn run LGH_SLAB_DYN=0
python - <<'PY' > $O/summary.txt 2>&1
import json
for f in ("wps1", "wps2", "wps2sw", "wps1nodyn"):
    d = json.loads(open('gpurun_out/r4_kron2/bench_%s.json' % f).read())
    k1 = [v for k, v in d['kernels'].items() if k.startswith('vcg_apply')][0]
    c3 = d['legs']['c3']; c3k1 = [v for k, v in c3['kernels'].items() if k.startswith('vcg_apply')][0]
    print('%-10s c2 %.3f ms/step value %.1f K1 %.1f us | c3 %.2f ms/step value %.1f K1 %.1f us' % (f, d['ms_per_step'], d['value'], k1['mean_us'], c3['ms_per_step'], c3['value'], c3k1['mean_us']))
PY
cat $O/summary.txt
