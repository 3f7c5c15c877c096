# Round 4: 64^3 Sedov (c3 leg) with and without the exact (r, z) accumulators, alternating on one box
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r4_c3ab; rm -rf $O; mkdir -p $O
run() { env "$@" timeout 300 python bench.py --no-cpu-baseline --legs c3 --steps 8 --warmup 3 2>/dev/null | grep '^{' > $O/bench_$TAG.json; }
TAG=limbs run LGH_X=1
TAG=ticket run LGH_RZ_LIMBS=0
TAG=limbs2 run LGH_X=1
TAG=ticket2 run LGH_RZ_LIMBS=0
TAG=nodyn run LGH_SLAB_DYN=0
python - <<'PY' > $O/summary.txt 2>&1
import json
for f in ("limbs", "ticket", "limbs2", "ticket2", "nodyn"):
    d = json.loads(open('gpurun_out/r4_c3ab/bench_%s.json' % f).read())
    v = d['legs']['c3']
    ks = {k.split('<')[0].split('(')[0]: round(x['mean_us'], 1) for k, x in v['kernels'].items()}
    print(f, 'c3 ms/step %.2f value %.1f' % (v['ms_per_step'], v['value']), ks)
PY
cat $O/summary.txt
