# Round 4: (r, z) of the velocity CG kept in exact integer accumulators (LGH_RZ_LIMBS, K2 without the ticketed grid
# reduction): parity tests first, then A/B of the bench line against the ticketed reduction on one box.
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r4_rz; rm -rf $O; mkdir -p $O
(timeout 600 python -m pytest tests/test_gpu_k1.py -x -q 2>&1 | tail -8) > $O/tests_k1.log 2>&1
(timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "cg or slab or lockstep or kernel_switches or hydro_mult or velocity" 2>&1 | tail -12) > $O/tests_kernels.log 2>&1
(timeout 900 python -m pytest tests/test_gpu_pipeline.py -x -q 2>&1 | tail -12) > $O/tests_pipeline.log 2>&1
run() { env "$@" timeout 200 python bench.py --no-cpu-baseline --legs tg --steps 20 --warmup 5 2>/dev/null | grep '^{' > $O/bench_$TAG.json; }
TAG=limbs run LGH_X=1
TAG=ticket run LGH_RZ_LIMBS=0
TAG=limbs2 run LGH_X=1
TAG=ticket2 run LGH_RZ_LIMBS=0
python - <<'PY' > $O/summary.txt 2>&1
import json
for f in ("limbs", "ticket", "limbs2", "ticket2"):
    d = json.loads(open('gpurun_out/r4_rz/bench_%s.json' % f).read())
    ks = {k.split('<')[0].split('(')[0]: v['mean_us'] for k, v in d['kernels'].items()}
    print(f, 'c2 ms/step %.3f value %.1f | tg ms/step %.2f value %.1f |' % (d['ms_per_step'], d['value'], d['legs']['tg']['ms_per_step'], d['legs']['tg']['value']), {k: round(v, 1) for k, v in ks.items()})
PY
cat $O/tests_k1.log $O/tests_kernels.log $O/tests_pipeline.log $O/summary.txt
