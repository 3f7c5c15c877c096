# Round 4: pipelined (persistent) row form of the quadrature update: parity, then timing against one element per
# workgroup (LGH_Q_GRID=0) and the point form (LGH_Q_FORM=0) on one box.
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r4_q2; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "qupdate or fused or hydro_mult or kernel_switches or problem7 or energies" 2>&1 | tail -15) > $O/tests.log 2>&1
run() { env "$@" timeout 200 python bench.py --no-cpu-baseline --legs tg --steps 10 --warmup 3 2>/dev/null | grep '^{' > $O/bench_$TAG.json; }
TAG=pipe run LGH_X=1
TAG=grid0 run LGH_Q_GRID=0
TAG=grid2 run LGH_Q_GRID=2
TAG=point run LGH_Q_FORM=0
python - <<'PY' > $O/summary.txt 2>&1
import json
for f in ("pipe", "grid0", "grid2", "point"):
    d = json.loads(open('gpurun_out/r4_q2/bench_%s.json' % f).read())
    q = [v for k, v in d['kernels'].items() if k.startswith(('qpoint', 'qrows'))][0]
    tq = [v for k, v in d['legs']['tg']['kernels'].items() if k.startswith(('qpoint', 'qrows'))][0]
    print(f, 'c2 ms/step %.3f value %.1f qupdate us %.1f | tg ms/step %.2f value %.1f qupdate us %.1f' % (d['ms_per_step'], d['value'], q['mean_us'], d['legs']['tg']['ms_per_step'], d['legs']['tg']['value'], tq['mean_us']))
PY
cat $O/tests.log $O/summary.txt
