# Round 4: element -> XCD mapping of the row-form quadrature update (LGH_Q_SWZ): time at C2 (early Sedov) and 64^3
# Taylor-Green, and the memory-side traffic of each mapping at C2.
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r4_q4; rm -rf $O; mkdir -p $O
for S in -1 0 3 5 7 10; do
  LGH_Q_SWZ=$S timeout 200 python bench.py --no-cpu-baseline --legs tg,c2dev --steps 10 --warmup 3 2>/dev/null | grep '^{' > $O/bench_$S.json
done
APP="./laghos_amd/laghos -p 1 -m data/cube01_hex.mesh -rs 4 -ok 3 -ot 2 -ms 3 -pa"
for S in -1 0 3 5 7 10; do
  export LGH_Q_SWZ=$S
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/f$S -o f --output-format csv -- $APP > $O/f$S.log 2>&1
  python tools/pmc_summary.py $O/f$S qrows_kernel > $O/fetch_$S.txt
done
find $O -name "*.csv" -delete
python - <<'PY' > $O/summary.txt 2>&1
import json, re
for f in (-1, 0, 3, 5, 7, 10):
    d = json.loads(open('gpurun_out/r4_q4/bench_%d.json' % f).read())
    q = [v for k, v in d['kernels'].items() if k.startswith(('qpoint', 'qrows'))][0]
    tq = [v for k, v in d['legs']['tg']['kernels'].items() if k.startswith(('qpoint', 'qrows'))][0]
    dq = [v for k, v in d['legs']['c2dev']['kernels'].items() if k.startswith(('qpoint', 'qrows'))][0]
    fe = re.search(r'FETCH_SIZE\s+med=([0-9.e+]+)', open('gpurun_out/r4_q4/fetch_%d.txt' % f).read())
    print('swz %3d: c2 %.3f ms/step qupdate %.1f us | c2dev qupdate %.1f us | tg %.2f ms/step qupdate %.1f us | FETCH_SIZE %s KB' % (f, d['ms_per_step'], q['mean_us'], dq['mean_us'], d['legs']['tg']['ms_per_step'], tq['mean_us'], fe.group(1) if fe else None))
PY
cat $O/summary.txt
