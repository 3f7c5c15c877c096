# Round 4 experiment (timing only - the results of these runs are wrong on purpose): what the slab K1 would cost if the
# lanes of a pair / a quad gathered from the same cache line (LGH_DBG_SHARE=2 / 4: the lanes copy their neighbour's row offset).
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r4_share; rm -rf $O; mkdir -p $O
for S in 0 2 4; do
  LGH_DBG_SHARE=$S timeout 300 python bench.py --no-cpu-baseline --no-legs --steps 5 --warmup 2 > $O/bench_$S.json 2> $O/bench_$S.err
done
python - <<'PY' > $O/summary.txt 2>&1
import json
for f in (0, 2, 4):
    try:
        d = json.loads([l for l in open('gpurun_out/r4_share/bench_%d.json' % f).read().splitlines() if l.startswith('{')][-1])
        ks = {k.split('<')[0].split('(')[0]: (round(v['mean_us'], 1), v['launches']) for k, v in d['kernels'].items()}
        print('share', f, 'c2 ms/step %.3f' % d['ms_per_step'], ks)
    except Exception as e:
        print('share', f, 'failed', e, open('gpurun_out/r4_share/bench_%d.err' % f).read()[-400:])
PY
cat $O/summary.txt
