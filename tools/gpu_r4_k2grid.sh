# Round 4: node ranges (workgroups) per CU of the bounded-grid K2 with the exact (r, z) accumulators (LGH_K2_GRID), one box
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r4_k2grid; rm -rf $O; mkdir -p $O
for G in 4 2 3 6 8 4; do
  LGH_K2_GRID=$G timeout 100 python bench.py --no-cpu-baseline --no-legs --steps 8 --warmup 3 2>/dev/null | grep '^{' > $O/b.json
  python - <<PY
import json
d = json.loads(open('$O/b.json').read())
print('ranges per CU', $G, 'ms/step %.3f' % d['ms_per_step'], {k.split('<')[0].split('(')[0]: round(v['mean_us'], 1) for k, v in d['kernels'].items() if 'vcg' in k})
PY
done
