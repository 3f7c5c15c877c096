#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3_trace2; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_pipeline.py -x -q -m gpu -k "slab or k1 or switches or multi_rank or mass_data" 2>&1 | tail -3
LGH_VCG_TRACE=$O/slab_enter.trace timeout 300 python bench.py --steps 3 --warmup 1 --no-legs --no-cpu-baseline > $O/x.json 2> $O/x.err
python tools/k1_trace_summary.py $O/slab_enter.trace
timeout 300 python bench.py --legs c3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],3), {k.split(' ')[0]: round(v['mean_us'],1) for k,v in d['kernels'].items() if 'vcg' in k})
for k,v in d['legs'].items(): print(k, round(v['value'],1), {kk.split(' ')[0]: round(vv['mean_us'],1) for kk,vv in v['kernels'].items() if 'vcg' in kk})"
