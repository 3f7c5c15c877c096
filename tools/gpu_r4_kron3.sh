cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r4_kron3; rm -rf $O; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_k1.py tests/test_gpu_kernels.py tests/test_gpu_pipeline.py -x -q -k "k1 or slab or mass_data or lockstep or cg_h1 or hydro_mult or multi_rank" 2>&1 | tail -5) > $O/tests.log 2>&1
APP="./laghos_amd/laghos -p 1 -m data/cube01_hex.mesh -rs 4 -ok 3 -ot 2 -ms 6 -pa"
LGH_VCG_TRACE=$O/trace_wall.txt timeout 120 $APP > $O/run1.log 2>&1
python tools/k1_trace_summary.py $O/trace_wall.txt > $O/summary_wall.txt 2>&1
timeout 300 python bench.py --no-cpu-baseline --no-legs --steps 20 --warmup 5 2>/dev/null | grep '^{' > $O/bench.json
python - <<'PY' > $O/summary.txt 2>&1
import json
d = json.loads(open('gpurun_out/r4_kron3/bench.json').read())
k1 = [v for k, v in d['kernels'].items() if k.startswith('vcg_apply')][0]
print('c2 %.3f ms/step value %.1f K1 %.1f us' % (d['ms_per_step'], d['value'], k1['mean_us']))
PY
cat $O/tests.log $O/summary_wall.txt $O/summary.txt
