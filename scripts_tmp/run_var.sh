set -x
mkdir -p gpurun_out/var
for v in 2 1 0; do
  LGH_VCG_VARIANT=$v timeout 300 python -m pytest tests/test_gpu_pipeline.py -m gpu -x -q -k "golden or oracle or Q3Q2 or q3q2" 2>&1 | tail -3
  LGH_VCG_VARIANT=$v timeout 300 python bench.py --steps 5 --warmup 2 > gpurun_out/var/bench_v$v.json 2> gpurun_out/var/bench_v$v.err
  tail -c 1500 gpurun_out/var/bench_v$v.json
done
