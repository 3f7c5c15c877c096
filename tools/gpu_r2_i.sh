cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out/r2i
timeout 2400 python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/r2i/pytest.log 2>&1; echo "pytest rc=$?"
tail -30 gpurun_out/r2i/pytest.log
timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/r2i/bench.json 2> gpurun_out/r2i/bench.err; echo "bench rc=$?"
tail -c 3000 gpurun_out/r2i/bench.json
