cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out/r2j
timeout 2400 python -m pytest tests -m gpu -q --durations=10 > gpurun_out/r2j/pytest.log 2>&1; echo "pytest rc=$?"
tail -40 gpurun_out/r2j/pytest.log
