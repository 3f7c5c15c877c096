cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out/r2q
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r2q/pytest.log 2>&1; echo "pytest rc=$?"
tail -8 gpurun_out/r2q/pytest.log
