cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out/r2k
timeout 2400 python -m pytest tests -m gpu -q -k "config5 or rk_integrators or print_dumps" > gpurun_out/r2k/pytest.log 2>&1; echo "pytest rc=$?"
tail -30 gpurun_out/r2k/pytest.log
