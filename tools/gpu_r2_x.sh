cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r2x; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -x -k "multi_rank or config4 or err_option" > $O/pytest.log 2>&1; echo "pytest rc=$?"
tail -15 $O/pytest.log
