cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/full; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x --durations=8 > $O/pytest.log 2>&1; echo "pytest rc=$?"
tail -16 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
