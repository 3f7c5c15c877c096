# the whole GPU suite + smoke (round 6 check points)
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r6_suite
rm -rf $O; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1; tail -8 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
