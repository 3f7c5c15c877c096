cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r6_num_tests
rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_general_numbering.py -x -q --durations=8 > $O/pytest.log 2>&1; tail -16 $O/pytest.log
