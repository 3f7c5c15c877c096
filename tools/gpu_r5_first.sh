# round 5, first box: the whole GPU suite on the build with the advisor fixes, then the bench line in its new form
# (contract line <= 8 KB on stdout, full record in gpurun_out/bench_detail.json)
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r5_first
rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
timeout 900 python bench.py --detail $O/bench_detail.json > $O/bench.json 2> $O/bench.err
tail -c 600 $O/bench.err
wc -c $O/bench.json
