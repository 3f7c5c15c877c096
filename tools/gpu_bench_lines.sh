# the two committed bench lines (under rocprofv3 --kernel-trace --stats, and plain) without the PMC passes
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/prof_r2
mkdir -p $O; rm -rf $O/stats
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python bench.py --steps 5 --warmup 2 --no-legs --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
find $O -name "*kernel_trace.csv" -delete
tail -c 300 $O/bench.json
