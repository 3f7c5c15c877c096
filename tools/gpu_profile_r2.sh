# round-2 profile set: the two PMC traffic passes (-> profiles/pmc_traffic.json), kernel stats under rocprofv3,
# the plain bench line
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/prof_r2
rm -rf $O; mkdir -p $O
APP="./laghos_amd/laghos -p 1 -m data/cube01_hex.mesh -rs 4 -ok 3 -ot 2 -ms 4 -pa"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_f -o f --output-format csv -- $APP > $O/pmc_f.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_w -o w --output-format csv -- $APP > $O/pmc_w.log 2>&1
python tools/pmc_summary.py $O/pmc_f > $O/pmc_f_summary.txt 2>&1
python tools/pmc_summary.py $O/pmc_w > $O/pmc_w_summary.txt 2>&1
python tools/update_pmc_traffic.py $O/pmc_f_summary.txt $O/pmc_w_summary.txt && cp profiles/pmc_traffic.json $O/pmc_traffic.json
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python bench.py --steps 5 --warmup 2 --no-legs --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
find $O -name "*kernel_trace.csv" -delete
find $O -name "*counter_collection.csv" -delete
find $O -name "*kernel_stats*" | head -3
tail -c 300 $O/bench.json
head -12 $O/pmc_f_summary.txt
head -8 $O/pmc_w_summary.txt
