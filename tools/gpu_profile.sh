cd /root/repo
export TMPDIR=/tmp
rm -rf gpurun_out/prof_r1; mkdir -p gpurun_out/prof_r1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r1/stats -o bench -- python bench.py --steps 5 --warmup 2 > gpurun_out/prof_r1/bench_under_rocprof.json 2> gpurun_out/prof_r1/bench_under_rocprof.err
timeout 600 python bench.py --steps 5 --warmup 2 > gpurun_out/prof_r1/bench.json 2> gpurun_out/prof_r1/bench.err
APP="./laghos_amd/laghos -p 1 -m data/cube01_hex.mesh -rs 4 -ok 3 -ot 2 -ms 4 -pa"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/prof_r1/pmc_f -o f --output-format csv -- $APP > gpurun_out/prof_r1/pmc_f.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/prof_r1/pmc_w -o w --output-format csv -- $APP > gpurun_out/prof_r1/pmc_w.log 2>&1
python tools/pmc_summary.py gpurun_out/prof_r1/pmc_f
python tools/pmc_summary.py gpurun_out/prof_r1/pmc_w
find gpurun_out/prof_r1 -name "*stats*" | head
tail -c 600 gpurun_out/prof_r1/bench.json
