cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out/r2u
APP="./laghos_amd/laghos -p 3 -m data/box01_hex.mesh -rs 4 -ok 5 -ot 4 -ms 2 -pa"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2u/c5 -o c5 -- $APP > gpurun_out/r2u/c5.log 2>&1; echo rc=$?
APP="./laghos_amd/laghos -p 1 -m data/cube01_hex.mesh -rs 4 -ok 4 -ot 3 -ms 5 -pa"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2u/q4 -o q4 -- $APP > gpurun_out/r2u/q4.log 2>&1; echo rc=$?
find gpurun_out/r2u -name "*kernel_stats.csv" | while read f; do echo $f; head -12 $f | cut -c1-200; done
find gpurun_out/r2u -name "*kernel_trace.csv" -delete
find gpurun_out/r2u -name "*.db" -delete
