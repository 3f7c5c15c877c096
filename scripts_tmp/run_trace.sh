mkdir -p gpurun_out/trace
LGH_VCG_TRACE=gpurun_out/trace/k1.txt timeout 300 ./laghos_amd/laghos -p 1 -m data/cube01_hex.mesh -rs 4 -ok 3 -ot 2 -ms 3 -pa > gpurun_out/trace/log.txt 2>&1
tail -3 gpurun_out/trace/log.txt
wc -l gpurun_out/trace/k1.txt
