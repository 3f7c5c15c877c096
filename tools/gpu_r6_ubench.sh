cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out/r6_ubench tools/bin
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/ubench_wgshape.hip -o tools/bin/ubench_wgshape 2> gpurun_out/r6_ubench/build.err && ./tools/bin/ubench_wgshape | tee gpurun_out/r6_ubench/wgshape.txt
tail -3 gpurun_out/r6_ubench/build.err
