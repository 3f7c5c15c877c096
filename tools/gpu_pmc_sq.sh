cd /root/repo
export TMPDIR=/tmp
APP="./laghos_amd/laghos -p 1 -m data/cube01_hex.mesh -rs 4 -ok 3 -ot 2 -ms 3"
rm -rf gpurun_out/pmc_sq; mkdir -p gpurun_out/pmc_sq
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d gpurun_out/pmc_sq/a -o a --output-format csv -- $APP > gpurun_out/pmc_sq/a.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVES -d gpurun_out/pmc_sq/b -o b --output-format csv -- $APP > gpurun_out/pmc_sq/b.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc_sq "qpoint_kernel<3, 4, 6, 3, 1, 6, 0>" "vcg_apply_plane" "vcg_update_p_k" > gpurun_out/pmc_sq/summary.txt; find gpurun_out/pmc_sq -name "*counter_collection.csv" -delete; find gpurun_out/pmc_sq -name "*kernel_trace.csv" -delete; cat gpurun_out/pmc_sq/summary.txt
