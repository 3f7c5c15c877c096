# Config 5 (Q5Q4): SQ / memory-side counters of the kernels of its energy CG and of the plane-form K1 (one step of the C++ driver)
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r3_pmc_c5; rm -rf $O; mkdir -p $O
APP="./laghos_amd/laghos -p 3 -m data/box01_hex.mesh -rs 4 -ok 5 -ot 4 -ms 1 -pa"
i=0
for SET in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR" "TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $SET -d $O/p$i -o p --output-format csv -- $APP > $O/p$i.log 2>&1
done
python tools/pmc_summary.py $O mass_apply_l2_plane cg_update_k vcg_apply_plane_ho qpoint_kernel > $O/summary.txt 2>&1
find $O -name "*.csv" -delete
cat $O/summary.txt
