# Round 3: memory-side counters of K1 in its forms (LGH_VCG_VARIANT=$V for V in $VARIANTS), C2, through the C++ driver
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r3_pmc; mkdir -p $O
APP="./laghos_amd/laghos -p 1 -m data/cube01_hex.mesh -rs ${RS:-4} -ok 3 -ot 2 -ms 3 -pa"
[ -f $O/counters.txt ] || rocprofv3 -L > $O/counters_all.txt 2>&1
grep -o "TC[CP]_[A-Z0-9_]*\|TA_[A-Z0-9_]*\|TD_[A-Z0-9_]*" $O/counters_all.txt | sort -u > $O/counters.txt
for V in $VARIANTS; do
  rm -rf $O/v$V; mkdir -p $O/v$V
  i=0
  for SET in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD"; do
    i=$((i+1))
    LGH_VCG_VARIANT=$V timeout 200 rocprofv3 --kernel-trace --pmc $SET -d $O/v$V/p$i -o p --output-format csv -- $APP > $O/v$V/p$i.log 2>&1
  done
  python tools/pmc_summary.py $O/v$V vcg_apply > $O/v${V}_summary.txt 2>&1
  find $O/v$V -name "*.csv" -delete
  cat $O/v${V}_summary.txt
done
