# Round 4: SQ and TCC counters of the quadrature-update kernel, row form (qrows_kernel) against point form (qpoint_kernel),
# C2 mesh through the C++ driver.  Separate --pmc passes, --kernel-trace only.
cd /root/repo
export TMPDIR=/tmp
APP="./laghos_amd/laghos -p 1 -m data/cube01_hex.mesh -rs ${RS:-4} -ok 3 -ot 2 -ms 3 -pa"
O=gpurun_out/r4_q_pmc${TAG:-}; rm -rf $O; mkdir -p $O
for F in ${FORMS:-1 0}; do
  export LGH_Q_FORM=$F
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/a$F -o a --output-format csv -- $APP > $O/a$F.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVES -d $O/b$F -o b --output-format csv -- $APP > $O/b$F.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/f$F -o f --output-format csv -- $APP > $O/f$F.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/w$F -o w --output-format csv -- $APP > $O/w$F.log 2>&1
  for P in a b f w; do python tools/pmc_summary.py $O/$P$F qrows_kernel "qpoint_kernel<3, 4, 6, 3, 1, 6, 0" >> $O/summary_form$F.txt; done
done
find $O -name "*.csv" -delete
cat $O/summary_form*.txt
