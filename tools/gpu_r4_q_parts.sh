# Round 4: where the vector instructions of the row-form quadrature update go: SQ_INSTS_VALU / SQ_WAVES of qrows_kernel
# with parts of the point body switched off by the library's own switches (no special builds):
#   default | eigen-decomposition never (LGH_Q_TINY_GRAD=1e300) | always (=-1) | no fused force products | Taylor-Green (visc off)
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r4_q_parts; rm -rf $O; mkdir -p $O
run() { # tag, app args, env...
  T=$1; shift; A=$1; shift
  env "$@" timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_LDS SQ_INSTS_SALU -d $O/$T -o p --output-format csv -- ./laghos_amd/laghos $A -ok 3 -ot 2 -ms 3 -pa > $O/$T.log 2>&1
  echo "== $T" >> $O/summary.txt; python tools/pmc_summary.py $O/$T qrows_kernel >> $O/summary.txt
}
S="-p 1 -m data/cube01_hex.mesh -rs 4"
run default "$S" LGH_X=1
run eig_never "$S" LGH_Q_TINY_GRAD=1e300
run eig_always "$S" LGH_Q_TINY_GRAD=-1
run no_forces "$S" LGH_FUSED_FTV=0 LGH_FUSED_F1=0
run tg "-p 0 -m data/cube01_hex.mesh -rs 4" LGH_X=1
run tg_no_forces "-p 0 -m data/cube01_hex.mesh -rs 4" LGH_FUSED_FTV=0 LGH_FUSED_F1=0
find $O -name "*.csv" -delete
cat $O/summary.txt
