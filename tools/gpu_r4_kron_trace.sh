# Round 4: per-workgroup wall-clock stamps and per-phase shader cycles of the Kronecker-form slab K1 at C2
# (LGH_VCG_TRACE / LGH_VCG_TRACE_PHASES, tools/k1_trace_summary.py), and the slab switch tests.
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r4_kron_trace; rm -rf $O; mkdir -p $O
(timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "slab_k1_switches" 2>&1 | tail -5) > $O/tests.log 2>&1
APP="./laghos_amd/laghos -p 1 -m data/cube01_hex.mesh -rs 4 -ok 3 -ot 2 -ms 6 -pa"
LGH_VCG_TRACE=$O/trace_wall.txt timeout 120 $APP > $O/run1.log 2>&1
python tools/k1_trace_summary.py $O/trace_wall.txt > $O/summary_wall.txt 2>&1
LGH_VCG_TRACE=$O/trace_phase.txt LGH_VCG_TRACE_PHASES=1 timeout 120 $APP > $O/run2.log 2>&1
python tools/k1_trace_summary.py $O/trace_phase.txt mfma > $O/summary_phase.txt 2>&1
cat $O/tests.log $O/summary_wall.txt $O/summary_phase.txt
