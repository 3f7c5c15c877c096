cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r3_trace; mkdir -p $O
LGH_VCG_VARIANT=3 LGH_VCG_TRACE=$O/mfma_c2.trace python bench.py --steps 3 --warmup 1 --no-legs --no-cpu-baseline --no-roofline > $O/mfma_c2.json 2> $O/mfma_c2.err
python tools/k1_trace_summary.py $O/mfma_c2.trace mfma
LGH_VCG_VARIANT=3 LGH_VCG_TRACE=$O/mfma_c3.trace python bench.py --workload c3 --steps 2 --warmup 1 --no-legs --no-cpu-baseline --no-roofline > $O/mfma_c3.json 2> $O/mfma_c3.err
python tools/k1_trace_summary.py $O/mfma_c3.trace mfma
