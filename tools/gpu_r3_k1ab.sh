# Round 3: parity of the K1 forms at Q3Q2 and A/B of the matrix-core K1 (LGH_VCG_VARIANT=3) against the plane form,
# C2 and 64^3, on one box.  Output: gpurun_out/r3_k1ab/
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r3_k1ab; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "k1_forms_agree_at_q3q2 or cg_h1" > $O/pytest.txt 2>&1
tail -5 $O/pytest.txt
run() {
env $3 python bench.py $2 --no-legs --no-cpu-baseline > $O/$1.json 2> $O/$1.err
python - <<P
import json
try:
    d=json.loads([l for l in open("$O/$1.json") if l.startswith("{")][-1])
    k1=[v for n,v in d["kernels"].items() if n.startswith("vcg_apply")][0]
    k2=[v for n,v in d["kernels"].items() if n.startswith("vcg_update")][0]
    q=[v for n,v in d["kernels"].items() if n.startswith("qpoint")][0]
    print("$1", round(d["value"],1), round(d["ms_per_step"],3), "K1", round(k1["mean_us"],1), "K2", round(k2["mean_us"],1), "Q", round(q["mean_us"],1), "e_norm", d["config"]["e_norm"])
except Exception as ex:
    print("$1 FAILED", ex)
P
}
for rep in 1 2; do
run plane_c2_$rep "--steps 20 --warmup 5" "LGH_VCG_VARIANT=2"
run mfma_c2_$rep "--steps 20 --warmup 5" "LGH_VCG_VARIANT=3"
done
run plane_c3 "--workload c3 --steps 4 --warmup 2" "LGH_VCG_VARIANT=2"
run mfma_c3 "--workload c3 --steps 4 --warmup 2" "LGH_VCG_VARIANT=3"
rm -rf $O/stats
LGH_VCG_VARIANT=3 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python bench.py --steps 5 --warmup 2 --no-legs --no-cpu-baseline > $O/mfma_under_rocprof.json 2> $O/mfma_under_rocprof.err
find $O -name "*kernel_trace.csv" -delete
find $O -name "*kernel_stats.csv" | head -1 | xargs head -12
