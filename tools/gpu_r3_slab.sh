# Round 3: parity + timing of the slab-form K1 (LGH_VCG_VARIANT=4) vs the plane form (2) and the matrix-core form (3)
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r3_slab; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "k1_forms_agree_at_q3q2" > $O/pytest.txt 2>&1
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
for v in $VARIANTS; do
run v${v}_c2 "--steps 20 --warmup 5" "LGH_VCG_VARIANT=$v"
done
for v in $VARIANTS; do
run v${v}_c3 "--workload c3 --steps 4 --warmup 2" "LGH_VCG_VARIANT=$v"
done
for v in $TRACE; do
LGH_VCG_VARIANT=$v LGH_VCG_TRACE=$O/v${v}_c2.trace python bench.py --steps 3 --warmup 1 --no-legs --no-cpu-baseline --no-roofline > $O/tr.json 2> $O/tr.err
python tools/k1_trace_summary.py $O/v${v}_c2.trace mfma
done
for w in $WPSLIST; do
LGH_SLAB_WPS=$w run v4w${w}_c2 "--steps 20 --warmup 5" "LGH_VCG_VARIANT=4"
LGH_SLAB_WPS=$w run v4w${w}_c3 "--workload c3 --steps 4 --warmup 2" "LGH_VCG_VARIANT=4"
LGH_SLAB_WPS=$w LGH_VCG_VARIANT=4 LGH_VCG_TRACE=$O/v4w${w}_c2.trace python bench.py --steps 3 --warmup 1 --no-legs --no-cpu-baseline --no-roofline > $O/tr.json 2> $O/tr.err
python tools/k1_trace_summary.py $O/v4w${w}_c2.trace mfma
done
