cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/k1; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -k "bit_identical or checks_table or config1 or config2 or test_hydro_mult or test_cg_h1 or full_size or forms_agree or multi_rank" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
for v in 1 0 1; do
LGH_B_SYM=$v timeout 600 python bench.py --steps 20 --warmup 5 --no-legs --no-cpu-baseline > $O/b_$v.json 2> $O/b_$v.err; echo "sym=$v rc=$?"
python - <<P
import json
d=json.loads([l for l in open("$O/b_$v.json") if l.startswith("{")][-1])
k1=[v for n,v in d["kernels"].items() if n.startswith("vcg_apply")][0]
k2=[v for n,v in d["kernels"].items() if n.startswith("vcg_update")][0]
print("sym=$v", round(d["value"],1), round(d["ms_per_step"],3), "K1", round(k1["mean_us"],1), "K2", round(k2["mean_us"],1), repr(d["config"]["e_norm"]))
P
done
LGH_B_SYM=1 timeout 600 python bench.py --workload c3 --steps 4 --warmup 2 --no-legs --no-cpu-baseline > $O/c3.json 2> $O/c3.err
python - <<P
import json
d=json.loads([l for l in open("$O/c3.json") if l.startswith("{")][-1])
k1=[v for n,v in d["kernels"].items() if n.startswith("vcg_apply")][0]
print("c3", round(d["value"],1), round(d["ms_per_step"],3), "K1", round(k1["mean_us"],1))
P
