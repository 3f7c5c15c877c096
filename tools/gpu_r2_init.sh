cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r2init; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -k "bit_identical or checks_table or config1 or config2 or test_hydro_mult" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-legs --no-cpu-baseline > $O/b.json 2> $O/b.err; echo "rc=$?"
python - <<P
import json
d=json.loads([l for l in open("$O/b.json") if l.startswith("{")][-1])
k1=[v for n,v in d["kernels"].items() if n.startswith("vcg_apply")][0]
k2=[v for n,v in d["kernels"].items() if n.startswith("vcg_update")][0]
print(round(d["value"],1), round(d["ms_per_step"],3), "K1", round(k1["mean_us"],1), "K2", round(k2["mean_us"],1), repr(d["config"]["e_norm"]))
P
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -o b -- python bench.py --steps 5 --warmup 2 --no-legs --no-cpu-baseline > $O/prof.json 2> $O/prof.err
grep "init_force\|vcg_update\|vcg_apply" $O/st/b_kernel_stats.csv | cut -c1-140
find $O -name "*kernel_trace.csv" -delete
