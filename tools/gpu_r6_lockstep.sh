# lockstep energy CG: the multi-rank tests on one GPU (in-process and cross-process loop-back) + the three N-rank legs, twice
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r6_lockstep
rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_multiproc.py -x -q -k "multi_rank or cross_process or multi or word" > $O/pytest.log 2>&1; tail -8 $O/pytest.log
for i in 1 2; do
timeout 600 python bench.py --no-cpu-baseline --legs c2multi,c2multi1c,c2multi1cseq --detail $O/d_$i.json > $O/b_$i.json 2>> $O/bench.err
python - <<PY
import json
d=json.load(open("$O/d_$i.json"))
print("headline", d["value"], d["ms_per_step"])
for n,g in d["legs"].items():
    print(n, g.get("value"), g.get("ms_per_step"), g.get("ms_per_step_minus_single_rank_path"), g.get("energy_lockstep"), g.get("error"))
PY
done
# A/B: the lockstep energy kernels beside K1 / K2 on the second stream (four events per iteration)
for i in 1 2; do
LGH_LOCKSTEP_STREAM2=1 timeout 600 python bench.py --no-cpu-baseline --legs c2multi1c --detail $O/d1s_$i.json > $O/b1s_$i.json 2>> $O/bench.err
python - <<PY
import json
d=json.load(open("$O/d1s_$i.json"))
for n,g in d["legs"].items():
    print("second stream:", n, g.get("value"), g.get("ms_per_step"), g.get("ms_per_step_minus_single_rank_path"), g.get("energy_lockstep"), g.get("error"))
PY
done
tail -3 $O/bench.err
