# two / four / eight ranks (processes sharing the one GPU) over the cross-process loop-back transport, 32^3 zones per rank (config 4's
# rank size): energy CG in lockstep / after the velocity CG on one communicator, and beside it on the second channel
cd /root/repo
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r6_lockstep_shm
rm -rf $O; mkdir -p $O
for n in 2 8; do
for mode in "LGH_COMM2=0 LGH_ENERGY_LOCKSTEP=1" "LGH_COMM2=0 LGH_ENERGY_LOCKSTEP=0" "LGH_COMM2=1"; do
tag=$(echo $mode | tr ' =' '__')
env $mode timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500 + n)) bench.py --gpus $n --transport shm --block 32 --steps 8 --warmup 3 --detail $O/d_${n}_$tag.json > $O/b_${n}_$tag.json 2> $O/e_${n}_$tag.err
python - <<PY
import json
try:
    d=json.load(open("$O/d_${n}_$tag.json"))
    print("$n ranks", "$mode", "value", round(d["value"],1), "ms/step", round(d["ms_per_step"],3), d["comm"].get("energy_lockstep"), "e", d["config"]["e_norm"])
except Exception as e:
    print("$n ranks", "$mode", "failed", e)
PY
done
done
