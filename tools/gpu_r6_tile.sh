# experiment: (y,z)-tiled zone order for the slab K1's static schedule at 64^3 (LGH_ORDER_TILE)
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r6_tile
rm -rf $O; mkdir -p $O
for T in 0 640 1280 320; do
  if [ $T = 0 ]; then unset LGH_ORDER_TILE; else export LGH_ORDER_TILE=$T; fi
  timeout 900 python bench.py --workload c3 --steps 4 --warmup 2 --no-legs --no-cpu-baseline --detail $O/detail_$T.json > $O/bench_$T.json 2>> $O/bench.err
done
python - <<PY
import json
for T in (0,640,1280,320):
    d=json.load(open("$O/detail_%d.json"%T))
    k=d["kernels"]
    print(T, round(d["value"],1), round(d["ms_per_step"],2), {n.split(" ")[0].split("<")[0]: round(v["mean_us"],1) for n,v in k.items()})
PY
