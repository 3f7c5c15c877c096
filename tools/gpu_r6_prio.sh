# priority of the energy solve's stream: default against highest / lowest (headline and config 5), one box
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r6_prio
rm -rf $O; mkdir -p $O
for i in 1 2; do
  for V in - h l; do
    if [ "$V" = "-" ]; then unset LGH_STREAM2_PRIORITY; else export LGH_STREAM2_PRIORITY=$V; fi
    timeout 900 python bench.py --legs c5,tg --no-cpu-baseline --detail $O/d_${V}_$i.json > /dev/null 2>> $O/err
  done
done
python - <<PY
import json
for V in ("-", "h", "l"):
    for i in (1,2):
        d=json.load(open("$O/d_%s_%d.json"%(V,i)))
        print("LGH_STREAM2_PRIORITY=%s"%V, "c2", round(d["value"],1), round(d["ms_per_step"],3), " c5", round(d["legs"]["c5"]["value"],1), round(d["legs"]["c5"]["ms_per_step"],2), " tg", round(d["legs"]["tg"]["value"],1), round(d["legs"]["tg"]["ms_per_step"],2))
PY
