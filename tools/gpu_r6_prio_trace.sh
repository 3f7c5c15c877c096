# kernel-trace timelines of the headline with the energy stream at default / lowest priority (same box)
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r6_prio_trace
rm -rf $O; mkdir -p $O
for V in d l; do
LGH_STREAM2_PRIORITY=$V timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O/s$V -o steps -- python bench.py --no-legs --no-cpu-baseline --no-roofline --steps 30 --warmup 5 --detail $O/detail_$V.json > $O/steps_$V.json 2> $O/steps_$V.err
python tools/step_timeline.py $O/s$V > $O/timeline_$V.txt 2>&1
rm -rf $O/s$V
echo "priority $V:"; head -1 $O/timeline_$V.txt; grep -A 12 "^idle per step" $O/timeline_$V.txt
LGH_STREAM2_PRIORITY=$V timeout 600 python bench.py --no-legs --no-cpu-baseline --no-roofline --steps 30 --warmup 5 --detail $O/plain_$V.json 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('unprofiled', d['ms_per_step'])"
done
