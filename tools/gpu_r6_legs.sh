cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r6_legs
rm -rf $O; mkdir -p $O
timeout 600 python bench.py --no-legs --detail $O/d_cpu.json > $O/b_cpu.json 2> $O/e_cpu.err
echo "cpu-baseline-only rc=$?"; tail -3 $O/e_cpu.err
timeout 900 python bench.py --no-cpu-baseline --detail $O/d_legs.json > $O/b_legs.json 2> $O/e_legs.err
echo "all-legs rc=$?"; tail -3 $O/e_legs.err
