# FETCH_SIZE / WRITE_SIZE of every kernel at C2 (32^3), at 64^3 Sedov (HBM-resident) and at 64^3 Taylor-Green, default
# dispatch, through the C++ driver (separate --pmc passes, --kernel-trace only).  Writes the per-kernel summaries under
# gpurun_out/pmc_traffic/ and the JSON bench.py quotes (profiles/r5_pmc_traffic.json is written on the GPU box into
# gpurun_out and copied by hand: the repo copy there is not merged back).
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/pmc_traffic; rm -rf $O; mkdir -p $O
run() { # name, driver arguments
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 500 rocprofv3 --kernel-trace --pmc $C -d $O/$1_$C -o p --output-format csv -- ./laghos_amd/laghos $2 -ok 3 -ot 2 -ms 3 -pa > $O/$1_$C.log 2>&1
    python tools/pmc_summary.py $O/$1_$C > $O/$1_$C.txt 2>&1
    find $O/$1_$C -name "*.csv" -delete
  done
}
run c2 "-p 1 -m data/cube01_hex.mesh -rs 4"
run c3 "-p 1 -m data/cube01_hex.mesh -rs 5"
run tg "-p 0 -m data/cube01_hex.mesh -rs 5"
python tools/update_pmc_traffic.py c2=$O/c2_FETCH_SIZE.txt,$O/c2_WRITE_SIZE.txt c3=$O/c3_FETCH_SIZE.txt,$O/c3_WRITE_SIZE.txt tg=$O/tg_FETCH_SIZE.txt,$O/tg_WRITE_SIZE.txt
cp profiles/r6_pmc_traffic.json $O/
head -30 $O/c3_FETCH_SIZE.txt
