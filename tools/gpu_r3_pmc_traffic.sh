# Round 3: FETCH_SIZE / WRITE_SIZE of every kernel at C2 (32^3) and at 64^3 (HBM-resident), default dispatch, through
# the C++ driver; writes profiles-ready summaries under gpurun_out/r3_pmc_traffic/ and refreshes profiles/r3_pmc_traffic.json
# (the JSON is written on the GPU box into gpurun_out and copied by hand: the repo copy there is not merged back)
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r3_pmc_traffic; rm -rf $O; mkdir -p $O
for RS in 4 5; do
  APP="./laghos_amd/laghos -p 1 -m data/cube01_hex.mesh -rs $RS -ok 3 -ot 2 -ms 3 -pa"
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 400 rocprofv3 --kernel-trace --pmc $C -d $O/rs${RS}_$C -o p --output-format csv -- $APP > $O/rs${RS}_$C.log 2>&1
    python tools/pmc_summary.py $O/rs${RS}_$C > $O/rs${RS}_$C.txt 2>&1
    find $O/rs${RS}_$C -name "*.csv" -delete
  done
done
python tools/update_pmc_traffic.py c2=$O/rs4_FETCH_SIZE.txt,$O/rs4_WRITE_SIZE.txt c3=$O/rs5_FETCH_SIZE.txt,$O/rs5_WRITE_SIZE.txt
cp profiles/r3_pmc_traffic.json $O/
head -30 $O/rs5_FETCH_SIZE.txt
