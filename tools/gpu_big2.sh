cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/big; mkdir -p $O
run() { env $1 timeout 900 ./laghos_amd/laghos -p 1 -m data/cube01_hex.mesh -rs 6 -ok 3 -ot 2 -ms 4 -pa -f > $O/x.log 2>&1; echo "$1 rc=$? $(grep 'CG (H1) total time' $O/x.log) FOM0 $(grep '^|      1' $O/x.log | cut -d'|' -f8)"; }
run LGH_K2_GRID=4
run LGH_K2_GRID=112
run LGH_K2_X=default
run LGH_K2_GRID=56
run LGH_K2_X=default
