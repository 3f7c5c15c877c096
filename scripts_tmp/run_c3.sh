for v in 1 0; do
echo "== perm=$v"
LGH_VCG_PERM=$v timeout 600 ./laghos_amd/laghos -p 1 -m data/cube01_hex.mesh -rs 5 -ok 3 -ot 2 -ms 6 -f -pa 2>&1 | grep -E "CG \(H1\)|Forces|UpdateQuad|Major kernels total rate|FOM|iterations" | head -12
done
