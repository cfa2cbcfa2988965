# same-box A/B of two builds (scratch/libs/lib_A.so, lib_B.so): launch-set time (bench_unet 64), alternating, $1 repetitions
cd $GRAFT_REPO_ROOT
cp iterseg_amd/libitersegx.so /tmp/lib_prod.so
for rep in $(seq 1 ${1:-2}); do for v in A B; do cp scratch/libs/lib_$v.so iterseg_amd/libitersegx.so; echo "#### $v (rep $rep): $(python scratch/bench_unet.py 64 2>&1 | tail -1)"; done; done
cp /tmp/lib_prod.so iterseg_amd/libitersegx.so
