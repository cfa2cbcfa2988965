# same-box A/B of two builds of libitersegx.so (scratch/h2/lib_A.so, lib_B.so): per-layer conv times, twice each, interleaved
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for v in A B; do cp scratch/h2/lib_$v.so iterseg_amd/libitersegx.so; echo "#### $v (rep $rep)"; H2_ENVS="ISX_H2_MT2=1" bash scratch/h2run.sh 2>&1 | grep -E "total|256-> 256 L3|64->  64 L1|128-> 128 L2|64->   5"; done; done
