cd $GRAFT_REPO_ROOT
cp iterseg_amd/libitersegx.so /tmp/lib_prod.so
cp scratch/libs/lib_prof.so iterseg_amd/libitersegx.so
ISX_FLOOD_STATS=1 timeout -k 10 300 python scratch/bench_c2.py > gpurun_out/c2_prof.log 2>&1
cp /tmp/lib_prod.so iterseg_amd/libitersegx.so
tail -12 gpurun_out/c2_prof.log
