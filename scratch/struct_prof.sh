cd $GRAFT_REPO_ROOT
cp iterseg_amd/libitersegx.so /tmp/lib_prod.so
cp scratch/libs/lib_prof.so iterseg_amd/libitersegx.so
ISX_FLOOD_STATS=1 timeout -k 10 300 python scratch/time_structured.py "" ${1:-1} > gpurun_out/struct_prof.log 2>&1
cp /tmp/lib_prod.so iterseg_amd/libitersegx.so
grep -E "flood/group\] cycles|flood/window\] sub-rounds|bursts cut|seed" gpurun_out/struct_prof.log | tail -5
