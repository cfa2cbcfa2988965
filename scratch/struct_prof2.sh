cd $GRAFT_REPO_ROOT
cp iterseg_amd/libitersegx.so /tmp/lib_prod.so
for w in 3 7; do
cp scratch/libs/lib_prof_w$w.so iterseg_amd/libitersegx.so
ISX_FLOOD_STATS=1 timeout -k 10 300 python scratch/time_structured.py "" ${1:-1} > gpurun_out/struct_prof_w$w.log 2>&1
echo "== wave $w"; grep -E "flood/group\] cycles" gpurun_out/struct_prof_w$w.log | tail -1
done
cp /tmp/lib_prod.so iterseg_amd/libitersegx.so
