# the same with the reporting wave chosen at build time (lib_prof_w3.so / lib_prof_w7.so: -DISX_FLOOD_PROF -DISX_FLOOD_PROF_WAVE=3 / 7)
cd $GRAFT_REPO_ROOT
for w in 3 7; do
ISX_LIB=$GRAFT_REPO_ROOT/scratch/libs/lib_prof_w$w.so ISX_FLOOD_STATS=1 timeout -k 10 300 python scratch/time_structured.py "" ${1:-1} > gpurun_out/struct_prof_w$w.log 2>&1
echo "== wave $w"; grep -E "flood/group\] cycles" gpurun_out/struct_prof_w$w.log | tail -1
done
