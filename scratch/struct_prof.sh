# phase timers of the work-group flood on a structured 512^3 volume (seed $1); the profiling build is selected with ISX_LIB, never
# copied over the product library
cd $GRAFT_REPO_ROOT
ISX_LIB=$GRAFT_REPO_ROOT/scratch/libs/lib_prof.so ISX_FLOOD_STATS=1 timeout -k 10 300 python scratch/time_structured.py "" ${1:-1} > gpurun_out/struct_prof.log 2>&1
grep -E "flood/group\] cycles|flood/window\] sub-rounds|bursts cut|seed" gpurun_out/struct_prof.log | tail -5
