# phase timers of the work-group flood on config 2 (scratch/libs/lib_prof.so: bash scratch/build_variant.sh prof "-DISX_FLOOD_PROF" ws.hip)
cd $GRAFT_REPO_ROOT
ISX_LIB=$GRAFT_REPO_ROOT/scratch/libs/lib_prof.so ISX_FLOOD_STATS=1 timeout -k 10 300 python scratch/bench_c2.py > gpurun_out/c2_prof.log 2>&1
tail -12 gpurun_out/c2_prof.log
