cd $GRAFT_REPO_ROOT
timeout -k 10 250 python scratch/ws_flake.py ${2:-300} > gpurun_out/wsflake_$1_a.log 2>&1 &
timeout -k 10 250 python scratch/ws_flake.py ${2:-300} > gpurun_out/wsflake_$1_b.log 2>&1
wait
grep -h "iter\|done" gpurun_out/wsflake_$1_a.log gpurun_out/wsflake_$1_b.log | head -12
