# $1 = tag; env passes through; runs two probes at once
cd $GRAFT_REPO_ROOT
timeout -k 10 250 python scratch/stack_flake.py ${2:-150} > gpurun_out/flake_$1_a.log 2>&1 &
timeout -k 10 250 python scratch/stack_flake.py ${2:-150} > gpurun_out/flake_$1_b.log 2>&1
wait
echo "== $1: $(grep -h 'done' gpurun_out/flake_$1_a.log gpurun_out/flake_$1_b.log | tr '\n' ' ')"
