# samples the shader clock and the package power while the headline bench runs (is the conv phase power / clock limited?)
cd $GRAFT_REPO_ROOT
( for i in $(seq 1 120); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo; sleep 0.25; done ) > gpurun_out/clock_watch.log 2>&1 &
W=$!
python3 bench.py --steps 3 --warmup 1 --no-extra-legs --no-cpu-baseline > gpurun_out/clock_bench.json 2>/dev/null
kill $W 2>/dev/null; wait $W 2>/dev/null
sort gpurun_out/clock_watch.log | uniq -c | sort -rn | head -12
