# Round-2 final-state evidence (one gpurun call): the default bench line with all legs, kernel stats of the headline,
# kernel stats of the config-2 watershed.  (PMC traffic of the conv kernels: profile_r02.sh; they did not change since.)
set -e
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2f; mkdir -p $O
cd $R
echo "== bench (default)"; python3 bench.py > $O/r02f_bench512.json 2> $O/bench.err; tail -c 400 $O/r02f_bench512.json; echo
echo "== kernel stats of the headline (warm-up + 1 step)"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o r02f -- python3 bench.py --steps 1 --warmup 1 --no-extra-legs > $O/r02f_bench512_under_rocprof.json 2> $O/prof_bench.err
echo "== kernel stats: config-2 watershed"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c2 -o r02f -- python3 scratch/bench_c2.py 64 512 512 > $O/c2.log 2>&1; tail -2 $O/c2.log
find $O -name "*stats*.csv" | head
