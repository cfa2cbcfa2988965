set -e
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3f; mkdir -p $O
cd $R
python3 bench.py > $O/r03f_bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o r03 -- python3 bench.py --steps 1 --warmup 1 --no-extra-legs > $O/r03f_bench512_under_rocprof.json 2> $O/prof_bench.err
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_split-f16_$c -o p -- python3 scratch/one_layer.py 64 > $O/pmc_$c.log 2>&1
done
