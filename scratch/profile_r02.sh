# Round-2 evidence run on the GPU box (one gpurun call): kernel stats of the default bench, PMC traffic passes of the
# wide-conv kernels (both arithmetic modes), kernel stats of the training step and of the config-2 flood.
set -e
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2p; mkdir -p $O
cd $R
echo "== bench (default)"; python3 bench.py > $O/r02a_bench512.json 2> $O/bench.err; tail -c 300 $O/r02a_bench512.json; echo
echo "== kernel stats of the headline (warm-up + 1 step)"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o r02a -- python3 bench.py --steps 1 --warmup 1 --no-extra-legs > $O/r02a_bench512_under_rocprof.json 2> $O/prof_bench.err
for prec in split-f16 fp32; do
  for c in FETCH_SIZE WRITE_SIZE; do
    echo "== pmc $c $prec"
    ISX_CONV_PRECISION=$prec rocprofv3 --pmc $c --output-format csv -d $O/pmc_${prec}_$c -o p -- python3 scratch/one_layer.py 64 > $O/pmc_${prec}_$c.log 2>&1
  done
done
echo "== kernel stats: training step"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train -o r02a -- python3 scratch/bench_train.py > $O/train.log 2>&1; tail -2 $O/train.log
echo "== kernel stats: config-2 watershed"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c2 -o r02a -- python3 scratch/bench_c2.py 64 512 512 > $O/c2.log 2>&1; tail -2 $O/c2.log
find $O -name "*.csv" | head -40
