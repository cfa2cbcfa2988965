# Round-3 evidence run on the GPU box (one gpurun call): kernel stats of the headline, PMC traffic passes of the wide-conv
# kernels (both arithmetic modes), kernel stats of the training step (fp32 / bf16) and of the config-2 and structured floods.
set -e
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3r; mkdir -p $O
cd $R
echo "== kernel stats of the headline (warm-up + 1 step)"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o r03 -- python3 bench.py --steps 1 --warmup 1 --no-extra-legs > $O/r03_bench512_under_rocprof.json 2> $O/prof_bench.err
for prec in split-f16 fp32; do
  for c in FETCH_SIZE WRITE_SIZE; do
    echo "== pmc $c $prec"
    ISX_CONV_PRECISION=$prec rocprofv3 --pmc $c --output-format csv -d $O/pmc_${prec}_$c -o p -- python3 scratch/one_layer.py 64 > $O/pmc_${prec}_$c.log 2>&1
  done
done
for prec in fp32 bf16; do
  echo "== kernel stats: training step $prec"
  ISX_TRAIN_PRECISION=$prec rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train_$prec -o r03 -- python3 scratch/bench_train.py > $O/train_$prec.log 2>&1; tail -2 $O/train_$prec.log
done
echo "== kernel stats: config-2 watershed"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c2 -o r03 -- python3 scratch/bench_c2.py 64 512 512 > $O/c2.log 2>&1; tail -2 $O/c2.log
echo "== kernel stats: structured 512^3 watershed (seed 3)"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_struct -o r03 -- python3 scratch/time_structured.py "" 3 > $O/struct.log 2>&1; tail -1 $O/struct.log
find $O -name "*.csv" | head -60
