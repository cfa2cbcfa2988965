# Round-5 evidence run on the GPU box (one gpurun call): bench line, kernel stats of the headline, PMC traffic passes of the
# wide-conv kernels (both arithmetic modes), one SQ pass, kernel stats of the training step (fp32 / bf16).  $1 = tag (r05a ...)
set -e
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; T=${1:-r05a}; O=$R/gpurun_out/$T; mkdir -p $O
cd $R
echo "== bench line"
python3 bench.py > $O/${T}_bench.json 2> $O/bench.err
echo "== kernel stats of the headline (warm-up + 1 step)"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o r05 -- python3 bench.py --steps 1 --warmup 1 --no-extra-legs > $O/${T}_bench512_under_rocprof.json 2> $O/prof_bench.err
for prec in split-f16 fp32; do
  for c in FETCH_SIZE WRITE_SIZE; do
    echo "== pmc $c $prec"
    ISX_CONV_PRECISION=$prec rocprofv3 --pmc $c --output-format csv -d $O/pmc_${prec}_$c -o p -- python3 scratch/one_layer.py 64 > $O/pmc_${prec}_$c.log 2>&1
  done
done
echo "== pmc SQ (split-f16)"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d $O/pmc_sq -o p -- python3 scratch/one_layer.py 64 > $O/pmc_sq.log 2>&1 || echo "sq pass failed"
for prec in fp32 bf16; do
  echo "== kernel stats: training step $prec"
  ISX_TRAIN_PRECISION=$prec rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train_$prec -o r05 -- python3 scratch/bench_train.py > $O/train_$prec.log 2>&1; tail -2 $O/train_$prec.log
done
find $O -name "*.csv" | head -60
echo "== config 5 (1024^3): f16 and split-f16 lines"
for prec in f16 split-f16; do
  timeout -k 10 400 python3 bench.py --steps 1 --warmup 1 --size 1024 --precision $prec --no-extra-legs --no-cpu-baseline > $O/${T}_bench_1024_$prec.json 2> $O/bench_1024_$prec.err || echo "1024 $prec failed"
done
