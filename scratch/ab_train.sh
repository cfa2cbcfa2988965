set -e
cd $GRAFT_REPO_ROOT; export PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/abt; mkdir -p $O
for wgs in 4096 2048 3072 1024; do
  for prec in fp32 bf16; do
    echo "wgs=$wgs prec=$prec" >> $O/log.txt
    ISX_WGRAD_WGS=$wgs ISX_TRAIN_PRECISION=$prec timeout -k 10 120 python3 scratch/bench_train.py 2>&1 | grep "ms" >> $O/log.txt
  done
done
