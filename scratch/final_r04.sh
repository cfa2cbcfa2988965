# round-4 final-state artifacts: bench line, kernel stats of the bench, PMC traffic / SQ passes over one 64-chunk launch set
set -e
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; T=${1:-r04a}; O=$R/gpurun_out/$T; mkdir -p $O
cd $R
python3 bench.py > $O/${T}_bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o r04 -- python3 bench.py --steps 1 --warmup 1 --no-extra-legs > $O/${T}_bench512_under_rocprof.json 2> $O/prof_bench.err
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_split-f16_$c -o p -- python3 scratch/one_layer.py 64 > $O/pmc_$c.log 2>&1
done
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F16 --output-format csv -d $O/pmc_sq -o p -- python3 scratch/one_layer.py 64 > $O/pmc_sq.log 2>&1 || true
ls $O
