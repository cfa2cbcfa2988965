set -e
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_l2; mkdir -p $O
cd $R
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $O/tcc -o p -- python3 scratch/one_layer.py 64 > $O/tcc.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES --output-format csv -d $O/sq -o p -- python3 scratch/one_layer.py 64 > $O/sq.log 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $O/tcp -o p -- python3 scratch/one_layer.py 64 > $O/tcp.log 2>&1 || echo "tcp pass failed"
ls $O/*
