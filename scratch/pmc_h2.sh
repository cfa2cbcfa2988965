# SQ counter passes (one rocprofv3 run per counter group) over one 64-chunk launch set, split-f16 kernels
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU" \
           "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" \
           "SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1)); rm -rf /tmp/pmc_$i
  timeout -k 10 400 rocprofv3 --pmc $grp --output-format csv -d /tmp/pmc_$i -o p -- python3 $GRAFT_REPO_ROOT/scratch/one_layer.py 64 > /tmp/pmc_$i.log 2>&1 || { echo "pass $i failed"; tail -3 /tmp/pmc_$i.log; continue; }
  python3 $GRAFT_REPO_ROOT/scratch/pmc_sum.py "/tmp/pmc_$i/p_counter_collection.csv"
done
