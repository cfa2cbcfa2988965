# SQ counter passes over the config-2 watershed (one giant component): dynamic instruction mix of flood_window_kernel
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" \
           "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_INSTS_BRANCH"; do
  i=$((i+1)); rm -rf /tmp/pf_$i
  timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d /tmp/pf_$i -o p -- python3 $GRAFT_REPO_ROOT/scratch/bench_c2.py 64 512 512 > /tmp/pf_$i.log 2>&1 || { echo "pass $i failed"; tail -3 /tmp/pf_$i.log; continue; }
  python3 - "$(find /tmp/pf_$i -name '*counter_collection.csv' | head -1)" <<'PY'
import csv, sys, collections
rows=list(csv.DictReader(open(sys.argv[1])))
acc=collections.defaultdict(float); n=0
for r in rows:
    if 'flood_window_kernel' in r['Kernel_Name']:
        acc[r['Counter_Name']]+=float(r['Counter_Value']); n+=1
print({k: f"{v:.4g}" for k,v in acc.items()}, 'rows', n)
PY
done
