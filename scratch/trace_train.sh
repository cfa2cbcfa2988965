set -e
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/tt; mkdir -p $O
cd $R
for prec in fp32 bf16; do
  ISX_TRAIN_PRECISION=$prec rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_$prec -o t -- python3 scratch/bench_train.py > $O/run_$prec.log 2>&1
  tail -2 $O/run_$prec.log
  f=$(find $O/p_$prec -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY' > $O/stats_$prec.txt
import sys, csv
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:22]:
    print(f'{float(r["TotalDurationNs"])/1e6:9.2f} ms {float(r["TotalDurationNs"])/tot*100:5.1f}% n={r["Calls"]:>5} avg {float(r["AverageNs"])/1e3:9.1f} us  {r["Name"][:90]}')
PY
done
