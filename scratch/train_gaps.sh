# kernel trace of the training step: busy time against the span (gaps between kernels = launch-bound share)
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/tg; mkdir -p $O
cd $R
for prec in fp32 bf16; do
  ISX_TRAIN_PRECISION=$prec rocprofv3 --kernel-trace --output-format csv -d $O/p_$prec -o t -- python3 scratch/bench_train.py > $O/run_$prec.log 2>&1
  tail -2 $O/run_$prec.log | head -1
  python3 - $O/p_$prec <<'PY'
import sys, glob, csv
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# the last training step: from the last 'loss' kernel backwards ... simpler: take the last 2000 kernels, find periodicity by the adam kernel
names = [r["Kernel_Name"] for r in rows]
ad = [i for i, n in enumerate(names) if "adam" in n.lower()]
# steps end with adam kernels; evaluate() calls follow the 7 train steps -> use adam occurrences
ends = [ad[i] for i in range(len(ad)) if i + 1 == len(ad) or ad[i + 1] - ad[i] > 5]
if len(ends) >= 2:
    a, b = ends[-2] + 1, ends[-1] + 1
    seg = rows[a:b]
    span = int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
    gaps = sorted((int(seg[i + 1]["Start_Timestamp"]) - int(seg[i]["End_Timestamp"]) for i in range(len(seg) - 1)), reverse=True)
    print(f"  step: {len(seg)} kernels, span {span/1e6:.3f} ms, busy {busy/1e6:.3f} ms, gaps {(span-busy)/1e6:.3f} ms; largest gaps (us): {[g/1e3 for g in gaps[:8]]}; median gap {gaps[len(gaps)//2]/1e3:.1f} us")
PY
done
