set -e
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ts; mkdir -p $O
cd $R
rocprofv3 --kernel-trace --output-format csv -d $O/p -o t -- python3 scratch/time_structured.py "" 3 > $O/run.log 2>&1
python3 - <<'PY' > $O/sum.txt
import glob, csv, collections
f = glob.glob("gpurun_out/ts/p/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last segment_output_image call: take the last occurrence of each flood kernel
last = {}
for r in rows:
    n = r["Kernel_Name"].split("(")[0][-60:]
    last[n] = (int(r["Start_Timestamp"]), int(r["End_Timestamp"]))
t0 = min(v[0] for k, v in last.items() if "flood" in k)
for k, v in sorted(last.items(), key=lambda kv: kv[1][0]):
    if v[0] >= t0 - 30e6: print(f"{(v[0]-t0)/1e6:9.3f} -> {(v[1]-t0)/1e6:9.3f} ms  ({(v[1]-v[0])/1e6:8.3f})  {k}")
PY
