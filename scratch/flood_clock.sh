# shader clock / power while the config-2 flood (one work-group on one CU) runs
cd $GRAFT_REPO_ROOT
( for i in $(seq 1 80); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo; sleep 0.1; done ) > gpurun_out/flood_clock.log 2>&1 &
W=$!
python scratch/bench_c2.py 2>&1 | grep "c2 " 
kill $W 2>/dev/null; wait $W 2>/dev/null
python3 - <<'PY'
import re
rows=[]
for l in open('gpurun_out/flood_clock.log'):
    m=re.search(r"\((\d+)Mhz\).*\(W\): ([\d.]+)", l)
    if m: rows.append((int(m.group(1)), float(m.group(2))))
print("samples", len(rows)); print(rows[::3])
PY
