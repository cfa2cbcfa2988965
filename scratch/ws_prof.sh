# kernel stats of the watershed stage of the headline frame (bench warm-up + 1 step): bash scratch/ws_prof.sh TAG [ENV=VAL ...]
T=$1; shift
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/wsp_$T -o w -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-extra-legs --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/wsp_$T.json 2>/dev/null
cd $GRAFT_REPO_ROOT; python3 - $T <<PY
import csv,glob,sys,json
T=sys.argv[1]
f=glob.glob(f"gpurun_out/wsp_{T}/**/w_kernel_stats.csv",recursive=True)[0]
tot=0
for r in csv.DictReader(open(f)):
    n=r["Name"]
    if any(k in n for k in ["ccl","gauss","minmax","aff_pad","mask_kernel","peak_kernel","hist","chanmax","flood_state","otsu"]):
        t=float(r["TotalDurationNs"])/1e6/2; tot+=t; print(f"  {n[:58]:58s} {int(r['Calls'])//2:2d} {t:7.3f}")
d=json.loads(open(f"gpurun_out/wsp_{T}.json").read().strip().splitlines()[-1])
print(T, "sum", round(tot,3), "t_ws", d["t_ws_ms"], d["stages_ms_per_step"])
PY
