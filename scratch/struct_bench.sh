cd $GRAFT_REPO_ROOT
for g in 262144 65536; do
  ISX_FLOOD_GROUP_MIN=$g python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/sb_$g.json 2>/dev/null
  python - <<PY
import json; d=json.loads(open("gpurun_out/sb_$g.json").read().strip().splitlines()[-1]); w=d["watershed_structured"]; print($g, w["ms"], w["stages_ms"], "config2", d["config2"]["stages_ms"]["ws_flood"])
PY
done
