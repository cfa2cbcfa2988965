# headline bench for two development builds (scratch/libs/lib_late.so / lib_early.so), alternating, selected with ISX_LIB
set -e
cd $GRAFT_REPO_ROOT
for v in late early late early; do
  ISX_LIB=$GRAFT_REPO_ROOT/scratch/libs/lib_${v}.so timeout -k 10 200 python bench.py --steps 6 --warmup 2 --no-extra-legs --no-cpu-baseline > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err
  python -c "
import json; d=json.loads(open('gpurun_out/ab_$v.json').read().strip().splitlines()[-1]); print('$v', d['ms_per_step'], d['roofline']['avg_launch_ms'])" >> gpurun_out/ab_summary.txt
done
