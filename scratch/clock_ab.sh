set -e
cd $GRAFT_REPO_ROOT
for v in old_plain pipe_plain pipe_clock old_plain pipe_plain; do
  cp scratch/libs/lib_${v}.so iterseg_amd/libitersegx.so
  ISX_H2_STAMP_PRINT=1 timeout -k 10 200 python bench.py --steps 6 --warmup 2 --no-extra-legs --no-cpu-baseline > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err
  python -c "
import json; d=json.loads(open('gpurun_out/ab_$v.json').read().strip().splitlines()[-1]); print('$v', d['ms_per_step'], d['roofline']['avg_launch_ms'])" >> gpurun_out/ab_summary.txt
done
cp scratch/libs/lib_pipe_plain.so iterseg_amd/libitersegx.so
