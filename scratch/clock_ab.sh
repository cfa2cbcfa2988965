set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -m pytest tests/test_gpu_unet.py -x -q -k "planar" > gpurun_out/ab_test.log 2>&1
for v in 0 1 0 1; do
  ISX_PLANAR=$v timeout -k 10 200 python bench.py --steps 6 --warmup 2 --no-extra-legs --no-cpu-baseline > gpurun_out/ab_t.json 2> gpurun_out/ab_t.err
  python -c "
import json; d=json.loads(open('gpurun_out/ab_t.json').read().strip().splitlines()[-1]); print('planar=$v', d['ms_per_step'], d['roofline']['avg_launch_ms'])" >> gpurun_out/ab_summary.txt
done
