set -e
cd $GRAFT_REPO_ROOT
for v in 0 2 0 2; do
  ISX_H2_NW8=$v timeout -k 10 200 python bench.py --steps 6 --warmup 2 --no-extra-legs --no-cpu-baseline > gpurun_out/ab_n$v.json 2> gpurun_out/ab_n$v.err
  python -c "
import json; d=json.loads(open('gpurun_out/ab_n$v.json').read().strip().splitlines()[-1]); print('nw8=$v', d['ms_per_step'], d['roofline']['avg_launch_ms'])" >> gpurun_out/ab_summary.txt
done
timeout -k 10 400 python -m pytest tests/test_gpu_unet.py -x -q -k "512_thread or core" > gpurun_out/ab_test.log 2>&1
