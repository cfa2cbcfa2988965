# package power and shader clock while one library build runs launch sets back to back: bash scratch/power_run.sh TAG prod abl8 ...
cd $GRAFT_REPO_ROOT; export PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/power_$1; mkdir -p $O; shift
cat > /tmp/loop.py <<'PY'
import sys, os, time; sys.path.insert(0, os.getcwd())
import torch
from iterseg_amd import weights
from iterseg_amd.unet import UNet
from tests.synth import positive_volume
B = 64
u = UNet(1, 5, batch=B); u.load_state_dict(weights.generate_state_dict(0))
vol = torch.from_numpy(positive_volume((64, 512, 512), 0)).cuda()
fm = torch.zeros((5, 64, 512, 512), device='cuda')
for _ in range(2): u.predict_volume(vol, (10, 256, 256), (1, 64, 64), feature_map=fm, first=0, count=B)
torch.cuda.synchronize(); print("START", flush=True); t0 = time.perf_counter(); n = 40
for _ in range(n): u.predict_volume(vol, (10, 256, 256), (1, 64, 64), feature_map=fm, first=0, count=B)
torch.cuda.synchronize(); print(f"launch set {(time.perf_counter() - t0) / n * 1e3:.2f} ms", flush=True)
PY
for v in "$@"; do
  L=$GRAFT_REPO_ROOT/iterseg_amd/libitersegx.so; [ $v = prod ] || L=$GRAFT_REPO_ROOT/scratch/libs/lib_$v.so
  ( for i in $(seq 1 60); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo; sleep 0.2; done ) > $O/watch_$v.log 2>&1 &
  W=$!
  ISX_LIB=$L python3 /tmp/loop.py > $O/loop_$v.log 2>&1
  kill $W 2>/dev/null; wait $W 2>/dev/null
  echo "== $v: $(tail -1 $O/loop_$v.log)"; grep -E "package power W: (9|1[0-9])[0-9][0-9]" $O/watch_$v.log | tail -8 | sed 's/sclk clock level: 1: //'
done
