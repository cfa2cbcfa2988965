import sys, os, time; sys.path.insert(0, os.getcwd())
import numpy as np, torch
from iterseg_amd import weights
from iterseg_amd.train import Trainer
from tests.synth import positive_volume
patch = (10, 256, 256)
t = Trainer(weights.generate_state_dict(0), 5, patch, precision=os.environ.get("ISX_TRAIN_PRECISION", "fp32"))
x = torch.from_numpy(positive_volume(patch, 1)).cuda()
y = torch.from_numpy((np.random.default_rng(1).random((5,) + patch) < 0.5).astype(np.float32)).cuda()
for _ in range(2): t.train_step(x, y)
torch.cuda.synchronize(); t0 = time.perf_counter(); n = 5
for _ in range(n): l, _ = t.train_step(x, y)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print(f"train step (fwd+loss+bwd+2xAdam) {dt*1e3:.1f} ms/step  loss {l:.4f}  -> {3*369.9e9/dt/1e12:.1f} TFLOP/s of 3x fwd FLOPs")
t0 = time.perf_counter()
for _ in range(n): t.evaluate(x, y)
torch.cuda.synchronize(); print(f"forward+loss only {(time.perf_counter()-t0)/n*1e3:.1f} ms")
