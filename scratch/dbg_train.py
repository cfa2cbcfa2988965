import sys, os; sys.path.insert(0, os.getcwd())
import numpy as np, torch
from iterseg_amd import weights
from iterseg_amd.train import Trainer
from oracle import unet_torch as ou
from tests.synth import positive_volume
torch.set_num_threads(16)
patch = (10, 256, 256) if len(sys.argv) < 2 else tuple(int(v) for v in sys.argv[1].split(","))
sd = weights.generate_state_dict(0)
x = positive_volume(patch, 9)
y = (np.random.default_rng(9).random((5,) + patch) < 0.5).astype(np.float32)
l32, _, g32, _ = ou.train_step(sd, x, y, lr=0.01, double_step=False)
l64, _, g64, _ = ou.train_step(sd, x, y, lr=0.01, double_step=False, dtype=torch.float64)
t = Trainer(sd, 5, patch)
loss, _ = t._call(x, y, "BCELoss", None, 0.01, 0, True)
print("loss", loss, l32, l64)
for k, shape, off in t.layout:
    ref = g64[k]; got = t.grad(k).cpu().numpy().astype(np.float64)
    mine = np.abs(got - ref).max(); theirs = np.abs(g32[k].astype(np.float64) - ref).max(); sc = np.abs(ref).max()
    flag = "  <<<" if mine > max(4 * theirs, 2e-3 * sc) + 1e-9 else ""
    print(f"{k:22s} max|g| {sc:.3e}  mine {mine:.2e} ({mine/sc*100:6.3f}%)  torch32 {theirs:.2e} ({theirs/sc*100:6.3f}%){flag}")
