import sys, os; sys.path.insert(0, os.getcwd())
import numpy as np, torch
from iterseg_amd import weights
from iterseg_amd.train import Trainer
from iterseg_amd.unet import UNet
from oracle import unet_torch as ou
from tests.synth import positive_volume
rng = np.random.default_rng(3)
x = positive_volume((4, 32, 32), 1); y = (rng.random((5, 4, 32, 32)) < 0.4).astype(np.float32)
sd = weights.generate_state_dict(0)
def T(): return Trainer(sd, out_channels=5, patch=(4, 32, 32))
a = T(); a.train_step(x, y); la = a.train_step(x, y)[0]
for geom in [(6, 48, 64), (4, 32, 48), (4, 48, 32), (6, 32, 32), (8, 64, 64), (4, 32, 32)]:
    b = T(); b.train_step(x, y)
    vol = positive_volume(geom, 2)
    fm = b.unet.predict_volume(torch.from_numpy(vol).cuda(), geom, (1, 8, 8)).cpu().numpy()
    u = UNet(1, 5, batch=1, precision="fp32"); u.load_state_dict(b.state_dict())
    fm2 = u.predict_volume(torch.from_numpy(vol).cuda(), geom, (1, 8, 8)).cpu().numpy()
    lb = b.train_step(x, y)[0]
    print(geom, "loss after", lb, "(undisturbed", la, ") params equal:", torch.equal(a.params, b.params),
          "| predict on trainer ctx vs fresh ctx max diff", float(np.abs(fm - fm2).max()), flush=True)
