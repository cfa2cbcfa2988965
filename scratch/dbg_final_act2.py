import sys, os; sys.path.insert(0, os.getcwd())
import numpy as np, torch
from iterseg_amd import weights
from iterseg_amd.train import Trainer
from oracle import unet_torch as ou
d = np.load("tests/golden/final_activations_4x32x32.npz")
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 4
loss_name = sys.argv[2] if len(sys.argv) > 2 else "MSELoss"
sd = weights.generate_state_dict(seed, 1, 5)
t = Trainer(sd, out_channels=5, patch=(4, 32, 32))
loss, _ = t._call(d["x"], d["y"], loss_name, None, 0.01, 0, True)
p = ou._as_params(sd, requires_grad=True, dtype=torch.float64)
taps = {}
yh = ou.unet_forward(p, torch.from_numpy(d["x"])[None, None].double(), taps=taps)
for v in taps.values():
    if v.requires_grad: v.retain_grad()
yy = torch.from_numpy(d["y"])[None].double()
l = torch.nn.functional.mse_loss(yh, yy) if loss_name == "MSELoss" else torch.nn.functional.binary_cross_entropy(yh, yy)
l.backward()
names = ["c0", "c1", "c2", "c3", "c4", "c5_0", "c6_0", "c7_0", "c8_0"]
for li in range(17, 1, -1):
    nm = f"{names[li // 2]}.conv{li % 2}"
    ref = taps[nm].grad[0].numpy()                      # (C, D, H, W)
    C = ref.shape[0]
    got = t.grad_tap(f"draw:{li}")[:, :C].T.reshape(ref.shape)
    e = np.abs(got - ref)
    z = taps[f"{names[li // 2]}.batch{li % 2}"][0].detach().numpy()
    i = np.unravel_index(np.argmax(e), e.shape)
    print(f"draw:{li} {nm}: max err {e.max():.3e} / max {np.abs(ref).max():.3e} = {e.max()/np.abs(ref).max():.2e}; n(err > 1e-3 max) {(e > 1e-3*np.abs(ref).max()).sum()} at {i} bn-out there {z[i]:.3e}; min|z| {np.abs(z).min():.2e}")
