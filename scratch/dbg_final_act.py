import sys, os; sys.path.insert(0, os.getcwd())
import numpy as np, torch
from iterseg_amd import weights
from iterseg_amd.train import Trainer
from oracle import unet_torch as ou
d = np.load("tests/golden/final_activations_4x32x32.npz")
for acts, loss_name in [(["sigmoid"], "MSELoss"), (["tanh"], "MSELoss"), (["relu"], "MSELoss"), (["softmax"], "BCELoss"), (["linear"], "MSELoss")]:
    sd = weights.generate_state_dict(4, 1, 5)
    t = Trainer(sd, out_channels=5, patch=(4, 32, 32), chan_final_activations=acts)
    loss, _ = t._call(d["x"], d["y"], loss_name, None, 0.01, 0, True)
    p = ou._as_params(sd, requires_grad=True, dtype=torch.float64)
    yh = ou.unet_forward(p, torch.from_numpy(d["x"])[None, None].double(), final=acts)
    yy = torch.from_numpy(d["y"])[None].double()
    l = torch.nn.functional.mse_loss(yh, yy) if loss_name == "MSELoss" else torch.nn.functional.binary_cross_entropy(yh, yy)
    l.backward()
    print(acts, loss_name, "loss", loss, float(l.detach()))
    worst = []
    for k, shape, off in t.layout:
        ref = p[k].grad.numpy()
        got = t.grad(k).detach().cpu().numpy().astype(np.float64)
        worst.append((np.abs(got - ref).max() / (np.abs(ref).max() + 1e-30), k, np.abs(ref).max()))
    worst = [w for w in worst if w[2] > 1e-9]; worst.sort(reverse=True)
    for w in worst[:6]: print("   ", w)
