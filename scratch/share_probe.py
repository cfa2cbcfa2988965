"""GPU-sharing probe of the giant-component flood and of the training step: run two copies at once; every repetition must
reproduce the first result of its own process bit for bit."""
import os, sys; sys.path.insert(0, os.getcwd())
import numpy as np, torch
from iterseg_amd import watershed as ws, weights
from iterseg_amd.unet import UNet
from iterseg_amd.train import Trainer
from tests.synth import positive_volume
bad = 0
# --- config 2's giant through the work-group flood
shape = (64, 512, 512)
u = UNet(1, 5, device=0, batch=32); u.load_state_dict(weights.generate_state_dict(0))
fm = u.predict_volume(torch.from_numpy(positive_volume(shape, 0)).cuda(), (10, 256, 256), (1, 64, 64)).clone()
def labels():
    lab = torch.zeros(tuple(s + 2 for s in shape), dtype=torch.int32, device='cuda')
    ws.segment_output_image(fm, (0, 1, 2), 4, 3, out=lab); return lab
ref = labels().clone()
for it in range(3):
    if not torch.equal(labels(), ref): bad += 1; print("giant flood: labels differ at repetition", it, flush=True)
print("giant flood done", flush=True)
del u, fm, ref; torch.cuda.empty_cache()
# --- training step, fp32 and bf16
rng = np.random.default_rng(3)
x = positive_volume((10, 128, 128), 2); y = (rng.random((5, 10, 128, 128)) < 0.4).astype(np.float32)
for prec in ("fp32", "bf16"):
    outs = []
    for rep in range(4):
        t = Trainer(weights.generate_state_dict(1), out_channels=5, patch=(10, 128, 128), precision=prec)
        for s in range(3): t.train_step(x, y, lr=0.01, loss="BCELoss")
        outs.append(t.params.clone())
        del t
    for rep in range(1, 4):
        if not torch.equal(outs[rep], outs[0]): bad += 1; print(f"training {prec}: parameters differ at repetition {rep}: max {float((outs[rep]-outs[0]).abs().max()):.3e}", flush=True)
    print("training", prec, "done", flush=True)
print(f"pid {os.getpid()} done, {bad} mismatches", flush=True)
