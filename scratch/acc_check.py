# split-f16 against the exact-fp32 mode on one 64-chunk launch set (max |difference| of the feature maps) and the launch-set time
import sys, os, time; sys.path.insert(0, os.getcwd())
import numpy as np, torch
from iterseg_amd import weights
from iterseg_amd.unet import UNet
from tests.synth import positive_volume
B = 64
sd = weights.generate_state_dict(0)
vol = torch.from_numpy(positive_volume((64, 512, 512), 0)).cuda()
fms = {}
for prec in ("split-f16", "fp32"):
    u = UNet(1, 5, batch=B, precision=prec); u.load_state_dict(sd)
    fm = torch.zeros((5, 64, 512, 512), device='cuda')
    for _ in range(2): u.predict_volume(vol, (10, 256, 256), (1, 64, 64), feature_map=fm, first=0, count=B)
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 6 if prec != "fp32" else 1
    for _ in range(n): u.predict_volume(vol, (10, 256, 256), (1, 64, 64), feature_map=fm, first=0, count=B)
    torch.cuda.synchronize(); print(f"{prec}: launch set {(time.perf_counter() - t0) / n * 1e3:.2f} ms", flush=True)
    fms[prec] = fm; del u
d = (fms["split-f16"] - fms["fp32"]).abs()
print(f"max |split-f16 - fp32| = {d.max().item():.3e}, mean {d.mean().item():.3e}")
