# Are two launch sets in flight on two streams faster than one after the other?  (tails of small kernels under the other's convolutions)
import sys, os, time; sys.path.insert(0, os.getcwd())
import numpy as np, torch
from iterseg_amd import weights
from iterseg_amd.unet import UNet
from tests.synth import positive_volume
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
sd = weights.generate_state_dict(0)
shape = (512, 512, 512)
vol = torch.from_numpy(positive_volume(shape, 0)).cuda()
chunk, margin = (10, 256, 256), (1, 64, 64)
us = [UNet(1, 5, batch=B), UNet(1, 5, batch=B)]
for u in us: u.load_state_dict(sd)
fm = torch.zeros((5,) + shape, device='cuda')
ss = [torch.cuda.Stream(), torch.cuda.Stream()]
nsets = 576 // B
def one():
    for k in range(nsets): us[0].predict_volume(vol, chunk, margin, feature_map=fm, first=k * B, count=B)
def two():
    for k in range(nsets):
        with torch.cuda.stream(ss[k & 1]):
            us[k & 1].predict_volume(vol, chunk, margin, feature_map=fm, first=k * B, count=B)
for f in (one, two): f(); torch.cuda.synchronize()
ref = None
for name, f in (("one stream", one), ("two streams", two), ("one stream", one), ("two streams", two)):
    torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    s = float(fm.double().sum())
    print(f"{name}: U-Net of the 512^3 frame {dt*1e3:.1f} ms (checksum {s:.6f})", flush=True)
