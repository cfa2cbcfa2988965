# time of one 64-chunk launch set of the U-Net (ablation builds: the results are wrong on purpose)
import sys, os, time; sys.path.insert(0, os.getcwd())
import numpy as np, torch
from iterseg_amd import weights
from iterseg_amd.unet import UNet
from tests.synth import positive_volume
B = 64
u = UNet(1, 5, batch=B); u.load_state_dict(weights.generate_state_dict(0))
vol = torch.from_numpy(positive_volume((64, 512, 512), 0)).cuda()
fm = torch.zeros((5, 64, 512, 512), device='cuda')
for _ in range(3): u.predict_volume(vol, (10, 256, 256), (1, 64, 64), feature_map=fm, first=0, count=B)
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 8
for _ in range(n): u.predict_volume(vol, (10, 256, 256), (1, 64, 64), feature_map=fm, first=0, count=B)
torch.cuda.synchronize(); print(f"{sys.argv[1] if len(sys.argv) > 1 else ''} launch set {(time.perf_counter() - t0) / n * 1e3:.2f} ms")
