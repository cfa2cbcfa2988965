# A/B of an isx_tune knob inside one process: two contexts (KNOB = v0 / v1), one 64-chunk launch set each, feature maps
# compared, launch-set time per variant (interleaved).  usage: knob_ab.py KNOB v0 v1 [precision]
import sys, os, time; sys.path.insert(0, os.getcwd())
import numpy as np, torch
from iterseg_amd import weights
from iterseg_amd.unet import UNet
from tests.synth import positive_volume
B = 64
knob, v0, v1 = sys.argv[1], sys.argv[2], sys.argv[3]
prec = sys.argv[4] if len(sys.argv) > 4 else "split-f16"
sd = weights.generate_state_dict(0)
us = {}
for k in (v0, v1):
    os.environ[knob] = k
    us[k] = UNet(1, 5, batch=B, precision=prec); us[k].load_state_dict(sd)
vol = torch.from_numpy(positive_volume((64, 512, 512), 0)).cuda()
fms = {k: torch.zeros((5, 64, 512, 512), device='cuda') for k in us}
def run(k, n):
    for _ in range(n): us[k].predict_volume(vol, (10, 256, 256), (1, 64, 64), feature_map=fms[k], first=0, count=B)
for k in us: run(k, 2)
torch.cuda.synchronize()
d = (fms[v0] - fms[v1]).abs().max().item()
print(f"{prec}: max |{knob}={v0} - {knob}={v1}| on the feature maps = {d:.3e}", flush=True)
for rep in range(4):
    for k in us:
        torch.cuda.synchronize(); t0 = time.perf_counter(); run(k, 4); torch.cuda.synchronize()
        print(f"  {knob}={k} launch set {(time.perf_counter() - t0) / 4 * 1e3:.2f} ms", flush=True)
