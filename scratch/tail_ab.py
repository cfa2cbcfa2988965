# A/B of the tail layer inside one process: two contexts (ISX_TAIL_ONEPASS = 0 / 1), one 64-chunk launch set each,
# feature maps compared, launch-set time per variant (interleaved).
import sys, os, time; sys.path.insert(0, os.getcwd())
import numpy as np, torch
from iterseg_amd import weights
from iterseg_amd.unet import UNet
from tests.synth import positive_volume
B = 64
prec = sys.argv[1] if len(sys.argv) > 1 else "split-f16"
sd = weights.generate_state_dict(0)
us = {}
for k in ("0", "1"):
    os.environ["ISX_TAIL_ONEPASS"] = k
    us[k] = UNet(1, 5, batch=B, precision=prec); us[k].load_state_dict(sd)
vol = torch.from_numpy(positive_volume((64, 512, 512), 0)).cuda()
fms = {k: torch.zeros((5, 64, 512, 512), device='cuda') for k in us}
def run(k, n):
    for _ in range(n): us[k].predict_volume(vol, (10, 256, 256), (1, 64, 64), feature_map=fms[k], first=0, count=B)
for k in us: run(k, 2)
torch.cuda.synchronize()
d = (fms["0"] - fms["1"]).abs().max().item()
print(f"{prec}: max |two-pass - one-pass| on the feature maps = {d:.3e}", flush=True)
for rep in range(3):
    for k in us:
        torch.cuda.synchronize(); t0 = time.perf_counter(); run(k, 4); torch.cuda.synchronize()
        print(f"  onepass={k} launch set {(time.perf_counter() - t0) / 4 * 1e3:.2f} ms", flush=True)
