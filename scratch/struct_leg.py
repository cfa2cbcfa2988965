import sys, os, time; sys.path.insert(0, os.getcwd())
import torch, bench
from iterseg_amd import weights
from iterseg_amd.unet import UNet
u = UNet(1, 5, device=0, batch=64); u.load_state_dict(weights.generate_state_dict(0))
for i in range(2):
    r = bench.structured_watershed_leg(u)
    print("leg alone:", round(r["ms"], 1), {k: round(v, 1) for k, v in r["stages_ms"].items()}, flush=True)
r = bench.config2_leg(u, 0, (10, 256, 256), (1, 64, 64), 64)
print("config2 flood", r["stages_ms"]["ws_flood"], flush=True)
for i in range(2):
    r = bench.structured_watershed_leg(u)
    print("leg after config2:", round(r["ms"], 1), {k: round(v, 1) for k, v in r["stages_ms"].items()}, flush=True)
