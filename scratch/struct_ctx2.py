import sys, os, time; sys.path.insert(0, os.getcwd())
import numpy as np, torch
from iterseg_amd import watershed as ws, weights
from iterseg_amd.unet import UNet
from tests.synth import blob_feature_map_tiled
mode = sys.argv[1]
n = 512
if mode == "unet_first":
    u = UNet(1, 5, device=0, batch=64); u.load_state_dict(weights.generate_state_dict(0))
dfm = torch.from_numpy(blob_feature_map_tiled(n, 0, 0.12, 3.0)).cuda()
lab = torch.zeros((n + 2,) * 3, dtype=torch.int32, device="cuda")
ts = []
for _ in range(3):
    lab.zero_()
    if mode == "prof_each":
        c = ws.get_context(0); c.prof_enable(False); c.prof_enable(True)
    torch.cuda.synchronize(); t = time.time()
    r = ws.segment_output_image(dfm, (0, 1, 2), 4, 3, out=lab); torch.cuda.synchronize(); ts.append(time.time() - t)
print(mode, [round(x * 1e3, 1) for x in ts], flush=True)
if mode == "prof_each":
    c = ws.get_context(0); print({k: round(c.prof_get(k)[0], 1) for k in ("ws_filters", "ws_ccl", "ws_flood")})
