"""Config 2 flood under knob settings, one process (feature map computed once): python scratch/c2_sweep.py "K=V,K2=V2" "K=V" ..."""
import sys, os, time; sys.path.insert(0, os.getcwd())
import numpy as np, torch
from iterseg_amd import watershed as ws, weights
from iterseg_amd.unet import UNet
from tests.synth import positive_volume
shape = (64, 512, 512)
u = UNet(1, 5, device=0, batch=32); u.load_state_dict(weights.generate_state_dict(0))
vol = torch.from_numpy(positive_volume(shape, 0)).cuda()
fm = u.predict_volume(vol, (10, 256, 256), (1, 64, 64)).clone()
torch.cuda.synchronize()
ctx = ws.get_context()
pshape = tuple(s + 2 for s in shape)
ref = None
for setting in [""] + sys.argv[1:] + [""]:
    kv = dict(p.split("=") for p in setting.split(",") if p)
    for k, v in kv.items(): os.environ[k] = v
    ctx.reload_knobs()
    ts = []
    for rep in range(3):
        lab = torch.zeros(pshape, dtype=torch.int32, device='cuda')
        ctx.prof_enable(False); ctx.prof_enable(True)
        seg, seeds, mask = ws.segment_output_image(fm, (0, 1, 2), 4, 3, out=lab)
        torch.cuda.synchronize()
        ts.append(ctx.prof_get("ws_flood")[0])
    if ref is None: ref = lab.clone()
    print(f"{setting or 'default':40s} flood ms {[round(t, 1) for t in ts]}  same labels {bool(torch.equal(ref, lab))}", flush=True)
    for k in kv: del os.environ[k]
