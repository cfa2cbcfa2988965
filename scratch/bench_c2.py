"""Flood of a random-weight U-Net feature map (one giant chain-like component): stage times + ISX_FLOOD_STATS."""
import sys, os, time; sys.path.insert(0, os.getcwd())
import numpy as np, torch
from iterseg_amd import watershed as ws, weights
from iterseg_amd.unet import UNet
from tests.synth import positive_volume
shape = tuple(int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (64, 512, 512)
sd = weights.generate_state_dict(0)
u = UNet(1, 5, device=0, batch=32); u.load_state_dict(sd)
vol = torch.from_numpy(positive_volume(shape, 0)).cuda()
fm = u.predict_volume(vol, (10, 256, 256), (1, 64, 64)).clone()
torch.cuda.synchronize()
ctx = ws.get_context()
pshape = tuple(s + 2 for s in shape)
for rep in range(2):
    lab = torch.zeros(pshape, dtype=torch.int32, device='cuda')
    ctx.prof_enable(False); ctx.prof_enable(True)
    torch.cuda.synchronize(); t = time.perf_counter()
    seg, seeds, mask = ws.segment_output_image(fm, (0, 1, 2), 4, 3, out=lab)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    p = {k: round(ctx.prof_get(k)[0], 2) for k in ("ws_filters", "ws_ccl", "ws_flood")}
    nl = int((lab > 0).sum())
    print(f"c2 {shape} rep {rep}: total {dt*1e3:.1f} ms; {p}; seeds {len(seeds)} labelled {nl} -> flood {nl/max(p['ws_flood'],1e-9)/1e3:.2f} Mpops/s", flush=True)
if os.environ.get("CHECK"):
    from oracle import watershed as ows
    ref = np.zeros(pshape, np.uint32)
    t = time.time(); ows.segment_output_image(fm.cpu().numpy(), out=ref.ravel()); print("oracle", time.time() - t, "s")
    print("labels bit-exact vs oracle:", bool(np.array_equal(ref, lab.cpu().numpy().astype(np.uint32))))
