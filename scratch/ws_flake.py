"""Determinism probe of the watershed alone: one feature map, repeated segment_output_image; mask, seeds and labels compared."""
import os, sys; sys.path.insert(0, os.getcwd())
import numpy as np, torch
from iterseg_amd import watershed as ws, weights
from iterseg_amd.unet import UNet
from tests.synth import positive_volume
chunk, margin = (4, 32, 32), (1, 8, 8)
u = UNet(1, 5, device=0, batch=4); u.load_state_dict(weights.generate_state_dict(0))
fms = []
for t in range(5):
    v = positive_volume((8, 40, 48), 30 + t); v = (v / v.max()).astype(np.float32)
    fms.append(u.predict_volume(torch.from_numpy(v).cuda(), chunk, margin).clone())
torch.cuda.synchronize()
def run(fm):
    shape = tuple(fm.shape[1:]); pshape = tuple(s + 2 for s in shape)
    lab = torch.zeros(pshape, dtype=torch.int32, device='cuda')
    seg, seeds, mask = ws.segment_output_image(fm, (0, 1, 2), 4, 3, out=lab)
    torch.cuda.synchronize()
    h = lambda a: a.cpu().numpy().copy() if isinstance(a, torch.Tensor) else np.asarray(a).copy()
    return h(lab), h(seeds), h(mask)
ref = [run(f) for f in fms]
bad = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 300):
    for t, f in enumerate(fms):
        lab, seeds, mask = run(f)
        el, es, em = np.array_equal(lab, ref[t][0]), np.array_equal(seeds, ref[t][1]), np.array_equal(mask, ref[t][2])
        if not (el and es and em):
            bad += 1
            d = np.argwhere(lab != ref[t][0])
            print(f"pid {os.getpid()} iter {it} frame {t}: labels {el} seeds {es} mask {em}; diff voxels {len(d)}: " +
                  ", ".join(f"{tuple(p)}: {ref[t][0][tuple(p)]}->{lab[tuple(p)]}" for p in d[:6]), flush=True)
print(f"pid {os.getpid()} done, {bad} mismatches", flush=True)
