"""Determinism probe at the timed geometry: (32,512,512) volume, chunk (10,256,256), margin (1,64,64); two copies at once."""
import os, sys; sys.path.insert(0, os.getcwd())
import numpy as np, torch
from iterseg_amd import weights, watershed as ws
from iterseg_amd.unet import UNet
from tests.synth import positive_volume
prec = os.environ.get("PREC", "split-f16")
u = UNet(1, 5, device=0, batch=32, precision=prec); u.load_state_dict(weights.generate_state_dict(0))
vols = [torch.from_numpy(positive_volume((32, 512, 512), s)).cuda() for s in (1, 2)]
ref = [u.predict_volume(v, (10, 256, 256), (1, 64, 64)).clone() for v in vols]
def labels(fm):
    lab = torch.zeros(tuple(s + 2 for s in fm.shape[1:]), dtype=torch.int32, device='cuda')
    ws.segment_output_image(fm, (0, 1, 2), 4, 3, out=lab); return lab
reflab = [labels(f).clone() for f in ref]
bad = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
    for t, v in enumerate(vols):
        fm = u.predict_volume(v, (10, 256, 256), (1, 64, 64))
        if not torch.equal(fm, ref[t]):
            bad += 1; d = (fm - ref[t]).abs(); print(f"{prec} iter {it} vol {t}: fm max diff {d.max().item():.3e}, {(d > 0).sum().item()} values", flush=True)
        elif not torch.equal(labels(fm), reflab[t]):
            bad += 1; print(f"{prec} iter {it} vol {t}: labels differ", flush=True)
print(f"{prec} pid {os.getpid()} done, {bad} mismatches", flush=True)
