"""Determinism probe of the chunked U-Net predict on a toy frame: feature maps must be bit-identical run to run."""
import os, sys; sys.path.insert(0, os.getcwd())
import numpy as np, torch
from iterseg_amd import weights
from iterseg_amd.unet import UNet
from tests.synth import positive_volume
chunk, margin = (4, 32, 32), (1, 8, 8)
u = UNet(1, 5, device=0, batch=4, precision=os.environ.get("PREC", "split-f16")); u.load_state_dict(weights.generate_state_dict(0))
vols = []
for t in range(5):
    v = positive_volume((8, 40, 48), 30 + t); vols.append(torch.from_numpy((v / v.max()).astype(np.float32)).cuda())
ref, reftaps = [], []
taps = ["c0.conv0", "c0.conv1", "c1.conv0", "c1.conv1", "c2.conv1", "c3.conv1", "c4.conv0", "c4.conv1", "c5_0.conv1", "c6_0.conv1", "c7_0.conv1", "c8_0.conv0", "c8_0.conv1", "c8_0"]
for v in vols:
    ref.append(u.predict_volume(v, chunk, margin).clone()); torch.cuda.synchronize()
    reftaps.append({k: u.tap(k) for k in taps})
taps = ["c0.conv0", "c0.conv1", "c1.conv0", "c1.conv1", "c2.conv1", "c3.conv1", "c4.conv0", "c4.conv1", "c5_0.conv1", "c6_0.conv1", "c7_0.conv1", "c8_0.conv0", "c8_0.conv1"]
reftap = None
bad = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 300):
    for t, v in enumerate(vols):
        fm = u.predict_volume(v, chunk, margin)
        if not torch.equal(fm, ref[t]):
            bad += 1
            d = (fm - ref[t]).abs()
            torch.cuda.synchronize()
            first = [k for k in taps if not np.array_equal(u.tap(k), reftaps[t][k])]
            if "c8_0.conv1" in first and bad <= 40:
                a, b = u.tap("c8_0.conv1"), reftaps[t]["c8_0.conv1"]
                w = np.argwhere(np.abs(a - b) > 1e-4 * np.abs(b).max())
                print("   c8_0.conv1 differs at", len(w), "positions (c,z,y,x):", [tuple(int(q) for q in p) for p in w[:10]], "values", [(float(a[tuple(p)]), float(b[tuple(p)])) for p in w[:4]], flush=True)
            if bad <= 12: print(f"pid {os.getpid()} iter {it} frame {t}: max diff {d.max().item():.3e}, differing values {(d > 0).sum().item()} of {d.numel()}; taps of the last set that differ: {first}", flush=True)
print(f"pid {os.getpid()} done, {bad} mismatches", flush=True)
