"""Determinism probe: the one-frame host path on a toy frame, repeated; feature map and labels must be bit-identical run to run.
Run two copies at once to load the GPU: python scratch/stack_flake.py & python scratch/stack_flake.py"""
import os, sys; sys.path.insert(0, os.getcwd())
import numpy as np, torch
from iterseg_amd import segmentation as seg, weights
from iterseg_amd.unet import UNet
from tests.synth import positive_volume
chunk, margin = (4, 32, 32), (1, 8, 8)
u = UNet(1, 5, device=0, batch=4); u.load_state_dict(weights.generate_state_dict(0))
data = [positive_volume((8, 40, 48), 30 + t) for t in range(5)]
def frame(vol):
    vol = (vol / vol.max()).astype(np.float32)
    cur = np.zeros(tuple(s + 2 for s in vol.shape), np.uint32)
    fm = np.zeros((5,) + vol.shape, np.float32)
    seg.affinity_watershed_for_chunks(vol, cur, chunk, margin, unet=u, output_volume=fm)
    return cur.copy(), fm
ref = [frame(v) for v in data]
bad = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 200):
    for t, v in enumerate(data):
        lab, fm = frame(v)
        el, ef = np.array_equal(lab, ref[t][0]), np.array_equal(fm, ref[t][1])
        if not (el and ef):
            bad += 1
            print(f"pid {os.getpid()} iter {it} frame {t}: labels equal {el} (diff voxels {(lab != ref[t][0]).sum()}), feature map equal {ef} (max diff {np.abs(fm - ref[t][1]).max():.3e})", flush=True)
print(f"pid {os.getpid()} done, {bad} mismatches", flush=True)
