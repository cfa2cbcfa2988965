"""Development: one big component with MANY seeds (seed queue spills from the window into the run and the cold tier) through the
work-group flood kernel against the C oracle."""
import os, sys
sys.path.insert(0, os.getcwd())
os.environ.setdefault("ISX_FLOOD_BURSTS", "515")          # 0x200: skip the serial continuation, report instead
os.environ.setdefault("ISX_FLOOD_STATS", "1")
os.environ.setdefault("ISX_FLOOD_GROUP_MIN", "1")
os.environ.setdefault("ISX_FLOOD_COOP_MIN", "1")
os.environ.setdefault("ISX_FLOOD_COLD_MIN", "1")
import numpy as np
from iterseg_amd import watershed as ws
from oracle import watershed as ows
from tests.synth import blob_feature_map
shape = (32, 96, 96)
fm = blob_feature_map(shape, 3, 0.5, 4.0)
aff = np.pad(fm[:3] / fm[:3].max(axis=(1, 2, 3)).reshape(-1, 1, 1, 1), ((0, 0), (1, 1), (1, 1), (1, 1))).astype(np.float32)
if len(sys.argv) > 2 and sys.argv[2] == "q":
    aff = (np.round(aff * 4) / 4).astype(np.float32)
mask = np.pad(fm[3] > 0.3, 1)
every = int(sys.argv[1]) if len(sys.argv) > 1 else 7
idx = np.argwhere(mask)
rng = np.random.default_rng(0)
seeds = idx[rng.permutation(len(idx))[: len(idx) // every]]
ref = ows.affinity_watershed(aff, seeds, mask).astype(np.int64)
out = np.zeros(mask.size, np.int32)
try:
    got = ws.affinity_watershed(aff, seeds, mask, out=out).astype(np.int64)
    print(f"mask {mask.sum()} seeds {len(seeds)} mismatches {(got != ref).sum()} unlabelled-in-mask {(mask & (got == 0)).sum()}")
except Exception as e:
    print("error:", e)
