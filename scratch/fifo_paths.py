"""Which drain paths does each stress mode hit on the tie-heavy test input?  (ISX_FLOOD_STATS prints to stderr.)"""
import os, sys; sys.path.insert(0, os.getcwd())
import numpy as np
os.environ["ISX_FLOOD_STATS"] = "1"
mode = sys.argv[1]
env = {"default": {},
       "cold-stress": {"ISX_FLOOD_COOP_MIN": "1", "ISX_FLOOD_COLD_MIN": "1", "ISX_FLOOD_RUN_LIMIT": "96", "ISX_FLOOD_DRAIN": "48"},
       "cold-tiny": {"ISX_FLOOD_COOP_MIN": "1", "ISX_FLOOD_COLD_MIN": "1", "ISX_FLOOD_RUN_LIMIT": "8", "ISX_FLOOD_DRAIN": "4"}}[mode]
os.environ.update(env)
from iterseg_amd import watershed as ws
from oracle import watershed as ows
from tests.synth import blob_feature_map, smooth_feature_map
for kind, shape, seed in (("blobs", (40, 128, 128), 2), ("smooth", (24, 96, 80), 0)):
    fm = blob_feature_map(shape, seed) if kind == "blobs" else smooth_feature_map(shape, seed)
    pshape = tuple(s + 2 for s in shape)
    ref = np.zeros(pshape, np.uint32); ows.segment_output_image(fm, out=ref.ravel())
    cur = np.zeros(pshape, np.uint32); ws.segment_output_image(fm, (0, 1, 2), 4, 3, out=cur.ravel())
    print(mode, kind, "equal", np.array_equal(cur, ref), flush=True)
