"""Development: the work-group flood kernel against the C oracle on one input, over a list of knob settings."""
import os, sys, subprocess, json
sys.path.insert(0, os.getcwd())
import numpy as np

CASES = {
    "default": {},
    "gwin1": {"ISX_FLOOD_GWIN": "1"},
    "gwin3": {"ISX_FLOOD_GWIN": "3"},
    "gwin64": {"ISX_FLOOD_GWIN": "64"},
    "nogroup": {"ISX_FLOOD_GROUP_MIN": "0"},
    "gwin65": {"ISX_FLOOD_GWIN": "65"},
    "gwin128": {"ISX_FLOOD_GWIN": "128"},
    "gwin256": {"ISX_FLOOD_GWIN": "256"},
    "bursts1": {"ISX_FLOOD_BURSTS": "1"},
    "glist48": {"ISX_FLOOD_GLIST": "48"},
    "nocold": {"ISX_FLOOD_COLD_MIN": "100000000"},
}

if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from iterseg_amd import watershed as ws
    from oracle import watershed as ows
    from tests.synth import smooth_feature_map, blob_feature_map
    kind, shape, seed = sys.argv[2], tuple(int(v) for v in sys.argv[3].split(",")), int(sys.argv[4])
    fm = blob_feature_map(shape, seed) if kind == "blobs" else smooth_feature_map(shape, seed)
    pshape = tuple(s + 2 for s in shape)
    ref = np.zeros(pshape, np.uint32)
    ows.segment_output_image(fm, out=ref.ravel())
    cur = np.zeros(pshape, np.uint32)
    ws.segment_output_image(fm, (0, 1, 2), 4, 3, out=cur.ravel())
    bad = np.argwhere(cur != ref)
    print(f"  labelled ref {(ref > 0).sum()} got {(cur > 0).sum()} mismatches {len(bad)} first {bad[:3].tolist()} "
          f"got {[int(cur[tuple(b)]) for b in bad[:3]]} ref {[int(ref[tuple(b)]) for b in bad[:3]]}", flush=True)
    sys.exit(0)

kind, shape, seed = (sys.argv[1:4] + ["smooth", "24,96,80", "0"][len(sys.argv) - 1:])[:3]
for name in (sys.argv[4].split(",") if len(sys.argv) > 4 else CASES):
    env = dict(os.environ, ISX_FLOOD_STATS="1", **CASES[name])
    print(f"== {name} {CASES[name]}", flush=True)
    r = subprocess.run([sys.executable, __file__, "child", kind, shape, str(seed)], env=env, capture_output=True, text=True, timeout=300)
    print(r.stdout[-1500:], r.stderr[-1500:].replace("/opt/amdgpu/share/libdrm/amdgpu.ids: No such file or directory\n", ""), flush=True)
