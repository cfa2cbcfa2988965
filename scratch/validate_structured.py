"""Exactness sweep of the window flood on structured inputs (tiled blobs) against the C oracle, several seeds / volume
fractions / burst counts."""
import sys, os, time; sys.path.insert(0, os.getcwd())
import numpy as np, torch
from iterseg_amd import watershed as ws
from oracle import watershed as ows
from tests.synth import blob_feature_map_tiled
n = int(sys.argv[1]); seeds = [int(s) for s in sys.argv[2].split(",")]; vfs = [float(v) for v in sys.argv[3].split(",")]
bursts = sys.argv[4].split(",")
pshape = (n + 2,) * 3
for seed in seeds:
    for vf in vfs:
        fm = blob_feature_map_tiled(n, seed, vf, 3.0)
        ref = np.zeros(pshape, np.uint32); ows.segment_output_image(fm, out=ref.ravel(), threads=16)
        dfm = torch.from_numpy(fm).cuda()
        res = []
        for b in bursts:
            os.environ["ISX_FLOOD_BURSTS"] = b
            lab = torch.zeros(pshape, dtype=torch.int32, device="cuda")
            t = time.time(); ws.segment_output_image(dfm, (0, 1, 2), 4, 3, out=lab); torch.cuda.synchronize(); dt = time.time() - t
            res.append((b, int((lab.cpu().numpy().astype(np.uint32) != ref).sum()), round(dt, 3)))
        print(f"n {n} seed {seed} vf {vf}: labelled {int((ref > 0).sum())}  (bursts, mismatches, s): {res}", flush=True)
