"""Development: the tiled structured volume at edge n through the flood (group kernel on by default) against the C oracle."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from iterseg_amd import watershed as ws
from oracle import watershed as ows
from tests.synth import blob_feature_map_tiled
n, seed = int(sys.argv[1]), int(sys.argv[2])
fm = blob_feature_map_tiled(n, seed, 0.12, 3.0)
pshape = (n + 2,) * 3
ref = np.zeros(pshape, np.uint32)
ows.segment_output_image(fm, out=ref.ravel(), threads=16)
dfm = torch.from_numpy(fm).cuda()
for rep in range(2):
    lab = torch.zeros(pshape, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize(); t = time.time()
    _, _, mask = ws.segment_output_image(dfm, (0, 1, 2), 4, 3, out=lab)
    torch.cuda.synchronize(); dt = time.time() - t
    got = lab.cpu().numpy().astype(np.uint32)
    _, sizes = ws.connected_components(mask)
    sz = torch.sort(torch.as_tensor(sizes).flatten(), descending=True).values[:5].tolist()
    print(f"n={n} seed={seed} rep {rep}: {dt*1e3:.1f} ms labelled {(ref>0).sum()} mismatches {(got != ref).sum()} largest comps {sz}", flush=True)
