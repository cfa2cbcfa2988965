"""Structured 512^3 (tiled blobs): watershed time of the whole volume against the same volume with every component but the
largest one masked out (thresholding channel zeroed outside it)."""
import sys, os, time; sys.path.insert(0, os.getcwd())
import numpy as np, torch
from iterseg_amd import watershed as ws
from tests.synth import blob_feature_map_tiled
n = 512
for seed in [int(a) for a in sys.argv[1:]] or [0]:
    fm = blob_feature_map_tiled(n, seed, 0.12, 3.0)
    dfm = torch.from_numpy(fm).cuda()
    lab = torch.zeros((n + 2,) * 3, dtype=torch.int32, device="cuda")
    seg, seeds, mask = ws.segment_output_image(dfm, (0, 1, 2), 4, 3, out=lab)
    roots, sizes = ws.connected_components(mask)
    sz = sizes.flatten().to(torch.int64); big = int(sz.argmax()); nbig = int(sz[big])
    ncomp = int((sz > 0).sum()); nmid = int(((sz >= 512)).sum())
    keep = (roots.view(torch.int32).flatten() == np.int32(big)).view(mask.shape)[1:-1, 1:-1, 1:-1] if isinstance(roots, torch.Tensor) else None
    dfm2 = dfm.clone(); dfm2[3] = torch.where(keep, dfm[3], torch.zeros_like(dfm[3]))
    ctx = ws.get_context()
    def timeit(x):
        ts = []
        for _ in range(3):
            lab.zero_(); ctx.prof_enable(False); ctx.prof_enable(True); torch.cuda.synchronize(); t = time.time()
            ws.segment_output_image(x, (0, 1, 2), 4, 3, out=lab); torch.cuda.synchronize(); ts.append((time.time() - t, ctx.prof_get("ws_flood")[0]))
        return min(ts)
    a, b = timeit(dfm), timeit(dfm2)
    print(f"seed {seed}: components {ncomp} (>= 512 voxels: {nmid}), largest {nbig}; whole volume {a[0]*1e3:.1f} ms (flood {a[1]:.1f}); largest component alone {b[0]*1e3:.1f} ms (flood {b[1]:.1f})", flush=True)
