import sys, os, time; sys.path.insert(0, os.getcwd())
import numpy as np, torch
from iterseg_amd import watershed as ws
from tests.synth import smooth_feature_map, blob_feature_map
kind = sys.argv[1]; n = int(sys.argv[2]); 
shape = (n, n, n) if len(sys.argv) < 4 else (int(sys.argv[3]), n, n)
t = time.time()
vf = float(os.environ.get("VF", "0.3")); bs = float(os.environ.get("BSIG", "3.0"))
fm = blob_feature_map(shape, 0, vf, bs) if kind == "blobs" else smooth_feature_map(shape, 0, 1.5)
print(f"{kind} {shape}: built in {time.time()-t:.1f}s", flush=True)
dfm = torch.from_numpy(fm).cuda()
ctx = ws.get_context()
pshape = tuple(s + 2 for s in shape)
for rep in range(2):
    lab = torch.zeros(pshape, dtype=torch.int32, device='cuda')
    ctx.prof_enable(False); ctx.prof_enable(True)
    torch.cuda.synchronize(); t = time.perf_counter()
    seg, seeds, mask = ws.segment_output_image(dfm, (0, 1, 2), 4, 3, out=lab)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    p = {k: ctx.prof_get(k)[0] for k in ("ws_filters", "ws_ccl", "ws_flood")}
    nl = int((lab > 0).sum())
    print(f"  rep {rep}: total {dt*1e3:.1f} ms; {p}; seeds {len(seeds)} labelled {nl} -> flood {nl/max(p['ws_flood'],1e-9)/1e3:.2f} Mpops/s; e2e {np.prod(shape)/dt/1e6:.1f} Mvox/s", flush=True)
roots, sizes = ws.connected_components(mask)
sz = sizes.flatten()[sizes.flatten() > 0].cpu().numpy() if hasattr(sizes, 'cpu') else sizes[sizes > 0]
sz = np.sort(np.asarray(sz).astype(np.int64))[::-1]
print("  components:", len(sz), "largest:", sz[:5].tolist())
