import sys, os, time; sys.path.insert(0, os.getcwd())
import numpy as np, torch
from iterseg_amd import watershed as ws
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ctx = ws.get_context()
g = torch.Generator(device='cuda'); g.manual_seed(0)
cases = {
    "noise": torch.rand((n, n, n), device='cuda', generator=g),
    "saturated": (torch.rand((n, n, n), device='cuda', generator=g) > 0.5).float(),
    "const_one_dot": torch.full((n, n, n), 0.5, device='cuda'),
}
cases["const_one_dot"][5, 5, 5] = 0.7
for name, vol in cases.items():
    for smooth in (False, True):
        for rep in range(2):
            ctx.prof_enable(False); ctx.prof_enable(True)
            torch.cuda.synchronize(); t = time.perf_counter()
            c = ws._get_centroids(vol, gaussian=smooth)
            torch.cuda.synchronize(); dt = time.perf_counter() - t
        print(f"{name:14s} smooth={smooth}: {dt*1e3:8.2f} ms  filters {ctx.prof_get('ws_filters')[0]:.2f} ms  peaks {len(c)}", flush=True)
