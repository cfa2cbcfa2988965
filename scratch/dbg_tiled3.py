import sys, os, time; sys.path.insert(0, os.getcwd())
import numpy as np, torch
from iterseg_amd import watershed as ws
from oracle import watershed as ows
from tests.synth import blob_feature_map_tiled
n = 512
fm = blob_feature_map_tiled(n, 0, 0.12, 3.0)
pshape = (n + 2,) * 3
ref = np.zeros(pshape, np.uint32); ows.segment_output_image(fm, out=ref.ravel(), threads=16)
aff = np.empty((3,) + pshape, np.float32); ows.lib().orc_normalise_pad(np.ascontiguousarray(fm[:3]), np.asarray(fm.shape[1:], np.int64), aff)
os.environ["ISX_FLOOD_BURSTS"] = "3"
lab = torch.zeros(pshape, dtype=torch.int32, device="cuda")
ws.segment_output_image(torch.from_numpy(fm).cuda(), (0, 1, 2), 4, 3, out=lab)
got = lab.cpu().numpy().astype(np.uint32)
bad = np.argwhere(got != ref)
print("mismatches", len(bad))
offs = [(-1,0,0),(0,-1,0),(0,0,-1),(0,0,1),(0,1,0),(1,0,0)]; axs = [0,1,2,2,1,0]
for z, y, x in bad[:6]:
    print(f"voxel {(z,y,x)} got {got[z,y,x]} ref {ref[z,y,x]}")
    for k, (dz, dy, dx) in enumerate(offs):
        nz, ny, nx = z+dz, y+dy, x+dx
        src = (z, y, x) if k < 3 else (nz, ny, nx)
        print(f"   nb{k} {(nz,ny,nx)} got {got[nz,ny,nx]} ref {ref[nz,ny,nx]} edge {aff[axs[k]][src]!r}")
