# Development: compare the commit order of the window flood (ISX_FLOOD_CHECK build, ISX_FLOOD_TRACE) with the oracle's pop order.
import sys, os, time, ctypes as C; sys.path.insert(0, os.getcwd())
import numpy as np, torch
from scipy import ndimage
from iterseg_amd import watershed as ws
from oracle import watershed as ows
from tests.synth import blob_feature_map_tiled
n = 512
fm = blob_feature_map_tiled(n, 0, 0.12, 3.0)
pshape = (n + 2,) * 3
L = ows.lib()
trace = np.zeros(40_000_000, np.int64)
L.orc_set_trace.argtypes = [C.c_void_p, C.c_int64]; L.orc_set_trace.restype = None
L.orc_trace_count.restype = C.c_int64
L.orc_set_trace(trace.ctypes.data, trace.size)
ref = np.zeros(pshape, np.uint32); ows.segment_output_image(fm, out=ref.ravel(), threads=1)
nt = L.orc_trace_count(); L.orc_set_trace(None, 0); trace = trace[:nt]
print("oracle pops", nt, flush=True)
os.environ["ISX_FLOOD_BURSTS"] = sys.argv[1] if len(sys.argv) > 1 else "3"
tpath = "/tmp/isx_trace.bin"; os.environ["ISX_FLOOD_TRACE"] = tpath
lab = torch.zeros(pshape, dtype=torch.int32, device="cuda")
ws.segment_output_image(torch.from_numpy(fm).cuda(), (0, 1, 2), 4, 3, out=lab)
got = lab.cpu().numpy().astype(np.uint32)
bad = np.argwhere(got != ref)
print("mismatches", len(bad), flush=True)
if len(bad) == 0: sys.exit(0)
raw = np.fromfile(tpath, np.uint8)
hdr = raw[:24].view(np.uint64); n_comp, n_seeds, total = (int(v) for v in hdr)
o = 24
comp_off = raw[o:o + 8 * (n_comp + 1)].view(np.uint64).astype(np.int64); o += 8 * (n_comp + 1)
comp_first = raw[o:o + 4 * n_comp].view(np.uint32).astype(np.int64); o += 4 * n_comp
keys = raw[o:o + 8 * n_seeds].view(np.uint64); o += 8 * n_seeds
dlog = raw[o:o + 4 * (total + 1)].view(np.uint32)
dlogE = raw[o + 4 * (total + 1):o + 8 * (total + 1)].view(np.uint32)
dlogN = raw[o + 8 * (total + 1):o + 12 * (total + 1)].view(np.uint32)
seed_idx = (keys & np.uint64(0xFFFFFFFF)).astype(np.int64)
cc, ncc = ndimage.label(ref > 0)
pos = np.full(ref.size, -1, np.int64); pos[trace] = np.arange(nt)
seen = set()
for z, y, x in bad:
    cid = cc[z, y, x]
    if cid in seen: continue
    seen.add(cid)
    # kernel component: the one whose first seed lies in this component
    first_seed_vox = seed_idx[comp_first]
    kc = int(np.nonzero(cc.ravel()[first_seed_vox] == cid)[0][0])
    csize = int(comp_off[kc + 1] - comp_off[kc])
    dl = dlog[comp_off[kc]:comp_off[kc] + csize]
    cnt = int(dl[csize - 1]); roots = dl[:min(cnt, csize - 1)].astype(np.int64)
    rp = pos[roots]
    print(f"component cc={cid} kernel comp {kc} size {csize}, committed episodes {cnt}, ref pops in comp {(cc.ravel()[trace] == cid).sum()}")
    d = np.diff(rp)
    viol = np.nonzero(d <= 0)[0]
    print("  order violations", len(viol), "unpopped roots", int((rp < 0).sum()))
    for i in viol[:5]:
        lo, hi = max(0, i - 3), min(len(roots), i + 5)
        for j in range(lo, hi):
            v = int(roots[j]); zz, yy, xx = np.unravel_index(v, pshape)
            print(f"   {'>>' if j == i + 1 else '  '} episode {j}: root {(zz, yy, xx)} refpos {rp[j]} got {got[zz,yy,xx]} ref {ref[zz,yy,xx]}")

    # per-episode popped sets against the oracle's segments
    comp_trace = trace[cc.ravel()[trace] == cid]
    cpos = {int(v): i for i, v in enumerate(comp_trace)}
    ent = dlogE[comp_off[kc]:comp_off[kc] + csize]; npn = dlogN[comp_off[kc]:comp_off[kc] + csize]
    eo = 0; shown = 0
    for j in range(len(roots)):
        npj = int(npn[j] & 0xFF); sub = int(npn[j] >> 8)
        es = ent[eo:eo + npj]; eo += npj
        a = cpos[int(roots[j])]; b = cpos[int(roots[j + 1])] if j + 1 < len(roots) else len(comp_trace)
        refp = [int(v) for v in comp_trace[a + 1:b]]
        gotp = [int(v & 0x7FFFFFFF) for v in es if v & 0x80000000]
        if set(refp) != set(gotp):
            print(f"  episode {j} (sub-round {sub}) root {np.unravel_index(int(roots[j]), pshape)} np {npj}: ref pops {len(refp)} got pops {len(gotp)}")
            print("     ref:", [tuple(int(q) for q in np.unravel_index(v, pshape)) for v in refp][:12])
            print("     got:", [tuple(int(q) for q in np.unravel_index(v, pshape)) for v in gotp][:12])
            print("     discovered:", [(tuple(int(q) for q in np.unravel_index(int(v & 0x7FFFFFFF), pshape)), bool(v >> 31)) for v in es][:16])
            for jj in range(max(0, j - 2), min(len(roots), j + 3)):
                print(f"       ep {jj}: sub {int(npn[jj] >> 8)} np {int(npn[jj] & 0xFF)} root {np.unravel_index(int(roots[jj]), pshape)} refpos {cpos[int(roots[jj])]}")
            shown += 1
            if shown >= 3: break
