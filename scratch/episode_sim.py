"""Experiment driver for scratch/episode_sim.c: how many pops does one exact speculative round commit?"""
import sys, os; sys.path.insert(0, os.getcwd())
import ctypes as C, numpy as np, time, subprocess
from scipy import ndimage as ndi
from oracle import watershed as ows
from tests.synth import smooth_feature_map, blob_feature_map
subprocess.check_call("gcc -O2 -shared -fPIC -o /tmp/sim/libes.so scratch/episode_sim.c -lm", shell=True)
L = C.CDLL('/tmp/sim/libes.so')
f32p = np.ctypeslib.ndpointer(np.float32); i64p=np.ctypeslib.ndpointer(np.int64); u8p=np.ctypeslib.ndpointer(np.uint8); u32p=np.ctypeslib.ndpointer(np.uint32)
L.episode_sim.argtypes=[f32p,i64p,C.c_int64,u8p,u32p,i64p,C.c_int,C.c_int,C.c_int,i64p]; L.episode_sim.restype=C.c_int64
def prep(fm, largest=True):
    shape = fm.shape[1:]; pshape = tuple(s+2 for s in shape)
    lab = np.zeros(pshape, np.uint32)
    seg, seeds, mask = ows.segment_output_image(fm, out=lab.ravel())
    aff = np.empty((3,)+pshape, np.float32); ows.lib().orc_normalise_pad(np.ascontiguousarray(fm[:3]), np.asarray(shape,np.int64), aff)
    if largest:
        cc, n = ndi.label(mask)
        sz = np.bincount(cc.ravel()); sz[0] = 0
        big = int(sz.argmax())
        keep = cc == big
        pc = tuple((seeds+1).T)
        sel = keep[pc]
        # labels of the kept seeds stay as they are in the oracle (position in the full list): remap for comparison
        seeds_k = seeds[sel]
        remap = np.zeros(len(seeds)+1, np.uint32); remap[np.nonzero(sel)[0]+1] = np.arange(1, sel.sum()+1, dtype=np.uint32)
        lab = np.where(keep, remap[lab], 0).astype(np.uint32)
        mask = keep; seeds = seeds_k
    sidx = np.ravel_multi_index(tuple((seeds+1).T), pshape).astype(np.int64)
    return aff.reshape(3,-1), sidx, np.ascontiguousarray(mask, np.uint8).ravel(), lab, pshape
def run(name, P, R, K, LCAP):
    aff, sidx, mask, lab, pshape = P
    out = np.zeros(int(np.prod(pshape)), np.uint32); st = np.zeros(32, np.int64)
    t=time.time(); L.episode_sim(aff, sidx, len(sidx), mask, out, np.asarray(pshape,np.int64), R, K, LCAP, st); dt=time.time()-t
    ok = np.array_equal(out.reshape(pshape), lab)
    rounds, pops, eps, subs = st[0], st[1], st[2], st[3]
    print(f"{name} R={R} K={K} LCAP={LCAP}: exact={ok} mask {int(mask.sum())} seeds {len(sidx)} pops {pops} rounds {rounds} pops/round {pops/rounds:.1f} eps/round {eps/rounds:.1f} subrounds/round {subs/rounds:.2f} "
          f"cuts: lost {st[4]} F-rule {st[5]} trunc {st[6]} wasted pops {st[7]} ({st[7]/max(pops,1):.2f}x) resid/push {st[9]/max(st[8],1):.2f} eplen hist {st[10:20].tolist()} [{dt:.1f}s]", flush=True)
if __name__ == "__main__" and sys.argv[0].endswith("episode_sim.py"):
    what = sys.argv[1]
    if what.endswith(".npy"): fm = np.load(what)
    elif what == "blobs": fm = blob_feature_map((64,)*3 if len(sys.argv) < 3 else (int(sys.argv[2]),)*3, 0, float(os.environ.get("VF", "0.3")))
    else: fm = smooth_feature_map((32,128,128), 0)
    P = prep(fm)
    for R, K, LC in [(64, 1, 64), (64, 4, 8), (64, 8, 8), (64, 16, 16), (256, 4, 8), (256, 8, 8), (256, 16, 16), (1024, 8, 8)]:
        run(what, P, R, K, LC)
