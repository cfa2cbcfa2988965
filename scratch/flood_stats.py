import sys, os; sys.path.insert(0, os.getcwd())
import ctypes as C, numpy as np, time
from oracle import watershed as ows
from tests.synth import smooth_feature_map, blob_feature_map
L = C.CDLL('scratch/libfs.so')
f32p = np.ctypeslib.ndpointer(np.float32); i64p=np.ctypeslib.ndpointer(np.int64); u8p=np.ctypeslib.ndpointer(np.uint8); u32p=np.ctypeslib.ndpointer(np.uint32)
L.flood_stats.argtypes=[f32p,i64p,C.c_int64,u8p,u32p,i64p,i64p,C.c_int,C.POINTER(C.c_int64)]; L.flood_stats.restype=C.c_int64
def run(name, fm):
    shape = fm.shape[1:]; pshape = tuple(s+2 for s in shape)
    lab = np.zeros(pshape, np.uint32)
    seg, seeds, mask = ows.segment_output_image(fm, out=lab.ravel())
    aff = np.empty((3,)+pshape, np.float32); ows.lib().orc_normalise_pad(np.ascontiguousarray(fm[:3]), np.asarray(shape,np.int64), aff)
    sidx = np.ravel_multi_index(tuple((seeds+1).T), pshape).astype(np.int64)
    out = np.zeros(int(np.prod(pshape)), np.uint32); hist = np.zeros(24, np.int64); mh = C.c_int64()
    t=time.time(); rounds = L.flood_stats(aff.reshape(3,-1), sidx, len(sidx), np.ascontiguousarray(mask,np.uint8).ravel(), out, np.asarray(pshape,np.int64), hist, 24, C.byref(mh)); dt=time.time()-t
    pops = int((out>0).sum())
    assert np.array_equal(out.reshape(pshape), lab)
    print(f"{name}: voxels {np.prod(shape)} mask {int(mask.sum())} seeds {len(sidx)} pops {pops} rounds {rounds} avg run {pops/max(rounds,1):.1f} max heap {mh.value} cpu {dt:.2f}s ({pops/dt/1e6:.1f} Mpops/s)")
    print("   run-length histogram (log2 bins):", hist[:16].tolist())
run("smooth sigma1.5 32x128x128", smooth_feature_map((32,128,128), 0))
run("smooth sigma3 32x128x128", smooth_feature_map((32,128,128), 1, sigma=3.0))
run("blobs 48x160x160", blob_feature_map((48,160,160), 0))
