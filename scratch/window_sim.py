"""Experiment driver for scratch/window_sim.c (ordered speculative window)."""
import sys, os; sys.path.insert(0, os.getcwd())
import ctypes as C, numpy as np, time, subprocess
from scratch.episode_sim import prep, blob_feature_map, smooth_feature_map
subprocess.check_call("gcc -O2 -shared -fPIC -o /tmp/sim/libwsim.so scratch/window_sim.c -lm", shell=True)
L = C.CDLL('/tmp/sim/libwsim.so')
f32p = np.ctypeslib.ndpointer(np.float32); i64p=np.ctypeslib.ndpointer(np.int64); u8p=np.ctypeslib.ndpointer(np.uint8); u32p=np.ctypeslib.ndpointer(np.uint32)
L.window_sim.argtypes=[f32p,i64p,C.c_int64,u8p,u32p,i64p,C.c_int,C.c_int,C.c_int,i64p,C.c_int,C.c_int]; L.window_sim.restype=C.c_int64
def run(name, P, W, LCAP, dac, LANES=64, T=32):
    aff, sidx, mask, lab, pshape = P
    out = np.zeros(int(np.prod(pshape)), np.uint32); st = np.zeros(32, np.int64)
    t=time.time(); L.window_sim(aff, sidx, len(sidx), mask, out, np.asarray(pshape,np.int64), W, LCAP, dac, st, LANES, T); dt=time.time()-t
    ok = np.array_equal(out.reshape(pshape), lab)
    print(f"{name} W={W} LANES={LANES} T={T} detect_at_commit={dac}: exact={ok} pops {st[1]} subrounds {st[0]} pops/subround {st[1]/st[0]:.1f} lane-util {st[1]/max(st[9],1):.2f} squashed eps {st[3]} pops {st[4]} "
          f"events lost {st[5]} bound {st[6]} evict {st[7]} mid-ins {st[8]} [{dt:.1f}s]", flush=True)
if __name__ == "__main__":
    what = sys.argv[1]
    if what.endswith(".npy"): fm = np.load(what)
    elif what == "blobs": fm = blob_feature_map((int(sys.argv[2]),)*3, 0, float(os.environ.get("VF", "0.3")))
    else: fm = smooth_feature_map((32,128,128), 0)
    P = prep(fm)
    for W, LANES, T in [(128, 64, 64), (192, 64, 48), (160, 64, 64), (128, 64, 96), (256, 64, 48)]:
        run(what, P, W, 0, 0, LANES, T)
