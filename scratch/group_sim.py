"""Driver of scratch/group_sim.c (the CPU model of flood_group_kernel): exactness against the sequential oracle and the
parallelism the work-group window reaches, over inputs and stress settings."""
import sys, os; sys.path.insert(0, os.getcwd())
import ctypes as C, numpy as np, time, subprocess
os.makedirs("/tmp/sim", exist_ok=True)
from scratch.episode_sim import prep
from tests.synth import smooth_feature_map, blob_feature_map
subprocess.check_call("gcc -O2 -shared -fPIC -o /tmp/sim/libgsim.so scratch/group_sim.c -lm " + os.environ.get("GSIM_CFLAGS", ""), shell=True)
L = C.CDLL('/tmp/sim/libgsim.so')
f32p = np.ctypeslib.ndpointer(np.float32); i64p = np.ctypeslib.ndpointer(np.int64); u8p = np.ctypeslib.ndpointer(np.uint8); u32p = np.ctypeslib.ndpointer(np.uint32)
L.group_sim.argtypes = [f32p, i64p, C.c_int64, u8p, u32p, i64p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint, i64p, C.c_int]
L.group_sim.restype = C.c_int64


def run(name, P, wcap=512, tcap=192, run_limit=1664, drain_n=1024, bursts=3, seed=1, kmax=64):
    aff, sidx, mask, lab, pshape = P
    out = np.zeros(int(np.prod(pshape)), np.uint32); st = np.zeros(32, np.int64)
    t = time.time()
    L.group_sim(aff, sidx, len(sidx), mask, out, np.asarray(pshape, np.int64), wcap, tcap, run_limit, drain_n, bursts, seed, st, kmax)
    dt = time.time() - t
    ok = np.array_equal(out.reshape(pshape), lab)
    print(f"{name} W={wcap} T={tcap} run={run_limit} drain={drain_n} bursts={bursts} seed={seed} kmax={kmax}: exact={ok} pops {st[1]} sub-rounds {st[0]} "
          f"({st[1] / max(st[0], 1):.1f}/sub-round) executed {st[2]} squashed eps {st[12]} (bound {st[14]} lost {st[11]} slow {st[19]}) "
          f"forced {st[15]} reloads {st[16]} run-ins {st[3]} win-ins {st[9]} win-evict {st[8]} cold {st[6]} run-evict {st[7]} drains {st[5]} [{dt:.1f}s]", flush=True)
    return ok


if __name__ == "__main__":
    inputs = {}
    if os.path.exists("/tmp/sim/fm_c2.npy"):
        inputs["c2"] = np.load("/tmp/sim/fm_c2.npy")
    inputs["blobs"] = blob_feature_map((96, 96, 96), 1, 0.3, 3.0)
    inputs["smooth"] = smooth_feature_map((24, 96, 96), 0)
    q = blob_feature_map((64, 64, 64), 2, 0.3, 3.0); q[:3] = np.round(q[:3] * 4) / 4
    inputs["ties"] = q
    allok = True
    for name, fm in inputs.items():
        P = prep(fm)
        allok &= run(name, P)
        allok &= run(name, P, seed=7)
        allok &= run(name, P, wcap=1, tcap=6)
        allok &= run(name, P, wcap=3, tcap=6)
        allok &= run(name, P, wcap=512, tcap=12)
        allok &= run(name, P, wcap=64, tcap=48, run_limit=8, drain_n=4)
        allok &= run(name, P, wcap=512, tcap=192, run_limit=64, drain_n=16, bursts=1)
        allok &= run(name, P, kmax=1)
        allok &= run(name, P, kmax=4, seed=3)
        allok &= run(name, P, wcap=256, tcap=128, kmax=8)
    print("ALL EXACT" if allok else "MISMATCH")
