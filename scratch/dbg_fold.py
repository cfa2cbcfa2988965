"""Folded up-convolution against the unfused form: taps of the decoder's first convolutions and the outputs."""
import sys, os, time; sys.path.insert(0, os.getcwd())
import numpy as np
from iterseg_amd import weights
from iterseg_amd.unet import UNet
shapes = [(4, 32, 32), (10, 64, 80), (10, 256, 256)]
prec = sys.argv[1] if len(sys.argv) > 1 else "split-f16"
sd = weights.generate_state_dict(0)
def mk(fold):
    os.environ["ISX_UP_FOLD"] = str(fold)
    u = UNet(1, 5, batch=2, precision=prec); u.load_state_dict(sd); return u
a, b = mk(1), mk(0)
rng = np.random.default_rng(0)
for shp in shapes:
    x = rng.random((1, 1) + shp, dtype=np.float32)
    ya, yb = a(x), b(x)
    print(shp, "y max diff", float(np.abs(ya - yb).max()), flush=True)
    for name in ["c5_0.conv0", "c6_0.conv0", "c7_0.conv0", "c8_0.conv0", "c8_0.conv1"]:
        ta, tb = a.tap(name), b.tap(name)
        d = np.abs(ta - tb)
        inner = d[:, 1:-1, 2:-2, 2:-2].max() if min(d.shape[1:]) > 4 else -1
        if d.max() > 1e-4: print("      worst at", np.unravel_index(d.argmax(), d.shape))
        print("   ", name, ta.shape, "max diff", float(d.max()), "interior", float(inner), "scale", float(np.abs(tb).max()), flush=True)
    print("   up1/up2.cropped tap diff", float(np.abs(a.tap("up1.cropped") - b.tap("up1.cropped")).max()), float(np.abs(a.tap("up2.cropped") - b.tap("up2.cropped")).max()))
