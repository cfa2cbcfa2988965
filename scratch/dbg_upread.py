"""Tail layer with up3 up-sampled on read against the materialised form."""
import sys, os; sys.path.insert(0, os.getcwd())
import numpy as np
from iterseg_amd import weights
from iterseg_amd.unet import UNet
prec = sys.argv[1] if len(sys.argv) > 1 else "split-f16"
sd = weights.generate_state_dict(0)
def mk(v):
    os.environ["ISX_UP_ON_READ"] = str(v)
    u = UNet(1, 5, batch=2, precision=prec); u.load_state_dict(sd); return u
a, b = mk(1), mk(0)
rng = np.random.default_rng(0)
for shp in [(4, 32, 32), (10, 64, 80), (10, 256, 256)]:
    x = rng.random((1, 1) + shp, dtype=np.float32)
    ya, yb = a(x), b(x)
    ta, tb = a.tap("c8_0.conv0"), b.tap("c8_0.conv0")
    d = np.abs(ta - tb)
    print(shp, "y max diff", float(np.abs(ya - yb).max()), "c8_0.conv0 diff", float(d.max()), "at", np.unravel_index(d.argmax(), d.shape), "scale", float(np.abs(tb).max()),
          "up3 tap diff", float(np.abs(a.tap("up3.cropped") - b.tap("up3.cropped")).max()) if np.prod(shp) < 300000 else "-", flush=True)
