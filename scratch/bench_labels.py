import sys, os, time; sys.path.insert(0, os.getcwd())
import numpy as np, torch
from scipy import ndimage as ndi
from iterseg_amd import labels as gl
rng = np.random.default_rng(0)
shape = (64, 512, 512)
f = ndi.gaussian_filter(rng.random(shape), (1.0, 3.0, 3.0))
lab, n = ndi.label(f > np.quantile(f, 0.85))
dl = torch.from_numpy(lab.astype(np.int32)).cuda()
for _ in range(2):
    torch.cuda.synchronize(); t = time.perf_counter()
    out = gl.get_training_labels(dl, ('z-1', 'y-1', 'x-1', 'centreness'), (4, 1, 1))
    torch.cuda.synchronize(); dt = time.perf_counter() - t
print(f"{shape} labels, {n} objects: 4 target channels in {dt*1e3:.2f} ms ({np.prod(shape)/dt/1e6:.0f} M voxels/s)")
