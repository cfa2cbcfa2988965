import sys, os, time; sys.path.insert(0, os.getcwd())
import numpy as np, torch
from iterseg_amd import weights
from iterseg_amd.unet import UNet
from tests.synth import positive_volume
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
u = UNet(1, 5, batch=B); u.load_state_dict(weights.generate_state_dict(0))
shape = (64, 512, 512)
vol = torch.from_numpy(positive_volume(shape, 0)).cuda()
fm = torch.zeros((5,) + shape, device='cuda')
chunk, margin = (10, 256, 256), (1, 64, 64)
u.predict_volume(vol, chunk, margin, feature_map=fm); torch.cuda.synchronize()
u.ctx.prof_enable(True)
t = time.perf_counter(); reps = 2
for _ in range(reps): u.predict_volume(vol, chunk, margin, feature_map=fm)
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / reps
mf, df = bench.conv_flops_per_chunk(chunk)
p = {k: u.ctx.prof_get(k) for k in ("conv_mfma", "conv_direct", "elementwise")}
nch = 72
print(f"batch {B}: wall {dt*1e3:.1f} ms / 72 chunks = {dt/nch*1e3:.3f} ms/chunk; conv_mfma {p['conv_mfma'][0]/reps:.1f} ms -> {mf*nch/(p['conv_mfma'][0]/reps*1e-3)/1e12:.1f} TF/s ({mf*nch/(p['conv_mfma'][0]/reps*1e-3)/1e12/157.3*100:.1f}%); direct {p['conv_direct'][0]/reps:.1f} ms; elementwise {p['elementwise'][0]/reps:.1f} ms")
