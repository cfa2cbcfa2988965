# one launch set of the U-Net (B chunks, default 64) for PMC runs: few launches, big kernels
import sys, os; sys.path.insert(0, os.getcwd())
import numpy as np, torch
from iterseg_amd import weights
from iterseg_amd.unet import UNet
from tests.synth import positive_volume
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
u = UNet(1, 5, batch=B); u.load_state_dict(weights.generate_state_dict(0))
vol = torch.from_numpy(positive_volume((64, 512, 512), 0)).cuda()     # 8 x 3 x 3 = 72 chunks
fm = torch.zeros((5, 64, 512, 512), device='cuda')
u.predict_volume(vol, (10, 256, 256), (1, 64, 64), feature_map=fm, first=0, count=B)
torch.cuda.synchronize()
print("done")
