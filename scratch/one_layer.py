# one launch set of the U-Net (8 chunks) for PMC runs: few launches, big kernels
import sys, os; sys.path.insert(0, os.getcwd())
import numpy as np, torch
from iterseg_amd import weights
from iterseg_amd.unet import UNet
from tests.synth import positive_volume
u = UNet(1, 5, batch=8); u.load_state_dict(weights.generate_state_dict(0))
vol = torch.from_numpy(positive_volume((24, 512, 512), 0)).cuda()     # z starts 0,8,14 -> 3 x 3 x 3 = 27 chunks
fm = torch.zeros((5, 24, 512, 512), device='cuda')
u.predict_volume(vol, (10, 256, 256), (1, 64, 64), feature_map=fm, first=0, count=8)
torch.cuda.synchronize()
print("done")
