# why is the structured watershed slower inside bench.py than alone?  (a) alone, (b) after a U-Net + config-2 watershed, (c) with the profiler on
import sys, os, time; sys.path.insert(0, os.getcwd())
import numpy as np, torch
from iterseg_amd import watershed as ws, weights
from iterseg_amd.unet import UNet
from tests.synth import blob_feature_map_tiled, positive_volume
n = 512
dfm = torch.from_numpy(blob_feature_map_tiled(n, 0, 0.12, 3.0)).cuda()
lab = torch.zeros((n + 2,) * 3, dtype=torch.int32, device="cuda")
def run(tag, reps=3, prof=False):
    ts = []
    for _ in range(reps):
        lab.zero_()
        if prof: c = ws.get_context(0); c.prof_enable(False); c.prof_enable(True)
        torch.cuda.synchronize(); t = time.time()
        ws.segment_output_image(dfm, (0, 1, 2), 4, 3, out=lab); torch.cuda.synchronize(); ts.append(time.time() - t)
    print(tag, [round(x * 1e3, 1) for x in ts], flush=True)
run("alone")
run("alone, profiler on", prof=True)
ws.get_context(0).prof_enable(False)
u = UNet(1, 5, device=0, batch=64); u.load_state_dict(weights.generate_state_dict(0))
vol = torch.from_numpy(positive_volume((64, 512, 512), 0)).cuda()
fm = u.predict_volume(vol, (10, 256, 256), (1, 64, 64)).clone()
run("after a U-Net predict")
l2 = torch.zeros((66, 514, 514), dtype=torch.int32, device="cuda")
ws.segment_output_image(fm, (0, 1, 2), 4, 3, out=l2); torch.cuda.synchronize()
run("after the config-2 watershed")
