"""Watershed time on the tiled structured 512^3 volume for a few ISX_FLOOD_* settings (argv: 'K=V,K=V;K=V;...')."""
import sys, os, time; sys.path.insert(0, os.getcwd())
import numpy as np, torch
from iterseg_amd import watershed as ws
from tests.synth import blob_feature_map_tiled
n = 512
cfgs = sys.argv[1].split(";") if len(sys.argv) > 1 else [""]
seeds = [int(s) for s in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["0"])]
for seed in seeds:
    dfm = torch.from_numpy(blob_feature_map_tiled(n, seed, 0.12, 3.0)).cuda()
    lab = torch.zeros((n + 2,) * 3, dtype=torch.int32, device="cuda")
    base = None
    for cfg in cfgs:
        keys = []
        for kv in filter(None, cfg.split(",")):
            k, v = kv.split("="); os.environ[k] = v; keys.append(k)
        ws.get_context().reload_knobs()              # the library reads its knobs once per context
        ts = []
        for rep in range(3):
            lab.zero_(); torch.cuda.synchronize(); t = time.time()
            ws.segment_output_image(dfm, (0, 1, 2), 4, 3, out=lab); torch.cuda.synchronize(); ts.append(time.time() - t)
        cur = lab.clone()
        same = True if base is None else bool(torch.equal(cur, base))
        if base is None: base = cur
        print(f"seed {seed} [{cfg or 'default'}] {min(ts)*1e3:.1f} ms  (runs {[round(x*1e3,1) for x in ts]}) same-as-first {same}", flush=True)
        for k in keys: os.environ.pop(k, None)
        ws.get_context().reload_knobs()
