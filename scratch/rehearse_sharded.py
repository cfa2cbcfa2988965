# 2-rank rehearsal of segment_volume_sharded on a one-GPU box (gloo backend, both ranks on cuda:0):
#   python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 scratch/rehearse_sharded.py
import os, sys, time; sys.path.insert(0, os.getcwd())
import numpy as np, torch, torch.distributed as dist
from iterseg_amd import weights, segmentation
from iterseg_amd.unet import UNet
from iterseg_amd.distributed import segment_volume_sharded
from tests.synth import positive_volume
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
torch.cuda.set_device(0)
shape, chunk, margin = (32, 256, 256), (10, 256, 256), (1, 64, 64)
u = UNet(1, 5, device=0, batch=4); u.load_state_dict(weights.generate_state_dict(0))
vol = torch.from_numpy(positive_volume(shape, 7)).cuda()
t0 = time.perf_counter()
lab = segment_volume_sharded(u, vol, chunk, margin)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
ref = segmentation.segment_volume_device(u, vol, chunk, margin)
same = bool(torch.equal(lab, ref))
print(f"rank {rank}/{world}: sharded == single-GPU: {same}; instances {int(ref.max())}; labelled {int((ref > 0).sum())}; {dt:.2f} s", flush=True)
ok = torch.tensor([1 if same else 0]); dist.all_reduce(ok, op=dist.ReduceOp.MIN)
dist.barrier(); dist.destroy_process_group()
sys.exit(0 if int(ok) == 1 else 1)
