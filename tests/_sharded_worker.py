"""Worker of tests/test_gpu_segmentation.py::test_two_rank_sharded_volume: run with torch.distributed.run, 2 ranks,
gloo backend, both ranks on cuda:0 (a one-GPU box; RCCL refuses two ranks per device)."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from iterseg_amd import weights, segmentation
    from iterseg_amd.unet import UNet
    from iterseg_amd.distributed import segment_volume_sharded
    from tests.synth import positive_volume
    dist.init_process_group("gloo")
    torch.cuda.set_device(0)
    chunk, margin = (4, 32, 32), (1, 8, 8)
    u = UNet(1, 5, device=0, batch=4)
    u.load_state_dict(weights.generate_state_dict(0))
    vol = torch.from_numpy(positive_volume((12, 64, 64), 7)).cuda()
    lab = segment_volume_sharded(u, vol, chunk, margin)
    ref = segmentation.segment_volume_device(u, vol, chunk, margin)
    ok = torch.tensor([1 if (torch.equal(lab, ref) and int(ref.max()) > 0) else 0])
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if int(ok) == 1 else 1)


if __name__ == "__main__":
    main()
