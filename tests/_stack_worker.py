"""Worker of tests/test_gpu_segmentation.py::test_segment_stack_with_the_hip_path: two gloo ranks on cuda:0 segment a
5-frame stack (frame t on rank t mod 2, uneven split) with the HIP path and all-gather the int32 label tiles (BASELINE
configs[2] at toy size); every rank must hold the labels a single process computes frame by frame."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from iterseg_amd import segmentation as seg, weights
    from iterseg_amd.unet import UNet
    from iterseg_amd.distributed import segment_stack
    from tests.synth import positive_volume
    dist.init_process_group("gloo")
    torch.cuda.set_device(0)
    chunk, margin = (4, 32, 32), (1, 8, 8)
    u = UNet(1, 5, device=0, batch=4)
    u.load_state_dict(weights.generate_state_dict(0))
    data = np.stack([positive_volume((8, 40, 48), 30 + t) for t in range(5)])

    def frame(vol):
        vol = (vol / vol.max()).astype(np.float32)
        cur = np.zeros(tuple(s + 2 for s in vol.shape), np.uint32)
        seg.affinity_watershed_for_chunks(vol, cur, chunk, margin, unet=u, output_volume=np.zeros((5,) + vol.shape, np.float32))
        return cur[1:-1, 1:-1, 1:-1].astype(np.int32)

    out = segment_stack(data, frame)
    ref = np.stack([frame(f) for f in data])
    ok = torch.tensor([1 if (np.array_equal(out.cpu().numpy(), ref) and ref.max() > 0) else 0])
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if int(ok) == 1 else 1)


if __name__ == "__main__":
    main()
