"""Worker of tests/test_gpu_train.py::test_two_rank_data_parallel_step: run with torch.distributed.run, 2 ranks, gloo
backend, both ranks on cuda:0 (a one-GPU box; RCCL refuses two ranks per device).  Every rank takes one data-parallel
training step on its own patch (Trainer.train_step with world size 2: backward, ONE all-reduce of the flat gradient
buffer, isx_adam_step) and then rebuilds the same update alone: both patches' gradients on one replica, their mean,
the same Adam update.  Parameters must be equal bit for bit (a two-term sum is order-independent)."""
import ctypes as C
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from iterseg_amd import _lib, weights
    from iterseg_amd.train import Trainer
    from tests.synth import positive_volume
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    patch = (4, 32, 32)
    sd = weights.generate_state_dict(0)
    xs = [positive_volume(patch, 100 + r) for r in range(world)]
    ys = [(np.random.default_rng(200 + r).random((5,) + patch) < 0.4).astype(np.float32) for r in range(world)]
    ok = True
    # fp32 with and without the duplicated optimiser step, then BASELINE config 4 itself: bf16 mixed precision AND data parallel
    for double_step, precision in ((True, "fp32"), (False, "fp32"), (True, "bf16")):
        t = Trainer(sd, out_channels=5, patch=patch, precision=precision)
        loss, _ = t.train_step(xs[rank], ys[rank], lr=0.01, double_step=double_step)       # world size 2 branch
        ref = Trainer(sd, out_channels=5, patch=patch, precision=precision)
        g = torch.zeros_like(ref.grads)
        losses = []
        for r in range(world):
            l, _ = ref._call(xs[r], ys[r], "BCELoss", None, 0.01, 0, True)                 # backward only
            losses.append(l)
            g += ref.grads
        ref.grads.copy_(g / world)
        ref.unet._bind_stream()
        _lib.check(ref.unet.ctx.lib.isx_adam_step(ref.unet.ctx.handle, C.c_float(0.01), 2 if double_step else 1))
        torch.cuda.synchronize()
        ok = ok and abs(loss - losses[rank]) < 1e-7 and torch.equal(t.params, ref.params)
        ok = ok and not torch.equal(t.params, Trainer(sd, out_channels=5, patch=patch, precision=precision).params)       # it did move
        # replicas stay in lock step: every rank holds the same parameters after the step
        both = [torch.empty_like(t.params) for _ in range(world)]
        dist.all_gather(both, t.params)
        ok = ok and torch.equal(both[0], both[1])
    flag = torch.tensor([1 if ok else 0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if int(flag) == 1 else 1)


if __name__ == "__main__":
    main()
