#!/usr/bin/env python3
"""Golden-vector generator, stage 1c: the forked network -- ``UNet(out_channels=(3, 2))`` as ``train_unet`` builds it for
``fork_channels`` (reference: train.py:139-141; unet.py:152-157,192-210,313-362).

Run ONLY in the build container (needs /root/reference, torch CPU):

    /usr/local/bin/python3 tests/golden/gen_stage1c_forked.py

Same recipe as gen_stage1_unet.py: the reference's ``unet.py`` is imported unmodified behind the GUI/IO stubs, fed the
build-owned seeded weights, and only data (inputs, outputs, gradients, updated parameters) is written next to this script.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from gen_stage1_unet import import_reference, positive_volume  # noqa: E402
from iterseg_amd.weights import generate_state_dict, to_torch_state_dict  # noqa: E402

FORKS = (3, 2)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    _, unet_mod, _ = import_reference()
    sd_np = generate_state_dict(seed=3, out_channels=FORKS)
    u = unet_mod.UNet(in_channels=1, out_channels=FORKS)
    ref_keys = list(u.state_dict().keys())
    assert ref_keys == list(sd_np.keys()), "schema of the forked network differs from iterseg_amd.weights.unet_schema"
    out = {"forks": np.asarray(FORKS), "keys": np.asarray(ref_keys)}
    rng = np.random.default_rng(11)
    x = positive_volume((4, 32, 32), seed=5)
    y = (rng.random((sum(FORKS), 4, 32, 32)) < 0.5).astype(np.float32)
    out["x"], out["y"] = x, y
    watch = ["c0.conv0.weight", "c2.conv1.weight", "c3.conv1.weight", "c4.conv1.weight", "c4.batch1.weight", "up0.weight", "up0.bias",
             "up1.weight", "up2.bias", "up3.weight", "c5_0.conv0.weight", "c5_1.conv0.weight", "c6_1.conv1.weight",
             "c7_0.batch0.bias", "c7_1.conv0.bias", "c8_0.conv0.weight", "c8_1.conv0.weight", "c8_0.conv1.weight",
             "c8_1.conv1.weight", "c8_1.conv1.bias", "c8_1.batch1.weight", "c8_0.batch1.bias"]
    for double_step in (False, True):
        u.load_state_dict(to_torch_state_dict(sd_np), strict=True)
        opt = torch.optim.Adam(u.parameters(), lr=0.01)
        loss = torch.nn.BCELoss()
        opt.zero_grad()                                     # train.py:349-353
        y_hat = u(torch.from_numpy(x[None, None]).float())  # forked_decoder: cat of the forks' outputs
        l = loss(y_hat, torch.from_numpy(y[None]))
        l.backward()
        opt.step()
        if double_step:                                     # train.py:310
            opt.step()
        params = dict(u.named_parameters())
        if not double_step:
            out["y_hat"] = y_hat.detach().numpy().copy()
            out["loss_bce"] = np.float64(l.item())
            out["loss_bce_per_channel"] = np.asarray([float(torch.nn.BCELoss()(y_hat[:, c], torch.from_numpy(y[None])[:, c]))
                                                      for c in range(sum(FORKS))])
            for k in watch:
                g = params[k].grad.detach().numpy()
                out["grad." + k] = g.reshape(g.shape[0], -1)[:, :64].copy()
        tag = "double" if double_step else "single"
        for k in watch:
            v = params[k].detach().numpy()
            out[f"param_{tag}." + k] = v.reshape(v.shape[0], -1)[:, :64].copy()
        post = u.state_dict()
        for k in ("c4.batch1.running_mean", "c8_1.batch1.running_var", "c8_1.batch1.num_batches_tracked"):
            out[f"post_{tag}." + k] = post[k].numpy().copy()
    np.savez_compressed(os.path.join(HERE, "forked_step_4x32x32.npz"), **out)
    print("forked: y_hat", out["y_hat"].shape, "loss", out["loss_bce"], "keys", len(ref_keys))


if __name__ == "__main__":
    main()
