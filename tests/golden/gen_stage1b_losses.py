#!/usr/bin/env python3
"""Golden-vector generator, stage 1b: loss classes of the reference on a stored (y_hat, y) pair.

Run ONLY in the build container (needs /root/reference, torch CPU):

    /usr/local/bin/python3 tests/golden/gen_stage1b_losses.py

Loads ``train_step_4x32x32.npz`` (written by gen_stage1_unet.py), imports the reference's
``custom_loss.py`` unmodified behind the stage-1 stubs and records, for DiceLoss and ChannelwiseLoss
(custom_loss.py:24-53, :131-159): the loss value, its gradient with respect to y_hat (torch autograd through
the reference's own forward), and what ``channel_losses_to_dict`` (custom_loss.py:8-17) logs per channel.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen_stage1_unet import import_reference  # noqa: E402


def main():
    _, _, closs = import_reference()
    d = np.load(os.path.join(HERE, "train_step_4x32x32.npz"))
    y = torch.from_numpy(d["y"][None])
    out = {}
    cases = {
        "dice": closs.DiceLoss(),
        "channelwise": closs.ChannelwiseLoss([torch.nn.BCELoss(), torch.nn.MSELoss()], [slice(0, 4), 4], "cpu"),
        "channelwise3": closs.ChannelwiseLoss([torch.nn.MSELoss(), torch.nn.BCELoss(), torch.nn.BCELoss()],
                                              [slice(0, 3), 3, 4], "cpu"),
    }
    for name, fn in cases.items():
        yh = torch.from_numpy(d["y_hat"]).clone().requires_grad_(True)
        l = fn(yh, y)
        l.backward()
        out[f"{name}.loss"] = np.float64(l.item())
        out[f"{name}.grad"] = yh.grad.numpy().copy()
    chan = {k: [] for k in "abcde"}
    closs.channel_losses_to_dict(torch.from_numpy(d["y_hat"]), y, tuple("abcde"), cases["dice"], chan)
    out["dice.per_channel_log"] = np.array([chan[k][0] for k in "abcde"])
    np.savez_compressed(os.path.join(HERE, "losses_4x32x32.npz"), **out)
    print({k: (v if np.ndim(v) == 0 else np.shape(v)) for k, v in out.items()})


if __name__ == "__main__":
    main()
