#!/usr/bin/env python3
"""Golden-vector generator, stage 1d: ``UNet(chan_final_activations=[...])`` as ``train_unet`` builds it (reference:
train.py:134-141; unet.py:98-106 ``ConvModule.forward``, :205-209) -- forward output, loss, gradients and one optimiser
step for every final activation the reference knows ('relu', 'softmax', 'tanh', any other string = none), on the plain and on
the forked network, under the losses that accept such outputs.

Run ONLY in the build container (needs /root/reference, torch CPU):

    /usr/local/bin/python3 -W ignore tests/golden/gen_stage1d_final_activations.py

Same recipe as gen_stage1_unet.py / gen_stage1c_forked.py: the reference's ``unet.py`` and ``custom_loss.py`` are imported
unmodified behind the GUI/IO stubs, fed the build-owned seeded weights; only data is written next to this script.
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

WATCH = ["c0.conv0.weight", "c2.conv1.weight", "c4.batch1.weight", "up0.weight", "up3.bias", "c5_0.conv0.weight", "c7_0.batch0.bias",
         "c8_0.conv0.weight", "c8_0.conv1.weight", "c8_0.conv1.bias", "c8_0.batch1.weight", "c8_0.batch1.bias"]
WATCH_FORK = ["c8_1.conv0.weight", "c8_1.conv1.weight", "c8_1.batch1.weight", "c8_1.batch1.bias", "c6_1.conv1.weight"]

WEIGHTS_SEED = int(os.environ.get("WEIGHTS_SEED", 21))

# (tag, out_channels, chan_final_activations, loss)
CASES = [
    ("tanh_mse", 5, ["tanh"], "mse"),
    ("relu_mse", 5, ["relu"], "mse"),
    ("none_mse", 5, ["linear"], "mse"),              # not a name ConvModule.forward knows: the BatchNorm output is returned
    ("softmax_bce", 5, ["softmax"], "bce"),
    ("softmax_dice", 5, ["softmax"], "dice"),
    ("tanh_dice", 5, ["tanh"], "dice"),
    ("fork_softmax_tanh_mse", (3, 2), ["softmax", "tanh"], "mse"),
    ("fork_sigmoid_softmax_bce", (3, 2), ["sigmoid", "softmax"], "bce"),
]


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    _, unet_mod, closs = import_reference()
    rng = np.random.default_rng(21)
    x = positive_volume((4, 32, 32), seed=7)
    y = (rng.random((5, 4, 32, 32)) < 0.5).astype(np.float32)
    out = {"x": x, "y": y, "cases": np.asarray([c[0] for c in CASES])}
    out["weights_seed"] = np.asarray([WEIGHTS_SEED])
    for tag, oc, acts, loss_name in CASES:
        forked = isinstance(oc, tuple)
        sd_np = generate_state_dict(seed=WEIGHTS_SEED, out_channels=oc)
        u = unet_mod.UNet(in_channels=1, out_channels=oc, chan_final_activations=acts)
        u.load_state_dict(to_torch_state_dict(sd_np), strict=True)
        # a BatchNorm output within rounding distance of 0 makes the ReLU mask of that voxel -- and with it a visible share of
        # the gradients -- depend on the summation order of the convolution: the seeds are chosen so that none is that close
        near = []
        hooks = [m.register_forward_hook(lambda mod, i, o, near=near: near.append(float(o.detach().abs().min())))
                 for m in u.modules() if isinstance(m, torch.nn.BatchNorm3d)]
        opt = torch.optim.Adam(u.parameters(), lr=0.01)
        loss = {"mse": torch.nn.MSELoss(), "bce": torch.nn.BCELoss(), "dice": closs.DiceLoss()}[loss_name]
        opt.zero_grad()                                     # train.py:349-353
        y_hat = u(torch.from_numpy(x[None, None]).float())
        for h in hooks:
            h.remove()
        assert min(near) > 2e-6, (tag, min(near))
        l = loss(y_hat, torch.from_numpy(y[None]))
        l.backward()
        opt.step()
        opt.step()                                          # train.py:310 (the duplicated step)
        params = dict(u.named_parameters())
        out[f"{tag}.y_hat"] = y_hat.detach().numpy().copy()
        out[f"{tag}.loss"] = np.float64(l.item())
        for k in WATCH + (WATCH_FORK if forked else []):
            g = params[k].grad.detach().numpy()
            out[f"{tag}.grad." + k] = g.reshape(g.shape[0], -1)[:, :64].copy()
            v = params[k].detach().numpy()
            out[f"{tag}.param." + k] = v.reshape(v.shape[0], -1)[:, :64].copy()
        print(tag, "loss", l.item(), "y_hat range", float(y_hat.min()), float(y_hat.max()), "min |BN output|", min(near))
    # what nn.BCELoss does with an output that leaves [0, 1] (train.py:344-363 reaches it on the first batch)
    u = unet_mod.UNet(in_channels=1, out_channels=5, chan_final_activations=["tanh"])
    u.load_state_dict(to_torch_state_dict(generate_state_dict(seed=WEIGHTS_SEED, out_channels=5)), strict=True)
    try:
        torch.nn.BCELoss()(u(torch.from_numpy(x[None, None]).float()), torch.from_numpy(y[None]))
        out["tanh_bce.raised"] = np.asarray("")
    except Exception as e:                                  # noqa: BLE001
        out["tanh_bce.raised"] = np.asarray(type(e).__name__ + ": " + str(e))
    print("tanh + BCELoss:", out["tanh_bce.raised"])
    np.savez_compressed(os.path.join(HERE, "final_activations_4x32x32.npz"), **out)


if __name__ == "__main__":
    main()
