"""CPU oracle for the affinity U-Net, chunked predict and one training step.

TEST INFRASTRUCTURE ONLY: imported by tests/, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py``; never by anything under ``iterseg_amd/``.

A *functional* fp32 restatement (torch CPU; the reference itself is torch.nn, so the
same ATen arithmetic is the natural oracle for this floating-point path) of

* ``UNet.forward``            -- reference src/iterseg/unet.py:284-364
* ``ConvModule.forward``      -- unet.py:91-106 (train-mode BatchNorm3d)
* ``make_chunks`` / ``process_chunks`` / ``predict_chunk_feature_map``
                              -- src/iterseg/predict.py:38-126
* ``_train_step`` + the second ``optimiser.step()`` -- src/iterseg/train.py:344-353, :310
* losses                      -- train.py:197-217, custom_loss.py:24-185

driven directly by a ``{key: ndarray}`` state dict (iterseg_amd.weights schema) instead
of nn.Module objects.  Pinned against tests/golden/{unet_taps,predict_*,train_step}*.npz
(captured from the reference's own modules) by tests/test_oracle_unet.py.
"""
from __future__ import annotations

import itertools
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5        # nn.BatchNorm3d default (unet.py:80-81)
BN_MOMENTUM = 0.1


# --------------------------------------------------------------------------- chunker
def make_chunks(arr_shape, chunk_shape, margin):
    """predict.py:38-61, restated with numpy like the reference."""
    ndim = len(arr_shape)
    if isinstance(margin, int):
        margin = [margin] * ndim
    starts, crops = [], []
    for arr, chk, mrg in zip(arr_shape, chunk_shape, margin):
        start = np.arange(0, arr - 2 * mrg, chk - 2 * mrg)
        start[-1] = arr - chk
        if len(start) > 1 and start[-1] == start[-2]:
            start = start[:-1]
        crop = np.array([(mrg, chk - mrg)] * len(start))
        crop[0, 0] = 0
        crop[-1, 0] = chk - (arr - np.sum(crop[:-1, 1] - crop[:-1, 0]))
        crop[-1, 1] = chk
        starts.append(start)
        crops.append(crop)
    return list(itertools.product(*starts)), list(itertools.product(*crops))


# --------------------------------------------------------------------------- forward
def _as_params(sd, requires_grad=False, dtype=None):
    p = OrderedDict()
    for k, v in sd.items():
        t = torch.from_numpy(np.ascontiguousarray(v)).clone() if isinstance(v, np.ndarray) else v
        if dtype is not None and t.dtype.is_floating_point:
            t = t.to(dtype)
        if requires_grad and t.dtype.is_floating_point and "running" not in k:
            t.requires_grad_(True)
        p[k] = t
    return p


# ---- bf16 mixed-precision emulation (BASELINE config 4's arithmetic; iterseg_amd/csrc/train.hip, conv_h2.hip PM = 2) ------
# The HIP training step in precision "bf16" rounds exactly these operands to bfloat16 (round to nearest even) and nothing
# else: in the FORWARD pass the input activations (after the producer's BatchNorm + ReLU, pooling, up-convolution and
# concatenation) and the weights of every wide convolution (Cin >= 32, Cout a multiple of 32: `use_mfma`), and in the DATA
# gradient of those layers (except c0.conv1... see _bf16_dgrad) the incoming gradient d(raw) and the weights.  Products are
# exact, accumulation, bias, BatchNorm (statistics and backward), weight gradients (from the UNROUNDED activations and
# gradients), the thin layers, the loss and Adam stay in the working precision.  With `bf16_emulation(True)` the convolutions
# below do the same, so the HIP step can be compared with this oracle at the fp32 tolerance instead of with its own fp32 run.
_BF16 = {"on": False}


class bf16_emulation:
    def __init__(self, on=True):
        self.on = on

    def __enter__(self):
        self.prev = _BF16["on"]
        _BF16["on"] = self.on

    def __exit__(self, *a):
        _BF16["on"] = self.prev


def _rb(t):
    return t.to(torch.float32).to(torch.bfloat16).to(t.dtype)     # the kernels round fp32 values


def _bf16_forward(cin, cout):
    return cin >= 32 and cin % 4 == 0 and cout % 32 == 0            # use_mfma (unet.hip)


def _bf16_dgrad(name, cin, cout):
    # train.hip isx_train_bind: a bf16 dgrad operator exists for wide layers with Cin % 32 == 0 and Cout % 8 == 0,
    # never for layer 0 (c0.conv0 has no data gradient at all)
    return _bf16_forward(cin, cout) and cin % 32 == 0 and cout % 8 == 0


def _bf16_wgrad(cin, cout):
    # train.hip wgrad(): the bf16 weight-gradient kernel serves every layer with Cin % 32 == 0
    return cin % 32 == 0


class _ConvMixed(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, rf, rd):
        ctx.save_for_backward(x, w)
        ctx.rd = rd
        ctx.rf = rf
        return F.conv3d(_rb(x) if rf else x, _rb(w) if rf else w, b, padding=1)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        gx = None
        if ctx.needs_input_grad[0]:
            gx = torch.nn.grad.conv3d_input(x.shape, _rb(w) if ctx.rd else w, _rb(g) if ctx.rd else g, padding=1)
        # the weight gradient of a wide layer rounds its two operands too (train.hip wgrad_bf16_kernel: the conv input after
        # BatchNorm + ReLU and d(raw), both to bfloat16; fp32 accumulation); _BF16["wgrad"] = False restates round 2's step
        rw = _bf16_wgrad(int(w.shape[1]), int(w.shape[0])) and _BF16.get("wgrad", True)
        gw = torch.nn.grad.conv3d_weight(_rb(x) if rw else x, w.shape, _rb(g) if rw else g, padding=1)
        return gx, gw, g.sum((0, 2, 3, 4)), None, None


def _conv(p, key, x):
    w, b = p[f"{key}.weight"], p[f"{key}.bias"]
    if not _BF16["on"]:
        return F.conv3d(x, w, b, padding=1)
    cout, cin = int(w.shape[0]), int(w.shape[1])
    return _ConvMixed.apply(x, w, b, _bf16_forward(cin, cout), _bf16_dgrad(key, cin, cout))


def conv_module(p, name, x, final="relu", taps=None):
    for j in (0, 1):
        x = _conv(p, f"{name}.conv{j}", x)
        if taps is not None:
            taps[f"{name}.conv{j}"] = x
        # train-mode BN: batch statistics, biased variance; running stats updated in place
        x = F.batch_norm(x, p[f"{name}.batch{j}.running_mean"], p[f"{name}.batch{j}.running_var"],
                         p[f"{name}.batch{j}.weight"], p[f"{name}.batch{j}.bias"],
                         training=True, momentum=BN_MOMENTUM, eps=BN_EPS)
        if f"{name}.batch{j}.num_batches_tracked" in p:
            p[f"{name}.batch{j}.num_batches_tracked"] += 1
        if taps is not None:
            taps[f"{name}.batch{j}"] = x
        if j == 0 or final == "relu":
            x = F.relu(x)
        elif final == "sigmoid":
            x = torch.sigmoid(x)
        elif final == "tanh":
            x = torch.tanh(x)
        elif final == "softmax":
            x = F.softmax(x, dim=1)                  # nn.Softmax() with dim=None on a 5-D tensor (unet.py:87,100-101)
        # any other name: the BatchNorm output (unet.py:98-106 has no else branch)
    if taps is not None:
        taps[name] = x
    return x


def unet_forward(p, x, taps=None, final="sigmoid"):
    """x: (1,1,D,H,W) float32 tensor -> (1,C,D,H,W).  ``final``: one name, or UNet(chan_final_activations=[...])'s list (one per
    decoder, unet.py:205-209).  One decoder, or -- when the parameters hold ``c5_1...`` -- the forked
    network (unet.py:313-325): every fork runs its own c5_i..c8_i on the shared encoder outputs and transposed convolutions,
    the outputs are concatenated in fork order."""
    c0 = conv_module(p, "c0", x, taps=taps)
    x = F.max_pool3d(c0, (1, 2, 2), (1, 2, 2), padding=(0, 1, 1))
    if taps is not None:
        taps["d0"] = x
    c1 = conv_module(p, "c1", x, taps=taps)
    x = F.max_pool3d(c1, (1, 2, 2), (1, 2, 2), padding=(0, 1, 1))
    c2 = conv_module(p, "c2", x, taps=taps)
    x = F.max_pool3d(c2, (1, 2, 2), (1, 2, 2), padding=(0, 1, 1))
    c3 = conv_module(p, "c3", x, taps=taps)
    x = F.max_pool3d(c3, (2, 2, 2), (2, 2, 2), padding=(0, 1, 1))
    if taps is not None:
        taps["d3"] = x
    c4 = conv_module(p, "c4", x, taps=taps)
    outs = []
    i = 0
    while f"c5_{i}.conv0.weight" in p:
        ftaps = taps if i == 0 else None                       # the tap names of the fixtures are those of fork 0
        x = F.conv_transpose3d(c4, p["up0.weight"], p["up0.bias"], stride=(2, 2, 2), groups=256)
        if ftaps is not None:
            ftaps["up0"] = x
        x = torch.cat([x[:, :, :, :-1, :-1], c3], 1)
        if ftaps is not None:
            ftaps["c5_0.in"] = x
        x = conv_module(p, f"c5_{i}", x, taps=ftaps)
        x = F.conv_transpose3d(x, p["up1.weight"], p["up1.bias"], stride=(1, 2, 2), groups=128)
        x = torch.cat([x[:, :, :, :-1, :-1], c2], 1)
        x = conv_module(p, f"c6_{i}", x, taps=ftaps)
        x = F.conv_transpose3d(x, p["up2.weight"], p["up2.bias"], stride=(1, 2, 2), groups=64)
        x = torch.cat([x[:, :, :, :-1, :-1], c1], 1)
        x = conv_module(p, f"c7_{i}", x, taps=ftaps)
        x = F.conv_transpose3d(x, p["up3.weight"], p["up3.bias"], stride=(1, 2, 2), groups=32)
        if ftaps is not None:
            ftaps["up3"] = x
        x = torch.cat([x[:, :, :, 1:-1, 1:-1], c0], 1)
        if ftaps is not None:
            ftaps["c8_0.in"] = x
        outs.append(conv_module(p, f"c8_{i}", x, final=final if isinstance(final, str) else final[i], taps=ftaps))
        i += 1
    return outs[0] if len(outs) == 1 else torch.cat(outs, 1)


def predict_volume(sd, volume, chunk_size, margin, threads=None):
    """process_chunks + predict_chunk_feature_map (predict.py:64-126): (Z,Y,X) -> (5,Z,Y,X)."""
    if threads:
        torch.set_num_threads(threads)
    p = _as_params(sd)
    out = np.zeros((5,) + tuple(volume.shape), np.float32)
    starts, crops = make_chunks(volume.shape, chunk_size, margin)
    with torch.no_grad():
        for st, cr in zip(starts, crops):
            sl = tuple(slice(int(s), int(s) + int(c)) for s, c in zip(st, chunk_size))
            y = unet_forward(p, torch.from_numpy(np.ascontiguousarray(volume[sl]))[None, None]).numpy()
            crs = tuple(slice(int(a), int(b)) for a, b in cr)
            out[(slice(None),) + sl][(slice(None),) + crs] = y[0][(slice(None),) + crs]
    return out


# --------------------------------------------------------------------------- losses / training
def flatten_channels(x, channel_dim=1):
    order = [channel_dim] + [i for i in range(x.dim()) if i != channel_dim]
    return torch.flatten(x.permute(*order), start_dim=1)


def loss_fn(name, y_hat, y, chan_weights=None):
    if name == "BCELoss":
        return F.binary_cross_entropy(y_hat, y)
    if name == "MSELoss":
        return F.mse_loss(y_hat, y)
    if name == "DiceLoss":                                   # custom_loss.py:24-53, smooth=1
        a, b = flatten_channels(y_hat), flatten_channels(y)
        inter = (a * b).sum(-1)
        dice = (2. * inter + 1) / (a.sum(-1) + b.sum(-1) + 1)
        return (1 - dice).mean()
    if name == "WeightedBCE":                                # custom_loss.py:161-185
        a, b = flatten_channels(y_hat), flatten_channels(y)
        per = F.binary_cross_entropy(a, b, reduction="none").mean(-1)
        return (per * torch.as_tensor(chan_weights, dtype=per.dtype)).mean()
    if name == "Channelwise":                                # custom_loss.py:131-159, reduction='mean'
        kinds, sels = chan_weights                           # e.g. (["BCELoss", "MSELoss"], [slice(0, 4), 4])
        vals = [loss_fn(k, y_hat[:, s], y[:, s]) for k, s in zip(kinds, sels)]
        return torch.stack(vals).mean()
    raise ValueError(name)


def channel_log(name, y_hat, y, chan_weights=None):
    """What channel_losses_to_dict (custom_loss.py:8-17) logs: the loss applied to y_hat[:, i] -- a
    (N, Z, Y, X) tensor, so for DiceLoss the z planes take the place of the channels."""
    return [float(loss_fn(name, y_hat[:, i].detach(), y[:, i].detach(), chan_weights)) for i in range(y_hat.shape[1])]


def train_step(sd, x, y, lr=0.01, loss="BCELoss", double_step=True, chan_weights=None, dtype=torch.float32, bf16=False):
    """One ``_train_step`` (+ the duplicated ``optimiser.step()`` of train.py:310).  ``bf16``: the mixed-precision
    arithmetic of the HIP step in precision "bf16" (see ``bf16_emulation``).

    Returns ``(loss, y_hat, grads, new_params)`` with numpy payloads.
    """
    p = _as_params(sd, requires_grad=True, dtype=dtype)
    leaves = [t for t in p.values() if t.requires_grad]
    opt = torch.optim.Adam(leaves, lr=lr)
    opt.zero_grad()
    with bf16_emulation(bool(bf16)):
        y_hat = unet_forward(p, torch.from_numpy(x)[None, None].to(dtype))
        l = loss_fn(loss, y_hat, torch.from_numpy(y)[None].to(dtype), chan_weights)
        l.backward()
    grads = {k: t.grad.detach().numpy().copy() for k, t in p.items() if t.requires_grad}
    opt.step()
    if double_step:
        opt.step()
    new = OrderedDict((k, t.detach().numpy().copy()) for k, t in p.items())
    return float(l.item()), y_hat.detach().numpy(), grads, new
