"""State-dict schema of the affinity U-Net and a build-owned seeded weight generator.

The on-disk weight format of the reference is ``torch.save(unet.state_dict())``
(reference: src/iterseg/train.py:414-422, loaded strictly at predict.py:34).  This
module restates the *schema* (key names, order, shapes) of
``iterseg.unet.UNet(in_channels=1, out_channels=5)`` (reference: unet.py:126-249) so
that weights can be produced and validated without importing the reference, and
provides the deterministic generator used by the golden fixtures, the tests and
``bench.py`` (the shipped default weights are absent upstream).

Nothing in here touches the GPU.
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np

# (module name, Cin, Cout) in the order the reference registers them
# (unet.py:192-210: c0..c4, then for fork i: c5_i, c6_i, c7_i, c8_i).
_ENCODER = (("c0", None, 32), ("c1", 32, 64), ("c2", 64, 128), ("c3", 128, 256),
            ("c4", 256, 256))
_DECODER = ((5, 512, 128), (6, 256, 64), (7, 128, 32))
# (name, channels, kernel) for the depthwise transposed convolutions (unet.py:215-242)
_UPS = (("up0", 256, (2, 2, 2)), ("up1", 128, (1, 2, 2)), ("up2", 64, (1, 2, 2)),
        ("up3", 32, (1, 2, 2)))


def _conv_module_entries(name, cin, cout):
    ent = []
    for j, ci in ((0, cin), (1, cout)):
        ent.append((f"{name}.conv{j}.weight", (cout, ci, 3, 3, 3)))
        ent.append((f"{name}.conv{j}.bias", (cout,)))
    for j in (0, 1):
        ent.append((f"{name}.batch{j}.weight", (cout,)))
        ent.append((f"{name}.batch{j}.bias", (cout,)))
        ent.append((f"{name}.batch{j}.running_mean", (cout,)))
        ent.append((f"{name}.batch{j}.running_var", (cout,)))
        ent.append((f"{name}.batch{j}.num_batches_tracked", ()))
    return ent


def unet_schema(in_channels=1, out_channels=5):
    """Ordered ``{key: shape}`` of ``UNet(in_channels, out_channels).state_dict()``.

    ``out_channels`` may be an int or a tuple (forked decoder, unet.py:152-157).
    """
    forks = out_channels if isinstance(out_channels, tuple) else (out_channels,)
    ent = []
    for name, cin, cout in _ENCODER:
        ent += _conv_module_entries(name, in_channels if cin is None else cin, cout)
    for i, c in enumerate(forks):
        for n, cin, cout in _DECODER:
            ent += _conv_module_entries(f"c{n}_{i}", cin, cout)
        ent += _conv_module_entries(f"c8_{i}", 64, c)
    for name, ch, k in _UPS:
        ent.append((f"{name}.weight", (ch, 1) + k))
        ent.append((f"{name}.bias", (ch,)))
    return OrderedDict(ent)


def generate_state_dict(seed=0, in_channels=1, out_channels=5, bn_jitter=0.1):
    """Deterministic weights as ``{key: np.ndarray}`` in schema order.

    Every key has its own PCG64 stream ``[seed, key_index]`` so the result does not
    depend on generation order.  Convolution weights/biases are uniform in
    ``+-1/sqrt(fan_in)``; BatchNorm gamma is ``1 +- bn_jitter`` and beta ``+-bn_jitter``
    so the affine part is exercised; running stats are (0, 1, 0).
    """
    sd = OrderedDict()
    for idx, (key, shape) in enumerate(unet_schema(in_channels, out_channels).items()):
        rng = np.random.Generator(np.random.PCG64([int(seed), idx]))
        leaf = key.rsplit(".", 1)[1]
        if leaf == "num_batches_tracked":
            arr = np.zeros((), np.int64)
        elif leaf == "running_mean":
            arr = np.zeros(shape, np.float32)
        elif leaf == "running_var":
            arr = np.ones(shape, np.float32)
        elif ".batch" in key:
            u = rng.random(shape, dtype=np.float32) * 2 - 1
            arr = (1.0 + bn_jitter * u if leaf == "weight" else bn_jitter * u)
            arr = arr.astype(np.float32)
        else:
            wkey = key.rsplit(".", 1)[0] + ".weight"
            wshape = unet_schema(in_channels, out_channels)[wkey]
            fan_in = int(np.prod(wshape[1:]))
            bound = np.float32(1.0 / np.sqrt(fan_in))
            arr = ((rng.random(shape, dtype=np.float32) * 2 - 1) * bound).astype(np.float32)
        sd[key] = arr
    return sd


def to_torch_state_dict(sd):
    """numpy state dict -> torch tensors (for ``load_state_dict`` / ``torch.save``)."""
    import torch
    # np.ascontiguousarray promotes 0-d arrays (num_batches_tracked) to shape (1,): keep their shape
    return OrderedDict((k, torch.from_numpy(np.ascontiguousarray(v)).clone().reshape(np.shape(v))) for k, v in sd.items())


def from_torch_state_dict(tsd):
    """torch state dict -> contiguous numpy arrays (CPU)."""
    return OrderedDict((k, v.detach().cpu().numpy().copy()) for k, v in tsd.items())


def check_state_dict(sd, in_channels=1, out_channels=5):
    """Strict schema check mirroring ``load_state_dict(strict=True)`` (predict.py:34)."""
    schema = unet_schema(in_channels, out_channels)
    missing = [k for k in schema if k not in sd]
    extra = [k for k in sd if k not in schema]
    if missing or extra:
        raise RuntimeError(f"state_dict mismatch: missing {missing[:4]}..., unexpected {extra[:4]}...")
    for k, shape in schema.items():
        if tuple(sd[k].shape) != tuple(shape):
            raise RuntimeError(f"size mismatch for {k}: {tuple(sd[k].shape)} vs {tuple(shape)}")
    return True
