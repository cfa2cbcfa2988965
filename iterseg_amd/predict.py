"""Mirror of ``iterseg.predict`` (reference: src/iterseg/predict.py) on the MI355X path.

* :func:`make_chunks`                -- predict.py:38-61 (host integer logic, chunking.py)
* :func:`load_unet`                  -- predict.py:25-35
* :func:`process_chunks`             -- predict.py:64-96
* :func:`predict_chunk_feature_map`  -- predict.py:100-126

``process_chunks`` keeps the reference's generic form (any ``process_data_function``); when it is
handed :func:`predict_chunk_feature_map` and an accelerated :class:`~iterseg_amd.unet.UNet` the
whole chunk loop runs on the device in launch sets (``isx_predict_volume``) and the stitched
feature map is copied into the caller's ``output_volume`` once, instead of one host round trip
per chunk.
"""
from __future__ import annotations

import numpy as np
import torch

from . import unet as unet_mod
from .chunking import make_chunks  # noqa: F401  (re-exported: same name as the reference)

IGNORE_CUDA = False          # kept for API compatibility; this build has no CPU path
DEFAULT_UNET_PATH = None     # upstream's default weights are not distributed (.MISSING_LARGE_BLOBS)


def load_unet(u_state_fn=DEFAULT_UNET_PATH, device=None, batch=32, precision=None):
    """``UNet(1, 5)`` with the state dict stored at ``u_state_fn`` (``torch.save`` format)."""
    if u_state_fn is None:
        raise FileNotFoundError(
            "no U-Net weights given: iterseg's default 232208_161159_plateseg.pt is not distributed; "
            "pass a .pt/.pth state dict (see iterseg_amd.weights.generate_state_dict for synthetic ones)")
    u = unet_mod.UNet(in_channels=1, out_channels=5, device=device, batch=batch, precision=precision)
    u.load_state_dict(torch.load(u_state_fn, map_location="cpu"))
    return u


def predict_chunk_feature_map(input_volume, sl, unet=False, default_only_mask=False, **kwargs):
    """One chunk through the U-Net; returns the (1, 5, cz, cy, cx) numpy array (predict.py:100-126)."""
    assert unet != False, 'Please ensure a unet is loaded and supplied'   # noqa: E712 (reference wording)
    sl = sl[1:]
    tensor = torch.from_numpy(np.ascontiguousarray(input_volume[sl])[np.newaxis, np.newaxis])
    predicted_array = unet(tensor.cuda()).detach().cpu().numpy()
    if default_only_mask:
        predicted_array = predicted_array[3, ...]
    return predicted_array


def process_chunks(input_volume, chunk_size, output_volume, margin, process_data_function, config=None):
    if config is None:
        config = {}
    ndim = len(chunk_size)
    u = config.get('unet')
    if (process_data_function is predict_chunk_feature_map and isinstance(u, unet_mod.UNet) and ndim == 3
            and not config.get('default_only_mask', False)):
        # device-resident chunk loop
        if isinstance(input_volume, torch.Tensor) and input_volume.is_cuda:
            vol = input_volume.float().contiguous()
        else:
            vol = torch.from_numpy(np.ascontiguousarray(input_volume, dtype=np.float32)).to(f"cuda:{u.device}")
        if isinstance(output_volume, torch.Tensor) and output_volume.is_cuda:
            u.predict_volume(vol, chunk_size, margin, feature_map=output_volume)
        else:
            fm = u.predict_volume(vol, chunk_size, margin)
            output_volume[...] = fm.cpu().numpy()
        return output_volume
    # generic path, same control flow as the reference (predict.py:77-96)
    chunk_starts, chunk_crops = make_chunks(input_volume.shape[-ndim:], chunk_size, margin=margin)
    for start, crop in zip(chunk_starts, chunk_crops):
        sl = tuple(slice(start0, start0 + step) for start0, step in zip(start, chunk_size))
        sl = (slice(None),) + sl
        predicted_array = process_data_function(input_volume, sl, **config)
        p_dim = predicted_array.ndim
        o_dim = output_volume.ndim
        cr = (slice(None),) * (p_dim - o_dim) + tuple(slice(i, j) for i, j in crop)
        pred_c = (0,) + cr
        output_volume[sl][cr] = predicted_array[pred_c]
    return output_volume


def get_device():
    return torch.device("cuda:0")
