"""Mirror of the training-data producer of ``iterseg.train_io`` (reference: src/iterseg/train_io.py) on the MI355X.

* :func:`get_image_chunks` -- train_io.py:224-275: ``n`` random ``shape`` crops of an image that pass the brightness test
  ``x.mean() / x.max() > min_brightness_prop``.  The image is uploaded once; a candidate costs one device reduction
  (``isx_patch_stats``), an accepted crop one gather (``isx_patch_gather``).  The draws come from ``rng`` (anything with
  ``randint(low, high)``; default: the ``numpy.random`` module, i.e. the global state the reference draws from) in the
  reference's order -- one ``randint(0, im.shape[j] - dim - 1)`` per axis and candidate -- so the same draws pick the same
  crops.
* :func:`get_labels_chunks` -- train_io.py:287-320: the training-label channels of the whole ground truth
  (``iterseg_amd.labels.get_training_labels``: on the device), cropped with the image's slices.
* :func:`augment_chunks` -- train_io.py:329-341 over :mod:`iterseg_amd.augment`.

``device=True`` keeps the chunks on the GPU (CUDA tensors, what ``train_unet`` consumes); the default returns numpy arrays
like the reference.
"""
from __future__ import annotations

import ctypes as C
from datetime import datetime

import numpy as np
import torch

from . import _lib
from .labels import get_training_labels
from .watershed import get_context


def _resident(image, dtype):
    if isinstance(image, torch.Tensor):
        return image.to(device="cuda", dtype=dtype).contiguous()
    return torch.from_numpy(np.ascontiguousarray(np.asarray(image)).astype(np.float32 if dtype == torch.float32 else np.int32)).cuda()


def _starts(slices):
    return np.ascontiguousarray([[s.start for s in sl] for sl in slices], dtype=np.int32).reshape(-1, 3)


def _gather(vol, slices, shape):
    """``[vol[s] for s in slices]`` as one device gather: (n,) + shape, same dtype (4-byte elements)."""
    ctx = get_context(vol.device.index)
    st = _starts(slices)
    out = torch.empty((len(slices),) + tuple(shape), dtype=vol.dtype, device=vol.device)
    _lib.check(ctx.lib.isx_patch_gather(ctx.handle, C.c_void_p(vol.data_ptr()), _lib._dims(vol.shape),
                                        st.ctypes.data_as(C.POINTER(C.c_int32)), len(slices), _lib._dims(shape),
                                        C.c_void_p(out.data_ptr())))
    return out


def get_image_chunks(image, shape=(10, 256, 256), n=25, min_brightness_prop=0.3, image_no=0, rng=None, device=False):
    rng = np.random if rng is None else rng
    was_np = not isinstance(image, torch.Tensor)
    im = _resident(image, torch.float32)
    assert im.dim() == len(shape) == 3
    ctx = get_context(im.device.index)
    dims = _lib._dims(im.shape)
    cshape = _lib._dims(shape)
    nvox = float(np.prod(shape))
    slices, starts = [], []
    df = {'z_start': [], 'y_start': [], 'x_start': []}
    ids = []
    i = 0
    s_out, m_out = (C.c_double * 1)(), (C.c_float * 1)()
    while i < n:
        # train_io.py:240-245 -- the upper bound excludes the last two positions, as upstream
        ri = [int(rng.randint(0, im.shape[j] - dim - 1)) for j, dim in enumerate(shape)]
        st = (C.c_int32 * 3)(*ri)
        _lib.check(ctx.lib.isx_patch_stats(ctx.handle, C.c_void_p(im.data_ptr()), dims, st, 1, cshape, s_out, m_out, None))
        mx = float(m_out[0])
        mbrightprop = (s_out[0] / nvox) / mx if mx != 0.0 else float('nan')      # x.mean() / x.max(): nan never passes
        if abs(mbrightprop - min_brightness_prop) <= 1e-5 * abs(min_brightness_prop):
            # within rounding of the threshold: upstream's quotient is numpy's pairwise float32 mean over the crop -- decide
            # on exactly that arithmetic (one crop back to the host, once in ~10^5 candidates)
            x = _gather(im, [tuple(slice(r, r + d) for r, d in zip(ri, shape))], shape)[0].cpu().numpy()
            mbrightprop = x.mean() / x.max()
        if mbrightprop > min_brightness_prop:
            slices.append(tuple(slice(r, r + d) for r, d in zip(ri, shape)))
            starts.append(ri)
            for key, r in zip(('z_start', 'y_start', 'x_start'), ri):
                df[key].append(r)
            ids.append(datetime.now().strftime("%y%m%d_%H%M%S") + f'_img-{image_no}_chunk-{i}')
            i += 1
    xs = _gather(im, slices, shape)
    if was_np and not device:
        src = np.asarray(image)
        xs = [src[s] for s in slices]            # the reference returns views of the caller's array, in its dtype
    else:
        xs = list(xs.unbind(0)) if device else [x.cpu() for x in xs.unbind(0)]
    import pandas as pd
    df['data_ids'] = ids
    return {'x': xs, 'slices': slices, 'ids': ids, 'df': pd.DataFrame(df), 'n': len(xs)}


def get_labels_chunks(chunk_dict, ground_truth, channels=('z-1', 'y-1', 'x-1', 'centreness-log'), scale=(4, 1, 1), device=False):
    if not isinstance(channels, dict):
        channels = {'y': channels}
    chunk_dict['channels'] = channels
    gt = _resident(ground_truth, torch.int32)
    slices = chunk_dict['slices']
    shape = tuple(s.stop - s.start for s in slices[0]) if slices else ()
    chunk_dict['ys'] = {}
    for key, chans in channels.items():
        labs = get_training_labels(gt, chans, scale).contiguous()         # (C, Z, Y, X) on the device
        nc, Z = int(labs.shape[0]), int(labs.shape[1])
        # all channels of all crops in one gather: the channels stacked along z, crop (i, c) starts at z = c * Z + z0_i
        sl = [(slice(c * Z + s[0].start, c * Z + s[0].stop), s[1], s[2]) for s in slices for c in range(nc)]
        ys = _gather(labs.view(nc * Z, labs.shape[2], labs.shape[3]), sl, shape).view((len(slices), nc) + tuple(shape))
        chunk_dict['ys'][key] = list(ys.unbind(0)) if device else [y.cpu().numpy() for y in ys.unbind(0)]
    g = _gather(gt, slices, shape)
    chunk_dict['ground_truth'] = list(g.unbind(0)) if device else [a.cpu().numpy() for a in g.unbind(0)]
    return chunk_dict


def _read_chunk_dict(chunk_dict):
    x, ys, gt, n = chunk_dict['x'], chunk_dict['ys'], chunk_dict['ground_truth'], chunk_dict['n']
    assert n == len(x) == len(gt)
    for key in ys:
        assert len(ys[key]) == n
    return x, ys, list(ys.keys()), gt, n


def augment_chunks(chunk_dict, rng=None):
    from .augment import augment_images
    x, ys, labs_keys, gt, n = _read_chunk_dict(chunk_dict)
    for i in range(n):
        labels_dict = {key: ys[key][i] for key in labs_keys}
        image, labels_dict, ground_truth = augment_images(x[i], labels_dict, gt[i], rng=rng)
        chunk_dict['x'][i] = image
        for key in labs_keys:
            chunk_dict['ys'][key][i] = labels_dict[key]
        chunk_dict['ground_truth'][i] = ground_truth
    return chunk_dict
