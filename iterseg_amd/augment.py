"""Mirror of ``iterseg.augment`` (reference: src/iterseg/augment.py:8-118,170-185) on the MI355X.

``augment_images(image, labels, ground_truth, augment_prob, rng)`` applies, with probability ``augment_prob``, an intensity
augmentation to the image (``augment_intensity``: ``image / image.max() * scale + shift`` with truncated-normal ``scale`` /
``shift``, then with probability ``noise_prob`` one of skimage's ``random_noise`` modes, else a clip to [0, 1]) and one
mirror / one (y, x) transposition to the image, every label set and the ground truth alike (``augment_order``).  Every
array goes through ONE device pass (``isx_augment_array``).  The decisions are drawn from ``rng`` (default: the
``numpy.random`` module, the global state the reference uses) in the reference's order; the noise fields are drawn from the
same ``rng`` (skimage's own generator is not reproducible from outside, upstream is unseeded).  Of skimage's modes
"gaussian" and "speckle" (variance 0.001 as upstream passes) run on the device; "localvar" is a Gaussian of variance 0.01
(skimage's default ``local_vars``) and runs the same way; "poisson" is drawn on the host (its level count needs
``np.unique`` of the image) and added as a precomputed field.
"""
from __future__ import annotations

import ctypes as C
from itertools import permutations

import numpy as np
import torch

from . import _lib
from .watershed import get_context


def continuous_choice(min_, max_, sigma, loc=0., size=1, rng=None):
    """augment.py:170-185: redraw N(loc, sigma) until it lies in [min_, max_]."""
    rng = np.random if rng is None else rng
    while True:
        out = rng.normal(loc=loc, scale=sigma, size=size)
        val = out if size == 1 else out.mean()
        if val >= min_ and val <= max_:
            return out


def _dev(a, dtype=torch.float32):
    if isinstance(a, torch.Tensor):
        return a.to(device="cuda", dtype=dtype).contiguous(), False
    return torch.from_numpy(np.ascontiguousarray(np.asarray(a)).astype(np.float32)).cuda(), True


def _pass(a, intensity=None, mirror_axis=0, transpose=0):
    """One device pass over a (Z, Y, X) or (C, Z, Y, X) array."""
    t, was_np = _dev(a)
    odtype = None if was_np else a.dtype
    shp = tuple(t.shape)
    s4 = shp if t.dim() == 4 else (1,) + shp
    oshape = s4[:2] + ((s4[3], s4[2]) if transpose else s4[2:])
    out = torch.empty(oshape, dtype=torch.float32, device=t.device)
    ctx = get_context(t.device.index)
    gain = shift = 0.0
    mode, noise = 0, None
    if intensity is not None:
        gain, shift, mode, noise = intensity
    _lib.check(ctx.lib.isx_augment_array(ctx.handle, C.c_void_p(t.data_ptr()), (C.c_int64 * 4)(*s4), int(intensity is not None),
                                         C.c_float(gain), C.c_float(shift), int(mode),
                                         C.c_void_p(noise.data_ptr()) if noise is not None else None, int(mirror_axis), int(transpose),
                                         C.c_void_p(out.data_ptr())))
    out = out if t.dim() == 4 else out[0]
    if was_np:
        r = out.cpu().numpy()
        return r.astype(np.asarray(a).dtype) if intensity is None else r
    return out.to(odtype) if intensity is None else out


def _intensity_plan(image, min_shift, max_shift, min_scale, max_scale, shift_sigma, scale_sigma, noise_prob, rng):
    t, _ = _dev(image)
    ctx = get_context(t.device.index)
    s_out, m_out = (C.c_double * 1)(), (C.c_float * 1)()
    v3 = t.view(-1, t.shape[-2], t.shape[-1])
    _lib.check(ctx.lib.isx_patch_stats(ctx.handle, C.c_void_p(v3.data_ptr()), _lib._dims(v3.shape), (C.c_int32 * 3)(0, 0, 0), 1,
                                       _lib._dims(v3.shape), s_out, m_out))
    mx = float(m_out[0])                                         # image.max(): one scalar back, like the reference's numpy
    scale = float(np.asarray(continuous_choice(min_scale, max_scale, scale_sigma, loc=1., rng=rng)).reshape(-1)[0])
    shift = float(np.asarray(continuous_choice(min_shift, max_shift, shift_sigma, rng=rng)).reshape(-1)[0])
    mode, noise = 0, None
    if rng.binomial(1, noise_prob):
        options = ['gaussian', 'localvar', 'poisson', 'speckle', 'gaussian', 'speckle']
        name = options[int(rng.randint(len(options)))]
        shape = tuple(t.shape)
        if name in ('gaussian', 'speckle'):
            noise = rng.normal(0.0, 0.001 ** 0.5, shape)         # var = 0.001 as upstream passes
            mode = 1 if name == 'gaussian' else 2
        elif name == 'localvar':
            noise = rng.normal(0.0, 0.01 ** 0.5, shape)          # skimage's default local_vars = 0.01 everywhere
            mode = 1
        else:                                                    # poisson: out = poisson(img * vals) / vals, as a field to add
            base = np.clip(t.cpu().numpy().astype(np.float64) / mx * scale + shift, 0.0, None)
            vals = 2 ** np.ceil(np.log2(max(len(np.unique(base)), 2)))
            noise = rng.poisson(base * vals) / float(vals) - base
            mode = 1
        noise = torch.from_numpy(np.ascontiguousarray(noise, dtype=np.float32)).cuda()
    return (scale / mx, shift, mode, noise)


def augment_intensity(image, min_shift=-0.1, max_shift=0.1, min_scale=0.8, max_scale=1.2, shift_sigma=0.02, scale_sigma=.05,
                      noise_prob=0.3, verbose=False, rng=None):
    rng = np.random if rng is None else rng
    plan = _intensity_plan(image, min_shift, max_shift, min_scale, max_scale, shift_sigma, scale_sigma, noise_prob, rng)
    return _pass(image, intensity=plan)


def _order_plan(ndim_min, mirror_prob, transpose_prob, used_axes, rng):
    """augment.py:84-104: (mirror axis 0/1/2, transpose flag) for used_axes = (-2, -1)."""
    if tuple(used_axes) != (-2, -1):
        raise NotImplementedError("augment_order: only the reference's default used_axes=(-2, -1) is built")
    mirror_axis = 0
    if rng.binomial(1, mirror_prob):
        mirror_axis = 1 + int(rng.randint(0, len(used_axes)))                # -2 (y) -> 1, -1 (x) -> 2
    transpose = 0
    if rng.binomial(1, transpose_prob):
        ps = [p for p in permutations(used_axes) if p != tuple(used_axes)]
        rng.randint(0, len(ps))                                              # one option: the swap
        transpose = 1
    return mirror_axis, transpose


def augment_order(images, mirror_prob=0.2, transpose_prob=0.2, used_axes=(-2, -1), verbose=False, rng=None):
    rng = np.random if rng is None else rng
    mirror_axis, transpose = _order_plan(3, mirror_prob, transpose_prob, used_axes, rng)
    return [_pass(img, None, mirror_axis, transpose) for img in images]


def augment_images(image, labels, ground_truth=None, augment_prob=0.9, rng=None):
    """augment.py:8-33, with the intensity and the order change of the image fused into one pass."""
    rng = np.random if rng is None else rng
    augment = rng.binomial(1, augment_prob)
    keys = list(labels.keys()) if isinstance(labels, dict) else None
    imgs = [labels[k] for k in keys] if keys is not None else [labels]
    if ground_truth is not None:
        imgs.append(ground_truth)
    if augment:
        plan = _intensity_plan(image, -0.1, 0.1, 0.8, 1.2, 0.02, .05, 0.3, rng)
        mirror_axis, transpose = _order_plan(3, 0.2, 0.2, (-2, -1), rng)
        image = _pass(image, plan, mirror_axis, transpose)
        imgs = [_pass(a, None, mirror_axis, transpose) for a in imgs]
    result = [image]
    nl = len(keys) if keys is not None else 1
    result.append({k: imgs[i] for i, k in enumerate(keys)} if keys is not None else imgs[0])
    if ground_truth is not None:
        result.append(imgs[nl])
    return tuple(result)
