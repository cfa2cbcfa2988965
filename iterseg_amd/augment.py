"""Mirror of ``iterseg.augment`` (reference: src/iterseg/augment.py:8-118,170-185) on the MI355X.

``augment_images(image, labels, ground_truth, augment_prob, rng)`` applies, with probability ``augment_prob``, an intensity
augmentation to the image (``augment_intensity``: ``image / image.max() * scale + shift`` with truncated-normal ``scale`` /
``shift``, then with probability ``noise_prob`` one of skimage's ``random_noise`` modes, else a clip to [0, 1]) and one
mirror / one axis permutation to the image, every label set and the ground truth alike (``augment_order``, any
``used_axes``).  Every array goes through ONE device pass (``isx_augment_array``); arrays that only change order move as raw
32-bit words, so integer labels keep every bit.

The decisions AND the noise fields are drawn from ``rng`` (default: the ``numpy.random`` module -- the global state both the
reference and scikit-image 0.18.3's ``random_noise`` draw from) in upstream's order and with upstream's calls, so
``np.random.seed(k)`` gives the reference's augmentation: pinned by tests/golden/augment.npz (generated from the reference,
tests/golden/gen_stage5_train_io.py) -- permutations exact, intensities within 2e-6 (float32 here, float64 upstream).
skimage's clip range is honoured: an image with a negative value after ``* scale + shift`` is clipped to [-1, 1] by the
noise modes and "poisson" goes through ``(image + 1) / (max + 1)`` (skimage/util/noise.py, 0.18.3).  "poisson" is drawn on the
host (its level count is ``np.unique`` of the image) and added as a precomputed field.
"""
from __future__ import annotations

import ctypes as C
from itertools import permutations

import numpy as np
import torch

from . import _lib
from .watershed import get_context

NOISE_OPTIONS = ['gaussian', 'localvar', 'poisson', 'speckle', 'gaussian', 'speckle']      # augment.py:58


def continuous_choice(min_, max_, sigma, loc=0., size=1, rng=None):
    """augment.py:170-185: redraw N(loc, sigma) until it lies in [min_, max_]."""
    rng = np.random if rng is None else rng
    while True:
        out = rng.normal(loc=loc, scale=sigma, size=size)
        val = out if size == 1 else out.mean()
        if val >= min_ and val <= max_:
            return out


# ---------------------------------------------------------------------------------------------------------------------
# array <-> 32-bit words on the device
# ---------------------------------------------------------------------------------------------------------------------
def _image_f32(a):
    if isinstance(a, torch.Tensor):
        return a.to(device="cuda", dtype=torch.float32).contiguous()
    return torch.from_numpy(np.ascontiguousarray(np.asarray(a), dtype=np.float32)).cuda()


def _to_words(a):
    """(int32 word tensor on the device with a leading axis of 1 or 2 words per element, restore(words) -> like ``a``)."""
    if isinstance(a, torch.Tensor):
        dt = a.dtype
        t = a.to("cuda").contiguous()
        if t.element_size() == 4:
            return t.view(torch.int32)[None], lambda w: w[0].view(dt)
        if t.element_size() == 8:
            w = t.view(torch.int32).view(tuple(t.shape) + (2,))
            return torch.stack((w[..., 0], w[..., 1])).contiguous(), lambda w: torch.stack((w[0], w[1]), -1).contiguous().view(dt).squeeze(-1)
        wide = torch.float32 if dt.is_floating_point else torch.int32
        return t.to(wide).view(torch.int32)[None], lambda w: w[0].view(wide).to(dt)
    a = np.asarray(a)
    dt = a.dtype
    if dt.itemsize == 4:
        w = torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).cuda()[None]
        return w, lambda w: w[0].cpu().numpy().view(dt)
    if dt.itemsize == 8:
        v = np.ascontiguousarray(a).view(np.int32).reshape(a.shape + (2,))
        w = torch.from_numpy(np.ascontiguousarray(np.stack((v[..., 0], v[..., 1])))).cuda()
        return w, lambda w: np.ascontiguousarray(np.stack((w[0].cpu().numpy(), w[1].cpu().numpy()), -1)).view(dt).reshape(w.shape[1:])
    wide = np.float32 if dt.kind == 'f' else np.int32                        # 1- and 2-byte types widen exactly
    w = torch.from_numpy(np.ascontiguousarray(a.astype(wide)).view(np.int32)).cuda()[None]
    return w, lambda w: w[0].cpu().numpy().view(wide).astype(dt)


def _order4(ndim, axis, p, used_axes):
    """augment.py:89-117 for an array of ``ndim`` <= 4 axes, as (mirror axis, perm) of the 4-D array padded in front."""
    if ndim > 4:
        raise ValueError("augment_order: arrays of up to 4 axes")
    pad = 4 - ndim
    m4 = -1 if axis is None else pad + (axis % ndim)
    axes = list(range(ndim))
    if p is not None:
        for i, ax in enumerate(used_axes):
            na = p[i]
            axes[ax] = na + ndim if na < 0 else na
        if sorted(axes) != list(range(ndim)):
            raise ValueError(f"axes don't match array: {axes}")              # numpy's np.transpose error
    return m4, list(range(pad)) + [pad + a for a in axes]


def _launch(ctx, src, shape4, intensity, m4, perm4, out):
    gain = shift = clip_low = 0.0
    mode, noise = 0, None
    if intensity is not None:
        gain, shift, mode, noise, clip_low = intensity
    _lib.check(ctx.lib.isx_augment_array(ctx.handle, C.c_void_p(src.data_ptr()), (C.c_int64 * 4)(*shape4), int(intensity is not None),
                                         C.c_float(gain), C.c_float(shift), int(mode),
                                         C.c_void_p(noise.data_ptr()) if noise is not None else None, C.c_float(clip_low), int(m4),
                                         (C.c_int32 * 4)(*perm4), C.c_void_p(out.data_ptr())))


def _pass_order(a, axis, p, used_axes):
    """Mirror + permutation of one array as raw words: same dtype, every bit kept."""
    ndim = a.dim() if isinstance(a, torch.Tensor) else np.asarray(a).ndim
    m4, perm4 = _order4(ndim, axis, p, used_axes)
    words, restore = _to_words(a)
    outs = []
    for w in words:                                                          # one or two word planes
        s4 = (1,) * (4 - w.dim()) + tuple(w.shape)
        out = torch.empty(tuple(s4[k] for k in perm4), dtype=torch.int32, device=w.device)
        _launch(get_context(w.device.index), w, s4, None, m4, perm4, out)
        outs.append(out.view(tuple(out.shape[4 - w.dim():])))
    return restore(torch.stack(outs))


def _pass_image(image, plan, axis, p, used_axes):
    """The image: intensity and order in one pass; float32 result (numpy in -> numpy out)."""
    t = _image_f32(image)
    m4, perm4 = _order4(t.dim(), axis, p, used_axes)
    s4 = (1,) * (4 - t.dim()) + tuple(t.shape)
    out = torch.empty(tuple(s4[k] for k in perm4), dtype=torch.float32, device=t.device)
    _launch(get_context(t.device.index), t, s4, plan, m4, perm4, out)
    out = out.view(tuple(out.shape[4 - t.dim():]))
    return out if isinstance(image, torch.Tensor) else out.cpu().numpy()


# ---------------------------------------------------------------------------------------------------------------------
# decisions, in upstream's draw order
# ---------------------------------------------------------------------------------------------------------------------
def _intensity_plan(image, min_shift, max_shift, min_scale, max_scale, shift_sigma, scale_sigma, noise_prob, rng):
    """augment.py:49-71: (gain, shift, noise mode, noise field on the device, clip_low)."""
    t = _image_f32(image)
    ctx = get_context(t.device.index)
    s_out, m_out, lo_out = (C.c_double * 1)(), (C.c_float * 1)(), (C.c_float * 1)()
    v3 = t.view(-1, t.shape[-2], t.shape[-1])
    _lib.check(ctx.lib.isx_patch_stats(ctx.handle, C.c_void_p(v3.data_ptr()), _lib._dims(v3.shape), (C.c_int32 * 3)(0, 0, 0), 1,
                                       _lib._dims(v3.shape), s_out, m_out, lo_out))
    mx, mn = np.float32(m_out[0]), np.float32(lo_out[0])       # image.max() / .min(): two scalars back, like upstream's numpy
    scale = float(np.asarray(continuous_choice(min_scale, max_scale, scale_sigma, loc=1., rng=rng)).reshape(-1)[0])
    shift = float(np.asarray(continuous_choice(min_shift, max_shift, shift_sigma, rng=rng)).reshape(-1)[0])
    mode, noise, clip_low = 0, None, 0.0
    if rng.binomial(1, noise_prob):
        name = NOISE_OPTIONS[int(rng.randint(len(NOISE_OPTIONS)))]
        shape = tuple(t.shape)
        # random_noise's view of the image: float64 (float32 quotient) * scale + shift; its minimum decides the clip range
        # (scale > 0, so the minimum of the image maps to the minimum of the result)
        lo = float(np.float64(mn / mx) * scale + shift) if scale >= 0 else float(np.float64(np.float32(1.0)) * scale + shift)
        clip_low = -1.0 if lo < 0 else 0.0
        if name in ('gaussian', 'speckle'):
            noise = rng.normal(0., 0.001 ** 0.5, shape)          # var = 0.001 as upstream passes (augment.py:62)
            mode = 1 if name == 'gaussian' else 2
        elif name == 'localvar':
            noise = rng.normal(0, (np.zeros(shape, np.float64) + 0.01) ** 0.5)    # skimage's default local_vars
            mode = 1
        else:                                                    # poisson: skimage's arithmetic on the host, as a field to add
            base = (t.cpu().numpy() / mx).astype(np.float64) * scale + shift
            vals = 2 ** np.ceil(np.log2(len(np.unique(base))))
            img = base
            if clip_low == -1.0:
                old_max = img.max()
                img = (img + 1.) / (old_max + 1.)
            target = rng.poisson(img * vals) / float(vals)
            if clip_low == -1.0:
                target = target * (old_max + 1.) - 1.
            noise = target - base
            mode = 1
        noise = torch.from_numpy(np.ascontiguousarray(noise, dtype=np.float32)).cuda()
    return (scale / float(mx), shift, mode, noise, clip_low)


def _order_plan(mirror_prob, transpose_prob, used_axes, rng):
    """augment.py:84-104: (mirror axis or None, permutation of used_axes or None)."""
    axis = p = None
    if rng.binomial(1, mirror_prob):
        axis = used_axes[int(rng.randint(0, len(used_axes)))]
    if rng.binomial(1, transpose_prob):
        ps = [q for q in permutations(used_axes) if q != tuple(used_axes)]
        p = ps[int(rng.randint(0, len(ps)))]
    return axis, p


def augment_intensity(image, min_shift=-0.1, max_shift=0.1, min_scale=0.8, max_scale=1.2, shift_sigma=0.02, scale_sigma=.05,
                      noise_prob=0.3, verbose=False, rng=None):
    rng = np.random if rng is None else rng
    plan = _intensity_plan(image, min_shift, max_shift, min_scale, max_scale, shift_sigma, scale_sigma, noise_prob, rng)
    return _pass_image(image, plan, None, None, (-2, -1))


def augment_order(images, mirror_prob=0.2, transpose_prob=0.2, used_axes=(-2, -1), verbose=False, rng=None):
    rng = np.random if rng is None else rng
    axis, p = _order_plan(mirror_prob, transpose_prob, used_axes, rng)
    return [_pass_order(img, axis, p, used_axes) for img in images]


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
        axis, p = _order_plan(0.2, 0.2, (-2, -1), rng)
        image = _pass_image(image, plan, axis, p, (-2, -1))
        imgs = [_pass_order(a, axis, p, (-2, -1)) for a in imgs]
    result = [image]
    nl = len(keys) if keys is not None else 1
    result.append({k: imgs[i] for i, k in enumerate(keys)} if keys is not None else imgs[0])
    if ground_truth is not None:
        result.append(imgs[nl])
    return tuple(result)
