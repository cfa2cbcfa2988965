"""Host-side mirror of ``iterseg.watershed`` driving the HIP kernels through the C ABI.

Same function names, argument meaning and error behaviour as the reference's
``src/iterseg/watershed.py`` (so a caller of ``ws.segment_output_image`` /
``ws.affinity_watershed`` can switch modules), but every stage runs on the MI355X:

* :func:`segment_output_image`      -- watershed.py:165-223  -> ``isx_segment_output_image``
* :func:`affinity_watershed`        -- watershed.py:17-35    -> ``isx_affinity_watershed``
* :func:`_get_mask`                 -- watershed.py:226-229  -> ``isx_ws_mask``
* :func:`_get_centroids`            -- watershed.py:232-236  -> ``isx_ws_seeds``
* :func:`_remove_unwanted_objects`  -- watershed.py:239-251  -> ``isx_ws_filter``

Inputs may be numpy arrays (copied to the device, results copied back -- the reference's
calling convention) or CUDA torch tensors (stay resident; results are returned as tensors).
PyTorch is only the device-memory container here.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib

_CTX = {}


def get_context(device=None):
    """Process-wide :class:`_lib.Context` per device (created on first use; needs a GPU)."""
    if device is None:
        device = torch.cuda.current_device() if torch.cuda.is_available() else 0
    device = int(device)
    if device not in _CTX:
        _CTX[device] = _lib.Context(device)
    ctx = _CTX[device]
    if torch.cuda.is_available():
        ctx.set_stream(torch.cuda.current_stream(device).cuda_stream)
    return ctx


def _dev(x, dtype, device):
    """-> (contiguous CUDA tensor of dtype, was_numpy)."""
    if isinstance(x, torch.Tensor):
        return x.to(device=device, dtype=dtype).contiguous(), False
    t = torch.from_numpy(np.ascontiguousarray(np.asarray(x)))
    if t.dtype == torch.bool and dtype == torch.uint8:
        t = t.view(torch.uint8)
    return t.to(device=device, dtype=dtype, non_blocking=False).contiguous(), True


def _ptr(t):
    return C.c_void_p(t.data_ptr())


def _device_of(*xs):
    for x in xs:
        if isinstance(x, torch.Tensor) and x.is_cuda:
            return x.device
    return torch.device("cuda", torch.cuda.current_device())


def gaussian(img, sigma):
    """``skimage.filters.gaussian`` as used by the reference (float32, sigma entries 0/1/2)."""
    dev = _device_of(img)
    ctx = get_context(dev.index)
    t, was_np = _dev(img, torch.float32, dev)
    out = torch.empty_like(t)
    sig = (C.c_int32 * 3)(*([int(sigma)] * 3 if np.isscalar(sigma) else [int(s) for s in sigma]))
    _lib.check(ctx.lib.isx_ws_gaussian(ctx.handle, _ptr(t), _ptr(out), _lib._dims(t.shape), sig))
    return out.cpu().numpy() if was_np else out


def threshold_otsu(img):
    dev = _device_of(img)
    ctx = get_context(dev.index)
    t, _ = _dev(img, torch.float32, dev)
    th = C.c_float()
    _lib.check(ctx.lib.isx_ws_threshold_otsu(ctx.handle, _ptr(t), t.numel(), C.byref(th)))
    return np.float32(th.value)


def _get_mask(img, sigma=2):
    if sigma != 2:
        raise ValueError("the reference hard-codes sigma=2 (watershed.py:226)")
    dev = _device_of(img)
    ctx = get_context(dev.index)
    t, was_np = _dev(img, torch.float32, dev)
    mask = torch.empty(t.shape, dtype=torch.uint8, device=dev)
    _lib.check(ctx.lib.isx_ws_mask(ctx.handle, _ptr(t), _lib._dims(t.shape), _ptr(mask), None))
    return mask.cpu().numpy().astype(bool) if was_np else mask.bool()


def _get_centroids(cent, gaussian=True, threshold_abs=.04):
    dev = _device_of(cent)
    ctx = get_context(dev.index)
    t, was_np = _dev(cent, torch.float32, dev)
    cap = max(1024, t.numel() // 8)
    while True:
        coords = torch.empty((cap, 3), dtype=torch.int64, device=dev)
        cnt = C.c_int64()
        _lib.check(ctx.lib.isx_ws_seeds(ctx.handle, _ptr(t), _lib._dims(t.shape), 1 if gaussian else 0,
                                        C.c_float(threshold_abs), _ptr(coords), cap, C.byref(cnt)))
        if cnt.value <= cap:
            break
        cap = cnt.value
    coords = coords[:cnt.value]
    return coords.cpu().numpy() if was_np else coords


def _remove_unwanted_objects(mask, centroids, min_area=0, max_area=1000000):
    dev = _device_of(mask, centroids)
    ctx = get_context(dev.index)
    m, was_np = _dev(mask, torch.uint8, dev)
    c, _ = _dev(np.asarray(centroids).reshape(-1, 3) if not isinstance(centroids, torch.Tensor) else centroids,
                torch.int64, dev)
    n = c.shape[0]
    mout = torch.empty_like(m)
    cout = torch.empty((max(n, 1), 3), dtype=torch.int64, device=dev)
    nout = C.c_int64(0)
    _lib.check(ctx.lib.isx_ws_filter(ctx.handle, _ptr(m), _lib._dims(m.shape), int(min_area), int(max_area),
                                     _ptr(mout), _ptr(c) if n else None, n, _ptr(cout) if n else None,
                                     C.byref(nout)))
    cout = cout[:nout.value]
    if was_np:
        return mout.cpu().numpy().astype(bool), cout.cpu().numpy()
    return mout.bool(), cout


def connected_components(mask):
    """``ndi.label`` stand-in (6-connectivity): ``(roots, sizes)`` -- the smallest raveled index
    of each voxel's component (0xFFFFFFFF off-mask) and the voxel count stored at the roots."""
    dev = _device_of(mask)
    ctx = get_context(dev.index)
    m, was_np = _dev(mask, torch.uint8, dev)
    roots = torch.empty(m.shape, dtype=torch.int32, device=dev)
    sizes = torch.empty(m.shape, dtype=torch.int32, device=dev)
    _lib.check(ctx.lib.isx_ws_components(ctx.handle, _ptr(m), _lib._dims(m.shape), _ptr(roots), _ptr(sizes)))
    if was_np:
        return roots.cpu().numpy().view(np.uint32), sizes.cpu().numpy().view(np.uint32)
    return roots, sizes


def affinity_watershed(image, marker_coords, mask, scale=None, out=None, py_func=False):
    """Kernel-boundary entry (watershed.py:17-35).  ``image``: (3, Z, Y, X) zero-padded
    affinities, ``marker_coords``: (n, 3) padded coordinates, ``mask``: bool with a zero border
    (``None`` = whole interior), ``out``: flat label buffer written in place (uint32/int32).
    ``py_func`` is accepted for signature compatibility and ignored."""
    dev = _device_of(image, mask)
    ctx = get_context(dev.index)
    img, was_np = _dev(image, torch.float32, dev)
    shape = tuple(img.shape[1:])
    mk = np.asarray(marker_coords).reshape(-1, 3) if not isinstance(marker_coords, torch.Tensor) else marker_coords
    if len(mk) == 0:
        # same failure as the reference: _prep_data -> np.apply_along_axis on an empty array
        raise ValueError("Cannot apply_along_axis when any iteration dimensions are 0")
    seeds, _ = _dev(mk, torch.int64, dev)
    if mask is None:
        m = torch.zeros(shape, dtype=torch.uint8, device=dev)
        m[1:-1, 1:-1, 1:-1] = 1
    else:
        m, _ = _dev(mask, torch.uint8, dev)
    if out is None:
        out_host = np.zeros(int(np.prod(shape)), np.int32) if was_np else None
        lab = torch.zeros(shape, dtype=torch.int32, device=dev)
    elif isinstance(out, torch.Tensor):
        out_host, lab = None, out
    else:
        out_host = out
        lab = torch.from_numpy(np.ascontiguousarray(out).view(np.int32).reshape(shape)).to(dev)
    sc = None
    if scale is not None:
        s = np.abs(np.asarray(scale, np.float32)).reshape(-1)
        sc = (C.c_float * 3)(*(list(s) if s.size == 3 else [float(s[0])] * 3))
    _lib.check(ctx.lib.isx_affinity_watershed(ctx.handle, _ptr(img), _ptr(seeds), seeds.shape[0], _ptr(m),
                                              _ptr(lab), _lib._dims(shape), sc))
    if out_host is None:
        return lab.reshape(shape)
    out_host.reshape(-1).view(np.int32)[:] = lab.reshape(-1).cpu().numpy()
    return out_host.reshape(shape)


def segment_output_image(unet_output, affinities_channels, centroids_channel, thresholding_channel,
                         scale=None, absolute_thresh=None, out=None, py_func=False, ctx=None):
    """W0-W4 (watershed.py:165-223).  Returns ``(segmentation, seeds, mask)`` like the reference:
    unpadded labels, unpadded seed coordinates, padded boolean mask.  With ``out`` given
    (flat, padded size -- ``current_output.ravel()`` on the plugin path) labels are written in
    place.  Hard-coded like upstream: min_area=10, max_area=10_000_000, threshold_abs=0.04."""
    dev = _device_of(unet_output)
    if ctx is None:                      # ``ctx``: run on a caller-owned context (e.g. one with a flood shard set)
        ctx = get_context(dev.index)
    if isinstance(unet_output, torch.Tensor):
        fm, was_np = unet_output.squeeze().to(dev, torch.float32), False
    else:
        fm, was_np = torch.from_numpy(np.asarray(np.squeeze(unet_output), np.float32)).to(dev), True
    order = list(affinities_channels) + [thresholding_channel, centroids_channel]
    if order != [0, 1, 2, 3, 4] or fm.shape[0] != 5:
        fm = fm[order]
    fm = fm.contiguous()
    shape = tuple(fm.shape[1:])
    pshape = tuple(s + 2 for s in shape)
    pn = int(np.prod(pshape))
    out_host = None
    if out is None:
        lab = torch.zeros(pshape, dtype=torch.int32, device=dev)
        if was_np:
            out_host = np.zeros(pn, np.int32)
    elif isinstance(out, torch.Tensor):
        lab = out
    else:
        out_host = out
        lab = torch.from_numpy(np.ascontiguousarray(out).view(np.int32).reshape(pshape)).to(dev)
    sc = None
    if scale is not None:                # |scale| multiplies the affinities inside the library (watershed.py:23-24)
        sv = np.abs(np.asarray(scale, np.float32)).reshape(-1)
        sc = (C.c_float * 3)(*(list(sv) if sv.size == 3 else [float(sv[0])] * 3))
    cap = max(1024, fm[0].numel() // 8)
    at = C.byref(C.c_float(absolute_thresh)) if absolute_thresh is not None else None
    while True:
        seeds = torch.empty((cap, 3), dtype=torch.int64, device=dev)
        mask = torch.empty(pshape, dtype=torch.uint8, device=dev)
        ns = C.c_int64(0)
        _lib.check(ctx.lib.isx_segment_output_image_scaled(ctx.handle, _ptr(fm), _lib._dims(shape), _ptr(lab), 10,
                                                           10000000, at, _ptr(seeds), cap, C.byref(ns), _ptr(mask),
                                                           None, sc))
        if ns.value <= cap:
            break
        # more seeds than the report buffer (large plateaus keep every voxel as a seed): run again with room for all of
        # them, on a clean label buffer, as _get_centroids does
        cap = int(ns.value)
        lab.zero_()
    seeds = seeds[:ns.value]
    seg_dev = lab.reshape(pshape)[1:-1, 1:-1, 1:-1]
    if out_host is not None:
        out_host.reshape(-1).view(np.int32)[:] = lab.reshape(-1).cpu().numpy()
        seg = out_host.reshape(pshape)[1:-1, 1:-1, 1:-1]
        return seg, seeds.cpu().numpy(), mask.cpu().numpy().astype(bool)
    return seg_dev, seeds, mask.bool()
