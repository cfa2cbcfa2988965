"""CPU oracle for the watershed post-processor -- reference-shaped Python front end.

TEST INFRASTRUCTURE ONLY: imported by tests/, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py``; never by anything under ``iterseg_amd/``.

Thin ctypes layer over ``oracle/ws_oracle.c`` (plain C restatement).  Function names
and argument meaning follow the reference's ``src/iterseg/watershed.py`` so that the
parity tests read like calls into the reference:

* :func:`segment_output_image`   -- watershed.py:165-223
* :func:`affinity_watershed`     -- watershed.py:17-35 (+ ``_prep_data`` :38-63)
* :func:`_get_mask`              -- watershed.py:226-229
* :func:`_get_centroids`         -- watershed.py:232-236
* :func:`_remove_unwanted_objects` -- watershed.py:239-251
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")


def build(force=False):
    so = os.path.join(_HERE, "_build", "libws_oracle.so")
    src = os.path.join(_HERE, "ws_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.orc_gaussian.argtypes = [_f32p, _f32p, _i64p, np.ctypeslib.ndpointer(np.int32)]
        L.orc_gaussian.restype = None
        L.orc_threshold_otsu.argtypes = [_f32p, C.c_int64]
        L.orc_threshold_otsu.restype = C.c_float
        L.orc_minmax.argtypes = [_f32p, C.c_int64, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.orc_hist_edges.argtypes = [C.c_float, C.c_float, _f32p]
        L.orc_histogram256.argtypes = [_f32p, C.c_int64, C.c_float, C.c_float, _f32p, _i64p]
        L.orc_get_mask.argtypes = [_f32p, _i64p, _u8p, C.c_void_p]
        L.orc_get_mask.restype = C.c_float
        L.orc_peak_local_max.argtypes = [_f32p, _i64p, C.c_float, _i64p, C.c_int64]
        L.orc_peak_local_max.restype = C.c_int64
        L.orc_label_filter.argtypes = [_u8p, _i64p, C.c_int64, C.c_int64, _u8p, C.c_void_p]
        L.orc_label_filter.restype = C.c_int64
        L.orc_affinity_watershed.argtypes = [_f32p, _i64p, C.c_int64, _u8p, _u32p, _i64p]
        L.orc_affinity_watershed.restype = C.c_int64
        L.orc_normalise_pad.argtypes = [_f32p, _i64p, _f32p]
        L.orc_segment_output_image.argtypes = [_f32p, _i64p, _u32p, C.POINTER(C.c_float), C.c_void_p,
                                               C.c_int64, C.c_void_p, C.c_int64, C.c_int64]
        L.orc_segment_output_image.restype = C.c_int64
        L.orc_segment_output_image_mt.argtypes = L.orc_segment_output_image.argtypes + [C.c_int]
        L.orc_segment_output_image_mt.restype = C.c_int64
        L.orc_affinity_watershed_mt.argtypes = L.orc_affinity_watershed.argtypes + [C.c_int]
        L.orc_affinity_watershed_mt.restype = C.c_int64
        _LIB = L
    return _LIB


def _dims(shape):
    return np.asarray(shape, np.int64)


def gaussian(img, sigma):
    """``skimage.filters.gaussian`` for float32 3-D input; sigma entries in {0, 1, 2}."""
    img = np.ascontiguousarray(img, np.float32)
    sig = np.asarray([sigma] * 3 if np.isscalar(sigma) else sigma, np.int32)
    out = np.empty_like(img)
    lib().orc_gaussian(img, out, _dims(img.shape), sig)
    return out


def threshold_otsu(img):
    img = np.ascontiguousarray(img, np.float32)
    return np.float32(lib().orc_threshold_otsu(img.reshape(-1), img.size))


def histogram256(img):
    a = np.ascontiguousarray(img, np.float32).reshape(-1)
    mn, mx = C.c_float(), C.c_float()
    lib().orc_minmax(a, a.size, C.byref(mn), C.byref(mx))
    edges = np.empty(257, np.float32)
    counts = np.empty(256, np.int64)
    lib().orc_hist_edges(mn, mx, edges)
    lib().orc_histogram256(a, a.size, mn, mx, edges, counts)
    return counts, edges


def _get_mask(img, sigma=2):
    assert sigma == 2
    img = np.ascontiguousarray(img, np.float32)
    mask = np.empty(img.shape, np.uint8)
    lib().orc_get_mask(img, _dims(img.shape), mask, None)
    return mask.astype(bool)


def _get_centroids(cent, gaussian=True, threshold_abs=np.float32(0.04)):
    g = globals()["gaussian"](cent, (0, 1, 1)) if gaussian else np.ascontiguousarray(cent, np.float32)
    idx = np.empty(g.size, np.int64)
    n = lib().orc_peak_local_max(g, _dims(g.shape), np.float32(threshold_abs), idx, idx.size)
    return np.stack(np.unravel_index(idx[:n], g.shape), axis=1).astype(np.int64).reshape(-1, 3)


def _remove_unwanted_objects(mask, centroids, min_area=0, max_area=1000000):
    m = np.ascontiguousarray(mask, np.uint8)
    out = np.empty_like(m)
    lib().orc_label_filter(m, _dims(m.shape), int(min_area), int(max_area), out, None)
    new_mask = out.astype(bool)
    centroids = np.asarray(centroids).reshape(-1, 3)
    keep = new_mask[tuple(centroids.T)] if len(centroids) else np.zeros(0, bool)
    return new_mask, centroids[keep]


def affinity_watershed(image, marker_coords, mask, scale=None, out=None, threads=0):
    """Kernel-boundary entry: ``image`` (3,Z,Y,X) f32 zero-padded, ``marker_coords`` (n,3)."""
    image = np.ascontiguousarray(image, np.float32)
    shape = image.shape[1:]
    raveled = image.reshape(3, -1).copy()
    if scale is not None:
        raveled *= np.abs(scale).astype(np.float32) if np.ndim(scale) else np.float32(abs(scale))
    marker_coords = np.asarray(marker_coords, np.int64).reshape(-1, 3)
    if len(marker_coords) == 0:
        raise ValueError("Cannot apply_along_axis when any iteration dimensions are 0")
    seeds = np.ravel_multi_index(tuple(marker_coords.T), shape).astype(np.int64)
    if mask is None:
        mask = np.pad(np.ones([s - 2 for s in shape], bool), 1, constant_values=0)
    m = np.ascontiguousarray(mask, np.uint8).reshape(-1)
    if out is None:
        out = np.zeros(m.size, np.int32)
    if threads > 0:
        lib().orc_affinity_watershed_mt(raveled, seeds, len(seeds), m, out.view(np.uint32), _dims(shape), int(threads))
    else:
        lib().orc_affinity_watershed(raveled, seeds, len(seeds), m, out.view(np.uint32), _dims(shape))
    return out.reshape(shape)


def segment_output_image(unet_output, affinities_channels=(0, 1, 2), centroids_channel=4,
                         thresholding_channel=3, out=None, min_area=10, max_area=10000000, threads=0):
    """Plugin-path form (segmentation.py:190-194).  Returns ``(segmentation, seeds, mask)``.
    ``threads`` > 0: the flood runs component by component on that many threads (same labels)."""
    fm = np.asarray(np.squeeze(unet_output), np.float32)
    order = list(affinities_channels) + [thresholding_channel, centroids_channel]
    fm = np.ascontiguousarray(fm[order])
    shape = fm.shape[1:]
    pshape = tuple(s + 2 for s in shape)
    if out is None:
        out = np.zeros(int(np.prod(pshape)), np.int32)
    labels = out.view(np.uint32)
    seeds = np.empty(fm[0].size, np.int64)
    maskp = np.empty(pshape, np.uint8)
    th = C.c_float()
    ns = lib().orc_segment_output_image_mt(fm, _dims(shape), labels, C.byref(th),
                                           seeds.ctypes.data_as(C.c_void_p), seeds.size,
                                           maskp.ctypes.data_as(C.c_void_p), int(min_area), int(max_area), int(threads))
    if ns == 0:
        raise ValueError("Cannot apply_along_axis when any iteration dimensions are 0")
    seg = out.reshape(pshape)[1:-1, 1:-1, 1:-1]
    sc = np.stack(np.unravel_index(seeds[:ns], pshape), axis=1) - 1
    return seg, sc, maskp.astype(bool)
