"""Mirror of the pinnable part of ``iterseg.metrics`` (reference: src/iterseg/metrics.py) on the MI355X.

* :func:`variation_of_information` -- what ``get_accuracy_metrics`` calls per chunk (metrics.py:105-108; scikit-image
  ``skimage.metrics.variation_of_information``): returns ``[H(mr|gt), H(gt|mr)]`` ("false splits, false merges").
* :func:`object_counts` -- metrics.py:111-119.
* :func:`contingency_table` -- the distinct ``(gt, mr)`` label pairs and their voxel counts (device sort + run lengths).

The average-precision part of the reference goes through ``umetrix``, which is not available in the build container;
it is not built (``get_accuracy_metrics(..., AP=True)`` raises ``NotImplementedError``).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from .watershed import get_context


def _dev_i32(x):
    if isinstance(x, torch.Tensor):
        return x.to(device=x.device if x.is_cuda else torch.device("cuda", torch.cuda.current_device()),
                    dtype=torch.int32).contiguous().reshape(-1)
    return torch.from_numpy(np.ascontiguousarray(x).astype(np.int32).reshape(-1)).cuda()


def contingency_table(gt, mr, cap=None):
    """-> (gt_labels int64 [m], mr_labels int64 [m], counts int64 [m]) as numpy arrays, pairs in ascending order."""
    a, b = _dev_i32(gt), _dev_i32(mr)
    if a.numel() != b.numel():
        raise ValueError("Input images must have the same shape.")        # skimage check_shape_equality
    ctx = get_context(a.device.index)
    cap = int(cap) if cap else 1 << 20
    while True:
        pairs = torch.empty(cap, dtype=torch.int64, device=a.device)
        counts = torch.empty(cap, dtype=torch.int64, device=a.device)
        m = C.c_int64(0)
        rc = ctx.lib.isx_label_contingency(ctx.handle, C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), a.numel(),
                                           C.c_void_p(pairs.data_ptr()), C.c_void_p(counts.data_ptr()), cap, C.byref(m))
        if rc != 0 and m.value > cap:
            cap = int(m.value)
            continue
        _lib.check(rc)
        break
    p = pairs[:m.value].cpu().numpy().view(np.uint64)
    return (p >> np.uint64(32)).astype(np.int64), (p & np.uint64(0xFFFFFFFF)).astype(np.int64), counts[:m.value].cpu().numpy()


def _vi_from_pairs(a, b, c):
    n = float(c.sum())
    p = c / n
    ua, ia = np.unique(a, return_inverse=True)
    ub, ib = np.unique(b, return_inverse=True)
    px = np.bincount(ia, weights=p)
    py = np.bincount(ib, weights=p)
    qa, qb = p / px[ia], p / py[ib]
    h1g0 = -np.sum(px * np.bincount(ia, weights=qa * np.log2(qa), minlength=len(ua)))
    h0g1 = -np.sum(py * np.bincount(ib, weights=qb * np.log2(qb), minlength=len(ub)))
    return np.array([h1g0, h0g1])


def variation_of_information(image0, image1):
    """``skimage.metrics.variation_of_information(image0, image1)`` (no ``ignore_labels``)."""
    return _vi_from_pairs(*contingency_table(image0, image1))


def object_counts(gt, mr):
    """(n_gt, n_mr, difference, difference in % of n_gt) with background counted like ``np.unique`` does upstream."""
    a, b, _ = contingency_table(gt, mr)
    n_gt, n_mr = int(np.unique(a).size), int(np.unique(b).size)
    return n_gt, n_mr, n_mr - n_gt, (n_mr - n_gt) / n_gt * 100


def get_accuracy_metrics(slices, gt_data, model_result, name="model", prefix="", VI=True, AP=False, ND=True,
                         out_path=None, exclude_chunks=10):
    """The VI / object-count columns of metrics.py:45-143 for chunks given as ``(slice tuple, crop tuple)`` pairs."""
    import pandas as pd
    if AP:
        raise NotImplementedError("average precision needs umetrix, which this build cannot pin")
    scores = {'VI: GT | Output': [], 'VI: Output | GT': [], 'Number objects (GT)': [], 'Number objects (model)': [],
              'Count difference': [], 'Count difference (%)': []}
    gt_data, model_result = np.asarray(gt_data), np.asarray(model_result)
    for s_, c_ in slices:
        gt = np.squeeze(gt_data[s_])[c_]
        mr = np.squeeze(model_result[s_])[c_]
        n_gt, n_mr, nd, ndp = object_counts(gt, mr)
        if n_gt > exclude_chunks + 1:
            if VI:
                vi = variation_of_information(gt, mr)
                scores['VI: GT | Output'].append(vi[0])
                scores['VI: Output | GT'].append(vi[1])
            if ND:
                scores['Count difference (%)'].append(ndp)
                scores['Number objects (GT)'].append(n_gt)
                scores['Number objects (model)'].append(n_mr)
                scores['Count difference'].append(nd)
    df = pd.DataFrame({k: v for k, v in scores.items() if len(v) > 0})
    df['model_name'] = [name] * len(df)
    return df
