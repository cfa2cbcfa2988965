"""Overlap-margin chunk geometry (host-side integer logic).

Mirrors ``iterseg.predict.make_chunks`` (reference: src/iterseg/predict.py:38-61):
the volume is cut into fixed-size chunks whose starts advance by ``chunk - 2*margin``;
the last start is pinned to ``size - chunk``; every chunk contributes only its centre
crop (first chunk keeps its low edge, last chunk keeps its high edge and takes
exactly what is left).  The device chunk scheduler consumes the flat int32 table
produced by :func:`chunk_table`.
"""
from __future__ import annotations

import itertools

import numpy as np


def _axis_plan(size, chunk, margin):
    """Starts and (lo, hi) crops along one axis; same arithmetic as predict.py:45-58."""
    size, chunk, margin = int(size), int(chunk), int(margin)
    step = chunk - 2 * margin
    if size < chunk:
        # The reference silently produces a negative start here (predict.py:48-49 is
        # undefined for size < chunk); a drop-in has nothing sane to mirror, so refuse.
        raise ValueError(f"volume axis of {size} is smaller than the chunk size {chunk}")
    if step <= 0 or size - 2 * margin <= 0:
        raise IndexError("chunk/margin combination yields no chunk starts")
    starts = list(range(0, size - 2 * margin, step))
    starts[-1] = size - chunk
    if len(starts) > 1 and starts[-1] == starts[-2]:
        starts.pop()                      # last step was a perfect fit
    crops = [[margin, chunk - margin] for _ in starts]
    crops[0][0] = 0
    taken = sum(hi - lo for lo, hi in crops[:-1])
    crops[-1][0] = chunk - (size - taken)
    crops[-1][1] = chunk
    return starts, [tuple(c) for c in crops]


def make_chunks(arr_shape, chunk_shape, margin):
    """Return ``(chunk_starts, chunk_crops)`` exactly like the reference.

    ``chunk_starts`` is a list of per-axis start tuples in C (first-axis-major) order,
    ``chunk_crops`` the matching list of per-axis ``(lo, hi)`` crops *relative to the
    chunk* (reference: predict.py:59-61).
    """
    ndim = len(arr_shape)
    if isinstance(margin, int):
        margin = [margin] * ndim
    plans = [_axis_plan(arr_shape[d], chunk_shape[d], margin[d]) for d in range(ndim)]
    chunk_starts = list(itertools.product(*[p[0] for p in plans]))
    chunk_crops = list(itertools.product(*[p[1] for p in plans]))
    return chunk_starts, chunk_crops


def chunk_table(arr_shape, chunk_shape, margin):
    """Flat ``int32 [n_chunks, 9]`` table ``(sz, sy, sx, lo_z, hi_z, lo_y, hi_y, lo_x, hi_x)``.

    This is the form handed to the C ABI (``isx_predict_volume``); 3-D only.
    """
    starts, crops = make_chunks(arr_shape, chunk_shape, margin)
    tab = np.empty((len(starts), 9), np.int32)
    for i, (s, c) in enumerate(zip(starts, crops)):
        tab[i, 0:3] = s
        tab[i, 3:9] = [v for lohi in c for v in lohi]
    return tab


def shard_range(n_items, rank, world_size):
    """Contiguous block partition ``[begin, end)`` of ``n_items`` units for ``rank``.

    Used for chunk sharding (single volume) and frame sharding (4-D stacks); the first
    ``n_items % world_size`` ranks get one extra unit.
    """
    base, rem = divmod(int(n_items), int(world_size))
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)
