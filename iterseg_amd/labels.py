"""Mirror of the target-generating part of ``iterseg.labels`` (reference: src/iterseg/labels.py) on the MI355X.

``get_training_labels(l, channels, scale)`` -- labels.py:23-68 -- for ``<axis>-<n>`` affinities, ``'centreness'``,
``'centreness-log'``, ``'offsets-<axis>'``, ``'mask'`` and the ``'<axis>-<n>-smooth'`` affinities (the only kind the
reference's ``-smooth`` suffix parses for: labels.py:36-65).  ``l`` may be a numpy array (result: numpy, float32) or a
CUDA tensor (result stays on the device); ``'centroid-gauss'`` (labels.py:293-309) too.
"""
from __future__ import annotations

import ctypes as C
import re

import numpy as np
import torch

from . import _lib
from .watershed import get_context

_KIND = {"affinity": 0, "centreness": 1, "mask": 2}


def _parse(channels):
    kind, axis, n = [], [], []
    for chan in channels:
        m = re.search(r'\d+', chan)
        if chan == 'centroid-gauss':
            kind.append(5); axis.append(0); n.append(0)
        elif chan == 'centreness-log':
            kind.append(3); axis.append(0); n.append(0)
        elif chan.startswith('offsets-'):
            if chan[-1] not in 'zyx':
                raise ValueError(f'Incompatible offset axis name: {chan}')        # labels.py:71-80
            kind.append(4); axis.append('zyx'.index(chan[-1])); n.append(0)
        elif m is not None and chan[0] in 'zyx':
            kind.append(0); axis.append('zyx'.index(chan[0])); n.append(int(m[0]))
        elif chan == 'centreness':
            kind.append(1); axis.append(0); n.append(0)
        elif chan == 'mask':
            kind.append(2); axis.append(0); n.append(0)
        else:
            msg = f'Unrecognised channel type: {chan} \n'                     # labels.py:60-64
            msg += 'Please enter str of form <axis>-<n> for nth affinity (e.g., z-1), \n'
            msg += 'centreness for centreness score (option of -log for log of centreness),\n'
            msg += 'or offset-<axis> (e.g., offset-z) for axis offsets'
            raise ValueError(msg)
    return kind, axis, n


def get_training_labels(l, channels=('z-1', 'y-1', 'x-1', 'centreness'), scale=(4, 1, 1)):
    kind, axis, n = _parse(channels)
    was_np = not isinstance(l, torch.Tensor)
    dev = torch.device("cuda", torch.cuda.current_device()) if was_np else l.device
    lab = (torch.from_numpy(np.ascontiguousarray(l).astype(np.int32)) if was_np else l.to(torch.int32)).to(dev).contiguous()
    if lab.dim() != 3:
        raise ValueError("labels must be a (Z, Y, X) volume")
    ctx = get_context(dev.index)
    out = torch.empty((len(channels),) + tuple(lab.shape), dtype=torch.float32, device=dev)
    nc = len(channels)
    arr = lambda v: (C.c_int32 * nc)(*v)
    _lib.check(ctx.lib.isx_training_labels(ctx.handle, C.c_void_p(lab.data_ptr()), _lib._dims(lab.shape), nc, arr(kind),
                                           arr(axis), arr(n), (C.c_double * 3)(*[float(s) for s in scale]),
                                           C.c_void_p(out.data_ptr())))
    for i, chan in enumerate(channels):
        if chan.endswith('-smooth'):              # labels.py:64-65,312-325: skimage gaussian(sigma=1) plane by plane
            from . import watershed
            out[i] = watershed.gaussian(out[i], (0, 1, 1))
    return out.cpu().numpy() if was_np else out


def nth_affinity(labels, n, axis):
    """labels.py:87-108"""
    return get_training_labels(labels, ('zyx'[axis] + f'-{n}',))[0]


def get_centreness(labels, scale=(4, 1, 1), log=False, power=False):
    """labels.py:142-178 (``power`` is not on the accelerated path)."""
    if power:
        raise NotImplementedError("power centreness")
    return get_training_labels(labels, ('centreness-log' if log else 'centreness',), scale)[0]


def get_centre_offsets(labels, scale):
    """labels.py:212-242 -> (3, Z, Y, X)"""
    return get_training_labels(labels, ('offsets-z', 'offsets-y', 'offsets-x'), scale)


def get_semantic_labels(labels):
    """labels.py:283-285"""
    return get_training_labels(labels, ('mask',))[0]
