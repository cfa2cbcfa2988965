"""CPU oracle for the training-target generation.  TEST INFRASTRUCTURE ONLY.

numpy restatement of ``get_training_labels`` (reference src/iterseg/labels.py:23-68): ``<axis>-<n>`` affinities
(labels.py:87-108), ``centreness`` and ``centreness-log`` (labels.py:142-205), ``offsets-<axis>`` (labels.py:212-276)
and ``mask`` (labels.py:283-285).  Pinned against tests/golden/training_labels.npz and training_labels_ext.npz
(generated from the reference with tests/golden/gen_stage3_labels.py) by tests/test_oracle_labels.py.
"""
import re

import numpy as np


def nth_affinity(labels, n, axis):
    """1.0 where voxel i and voxel i - n along ``axis`` carry different labels; the first n positions compare with the
    reflected index n - i (np.pad(..., mode='reflect'), labels.py:89)."""
    L = np.asarray(labels)
    idx = np.arange(L.shape[axis]) - n
    idx = np.where(idx < 0, -idx, idx)
    other = np.take(L, idx, axis=axis)
    return np.where(L - other != 0, 1.0, 0.0)


def get_centreness(labels, scale=(4, 1, 1), log=False):
    """Per object: 1 - d / d_max with d the scaled Euclidean distance (float64) of the voxel to the object's
    centroid; float32 result; single-voxel objects give 0/0 -> nan -> 0 (np.nan_to_num, labels.py:175).
    ``log`` (labels.py:191-201): d -> ln d where d > 0, then d += |min d| over the object."""
    L = np.asarray(labels)
    scale = np.asarray(scale, np.float64)
    out = np.zeros(L.shape, np.float32)
    ids = np.unique(L)
    coords = np.argwhere(L > 0)
    lab = L[L > 0]
    for i in ids[ids > 0]:
        pts = coords[lab == i]
        c = np.mean(pts, axis=0)
        diff = (c - pts) * scale
        d = np.sqrt(np.einsum("ij,ij->i", diff, diff))
        if log:
            pos = np.abs(d) > 0
            d = np.where(pos, np.log(np.where(pos, d, 1.0)), d)
            d = d + np.abs(d.min())
        with np.errstate(invalid="ignore", divide="ignore"):
            v = 1 - d / d.max()
        out[tuple(pts.T)] = v
    return np.nan_to_num(out)


def get_centre_offsets(labels, scale):
    """(3, Z, Y, X) float64 holding float32 values: per object and axis o = (centroid - voxel) * scale, o / max o for
    o > 0, -(o / min o) for o < 0, then (v + 1) / 2; background 0.5 (labels.py:212-276)."""
    L = np.asarray(labels)
    scale = np.asarray(scale, np.float64)
    new = np.where(np.stack([L > 0] * 3, axis=0), 0., 0.5)
    coords = np.argwhere(L > 0)
    lab = L[L > 0]
    for i in np.unique(lab):
        pts = coords[lab == i]
        diff = (np.mean(pts, axis=0) - pts) * scale
        for a in range(3):
            d = diff[:, a]
            with np.errstate(invalid="ignore", divide="ignore"):
                v = np.where(d > 0, d / d.max(), np.where(d < 0, -(d / d.min()), 0.0))
            v = ((v - (-1)) / 2).astype(np.float32)
            new[(a,) + tuple(pts.T)] += v
    return np.nan_to_num(new)


def smooth(image):
    """labels.py:312-325: skimage.filters.gaussian(plane, sigma=1) for every z plane = scipy's gaussian_filter with
    mode='nearest', truncate=4 on (Y, X) (scikit-image 0.18 `_shared/filters.py`), in the image's float dtype."""
    from scipy import ndimage as ndi
    return ndi.gaussian_filter(np.asarray(image, np.float64), (0, 1, 1), mode='nearest', truncate=4.0)


def get_gauss_centroids(labels):
    """labels.py:293-309: 1 at every object's rounded centroid (np.round), Gaussian (sigma 1) plane by plane in
    float64, then (out - min) / max."""
    L = np.asarray(labels)
    img = np.zeros(L.shape, np.float64)
    coords = np.argwhere(L > 0)
    lab = L[L > 0]
    for i in np.unique(lab):
        c = np.round(np.mean(coords[lab == i], axis=0)).astype(int)
        img[tuple(c)] = 1.
    out = smooth(img)
    out = out - out.min()
    return out / out.max()


def get_training_labels(l, channels=('z-1', 'y-1', 'x-1', 'centreness'), scale=(4, 1, 1)):
    labs = []
    offsets = get_centre_offsets(l, scale) if any(c.startswith('offsets-') for c in channels) else None
    for chan in channels:
        axis = {'z': 0, 'y': 1, 'x': 2}.get(chan[0])
        n = re.search(r'\d+', chan)
        if n is not None:
            a = nth_affinity(l, int(n[0]), axis)
            labs.append(smooth(a) if chan.endswith('-smooth') else a)
        elif chan == 'centreness':
            labs.append(get_centreness(l, scale))
        elif chan == 'centreness-log':
            labs.append(get_centreness(l, scale, log=True))
        elif chan == 'centroid-gauss':
            labs.append(get_gauss_centroids(l))
        elif chan.startswith('offsets-'):
            labs.append(offsets['zyx'.index(chan[-1])])
        elif chan == 'mask':
            labs.append(np.where(np.asarray(l) > 1, 1., 0.))
        else:
            raise ValueError(f'Unrecognised channel type: {chan}')
    return np.stack(labs, axis=0)
