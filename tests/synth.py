"""Seeded synthetic inputs shared by tests, smoke() and bench.py (numpy/scipy only -- nothing
here needs the reference or scikit-image)."""
import numpy as np
from scipy import ndimage as ndi


def positive_volume(shape, seed):
    """uniform (0, 1] float32, divided by its max (segmentation.py:889)."""
    rng = np.random.default_rng(seed)
    v = 1.0 - rng.random(shape, dtype=np.float32)
    return (v / v.max()).astype(np.float32)


def _field(shape, rng, sigma):
    f = ndi.gaussian_filter(rng.standard_normal(shape).astype(np.float32), sigma)
    return ((f - f.min()) / (f.max() - f.min())).astype(np.float32)


def smooth_feature_map(shape, seed, sigma=1.5):
    """A (5, Z, Y, X) float32 'U-Net output' with structure at the scale of a few voxels:
    three affinity-like channels, a mask channel and a sparse centre channel."""
    rng = np.random.default_rng(seed)
    fm = np.stack([_field(shape, rng, sigma) for _ in range(5)])
    fm[3] = _field(shape, rng, 2.5 * sigma)
    fm[4] = fm[4] ** 2
    return np.ascontiguousarray(fm, np.float32)


def blob_feature_map(shape, seed, volume_fraction=0.3, blob_sigma=3.0):
    """Structured watershed input in the spirit of the reference's self-check
    (watershed.py:256-262): thresholded smooth noise as foreground, EDT as the centre map,
    gradient magnitude per axis as affinities.  Many separate mask components."""
    rng = np.random.default_rng(seed)
    f = ndi.gaussian_filter(rng.standard_normal(shape).astype(np.float32), blob_sigma)
    fg = f > np.quantile(f, 1.0 - volume_fraction)
    cent = ndi.distance_transform_edt(fg).astype(np.float32)
    cent /= max(cent.max(), 1e-6)
    affs = [np.abs(ndi.sobel(fg.astype(np.float32), axis=a)) for a in range(3)]
    affs = [a / max(a.max(), 1e-6) for a in affs]
    soft = ndi.gaussian_filter(fg.astype(np.float32), 1.0)
    return np.ascontiguousarray(np.stack(affs + [soft, cent]), np.float32)


def blob_feature_map_mirrored(n, seed, volume_fraction=0.3, blob_sigma=3.0):
    """``blob_feature_map`` at ``(n, n, n)`` built from the ``(n/2)^3`` one by mirror tiling (2 x 2 x 2 reflections).  This
    IS the same construction applied to the mirrored foreground -- the distance transform and the reflect-mode filters of
    a mirrored image are the mirrored results -- at an eighth of the scipy time (the 512^3 bench input in seconds);
    components that touch a mirror plane merge with their images, so the largest ones are up to 8x larger."""
    assert n % 2 == 0
    h = blob_feature_map((n // 2,) * 3, seed, volume_fraction, blob_sigma)
    for ax in (1, 2, 3):
        h = np.concatenate([h, np.flip(h, axis=ax)], axis=ax)
    return np.ascontiguousarray(h, np.float32)
