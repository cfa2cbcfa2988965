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


def blob_feature_map_tiled(n, seed, volume_fraction=0.3, blob_sigma=3.0, workers=8):
    """An ``(n, n, n)`` structured input made of eight independently seeded ``(n/2)^3`` demo volumes side by side (seeds
    ``8*seed .. 8*seed+7``; each one is the reference's construction, watershed.py:256-262).  Components keep the size
    distribution of the demo volume (they only merge where two tiles happen to have foreground on both sides of a
    shared face), unlike ``blob_feature_map_mirrored`` whose mirror planes glue every touching component to its image.
    The tiles are built by child interpreters (scipy's distance transform is single-threaded; plain subprocesses, so it
    does not matter whether the caller holds a GPU context or has a ``__main__`` guard)."""
    import os
    import subprocess
    import sys
    import tempfile
    assert n % 2 == 0
    h = n // 2
    out = np.empty((5, n, n, n), np.float32)

    def place(k, t):
        z, y, x = (k >> 2) & 1, (k >> 1) & 1, k & 1
        out[:, z * h:(z + 1) * h, y * h:(y + 1) * h, x * h:(x + 1) * h] = t

    if workers <= 1:
        for k in range(8):
            place(k, blob_feature_map((h, h, h), 8 * seed + k, volume_fraction, blob_sigma))
        return out
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tmp = tempfile.mkdtemp(prefix="isx_tiles_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    code = ("import sys, numpy as np; sys.path.insert(0, %r); from tests.synth import blob_feature_map; "
            "np.save(sys.argv[1], blob_feature_map((%d, %d, %d), int(sys.argv[2]), %r, %r))" % (root, h, h, h, volume_fraction, blob_sigma))
    try:
        procs = [subprocess.Popen([sys.executable, "-c", code, os.path.join(tmp, f"t{k}.npy"), str(8 * seed + k)])
                 for k in range(8)]
        for k, p in enumerate(procs):
            if p.wait() != 0:
                raise RuntimeError("tile generator failed")
            place(k, np.load(os.path.join(tmp, f"t{k}.npy")))
            os.remove(os.path.join(tmp, f"t{k}.npy"))
    finally:
        import shutil
        shutil.rmtree(tmp, ignore_errors=True)
    return out
