#!/usr/bin/env python3
"""Golden-vector generator, stage 2: watershed post-processor (W0-W4).

Run ONLY in the build container, with the interpreter that has scikit-image 0.18.3 /
scipy 1.7.1:

    /opt/conda/bin/python3.9 -W ignore tests/golden/gen_stage2_watershed.py

It loads the reference's ``watershed.py`` *unmodified* by file path (read-only, from
/root/reference) with a ``numba`` shim whose ``jit`` returns the plain function carrying
``.py_func`` (numba 0.54.1 cannot initialise against numpy 1.26 here), and always calls
the ``py_func=True`` path, which the reference itself designates as equal to the jitted
kernel (watershed.py:25-28, :294).  Outputs: small ``.npz`` fixtures next to this script.
"""
import importlib.util
import os
import sys
import types

import numpy as np
from scipy import ndimage as ndi

HERE = os.path.dirname(os.path.abspath(__file__))

nb = types.ModuleType("numba")


def _jit(*a, **k):
    def deco(f):
        f.py_func = f
        return f
    return deco


nb.jit = _jit
sys.modules["numba"] = nb
spec = importlib.util.spec_from_file_location("ref_ws", "/root/reference/src/iterseg/watershed.py")
ws = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ws)

from skimage import data, filters  # noqa: E402


def save(name, **arrs):
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **arrs)
    print(name, {k: (v.shape, str(v.dtype)) for k, v in arrs.items() if hasattr(v, "shape")})


def smooth_field(shape, seed, sigma):
    rng = np.random.default_rng(seed)
    f = ndi.gaussian_filter(rng.standard_normal(shape), sigma)
    f = (f - f.min()) / (f.max() - f.min())
    return f.astype(np.float32)


def plugin_path(fm):
    """segmentation.py:190-194 call form; returns labels (uint32, unpadded) + intermediates."""
    fm = fm.copy()
    cur = np.pad(np.zeros(fm.shape[1:], np.uint32), 1)
    seg, seeds, mask = ws.segment_output_image(
        fm, affinities_channels=(0, 1, 2), thresholding_channel=3, centroids_channel=4,
        out=cur.ravel(), py_func=True)
    return cur, seeds, mask


def stage_intermediates(fm):
    """Re-run W1-W3 piecewise through the same reference helpers to pin every stage."""
    out = {}
    sm = filters.gaussian(fm[3], sigma=2)
    out["mask_smooth"] = sm
    out["thresh"] = np.asarray(filters.threshold_otsu(sm))
    mask = ws._get_mask(fm[3])
    out["mask_raw"] = mask
    out["cent_smooth"] = filters.gaussian(fm[4], sigma=(0, 1, 1))
    cents = ws._get_centroids(fm[4])
    out["seeds_raw"] = cents.astype(np.int64)
    out["seeds_raw_intensity"] = out["cent_smooth"][tuple(cents.T)]
    mp = np.pad(mask, 1, constant_values=0)
    m2, c2 = ws._remove_unwanted_objects(mp, cents + 1, min_area=10, max_area=10000000)
    out["mask_filtered_padded"] = m2
    out["seeds_filtered_padded"] = c2.astype(np.int64)
    aff = fm[[0, 1, 2]].copy()
    aff /= np.max(aff, axis=(1, 2, 3)).reshape((-1, 1, 1, 1))
    out["aff_padded"] = np.pad(aff, ((0, 0), (1, 1), (1, 1), (1, 1)))
    return out


def main():
    # ------------------------------------------------------------------ (a) U-Net feature map
    for tag in ("10x48x48", "9x40x56"):
        fm = np.load(os.path.join(HERE, f"predict_{tag}.npz"))["feature_map"]
        cur, seeds, mask = plugin_path(fm)
        inter = stage_intermediates(fm)
        save(f"ws_fm_{tag}", labels_padded=cur, **inter)
        print("  labelled", int((cur > 0).sum()), "seeds", len(inter["seeds_filtered_padded"]),
              "thresh", float(inter["thresh"]))

    # ------------------------------------------------------------------ (b) reference's own synthetic
    # watershed.py:256-262 construction (binary_blobs + EDT + Scharr), seeded, absolute threshold.
    for L, vf, seed in ((40, 0.35, 0), (32, 0.15, 1)):
        fg = data.binary_blobs(length=L, n_dim=3, volume_fraction=vf, seed=seed)
        cent = ndi.distance_transform_edt(fg)
        affs = [filters.scharr(fg.astype(float), axis=i) for i in range(3)]
        vol = np.stack(affs + [cent, fg], axis=0).astype(np.float32)
        seg, seeds, mask = ws.segment_output_image(vol.copy(), (0, 1, 2), 3, 4, absolute_thresh=0.5,
                                                   py_func=True)
        # kernel-boundary inputs, recomputed exactly as segment_output_image does
        aff = vol[[0, 1, 2]].copy()
        aff /= np.max(aff, axis=(1, 2, 3)).reshape((-1, 1, 1, 1))
        aff = np.pad(aff, ((0, 0), (1, 1), (1, 1), (1, 1)))
        save(f"ws_blobs{L}", volume=vol, labels=seg.astype(np.int32), seeds_padded=(seeds + 1).astype(np.int64),
             mask_padded=mask, aff_padded=aff,
             cent_smooth=filters.gaussian(vol[3], sigma=(0, 1, 1)),
             seeds_raw=ws._get_centroids(vol[3]).astype(np.int64))
        print("  blobs", L, "n seeds", len(seeds), "labelled", int((seg > 0).sum()), "ncomp", ndi.label(mask)[1])

    # ------------------------------------------------------------------ kernel-boundary cases (W4/W4a)
    def run_kernel(name, aff, seeds, mask, scale=None, out_dtype=None):
        out = None if out_dtype is None else np.zeros(mask.size, out_dtype)
        lab = ws.affinity_watershed(aff.copy(), seeds, mask, scale=scale, out=out, py_func=True)
        extra = {} if scale is None else {"scale": np.asarray(scale)}
        save(name, aff_padded=aff, seeds_padded=seeds.astype(np.int64), mask_padded=mask,
             labels_padded=lab, **extra)
        return lab

    def random_seeds(mask, n, rng):
        idx = np.flatnonzero(mask.ravel())
        pick = rng.choice(idx, size=min(n, len(idx)), replace=False)
        return np.stack(np.unravel_index(pick, mask.shape), axis=1)

    # (c) tie-heavy: quantised affinities, ~50 % random mask (giant + many small components)
    rng = np.random.default_rng(10)
    shp = (14, 30, 34)
    mask = np.pad(rng.random(shp) < 0.5, 1, constant_values=0)
    aff = np.pad((rng.integers(0, 5, (3,) + shp) / 4).astype(np.float32), ((0, 0), (1, 1), (1, 1), (1, 1)))
    seeds = random_seeds(mask, 60, rng)
    run_kernel("ws_k_ties", aff, seeds, mask, out_dtype=np.uint32)

    # (d) one giant component (full interior), few seeds, continuous affinities
    rng = np.random.default_rng(11)
    shp = (10, 24, 28)
    mask = np.pad(np.ones(shp, bool), 1, constant_values=0)
    aff = np.pad(rng.random((3,) + shp, dtype=np.float32), ((0, 0), (1, 1), (1, 1), (1, 1)))
    seeds = random_seeds(mask, 5, rng)
    run_kernel("ws_k_giant", aff, seeds, mask, out_dtype=np.uint32)
    # mask=None form of _prep_data (watershed.py:53-57) and int32 default output (:59-60)
    lab = ws.affinity_watershed(aff.copy(), seeds, None, py_func=True)
    save("ws_k_giant_nomask", aff_padded=aff, seeds_padded=seeds.astype(np.int64), labels_padded=lab)

    # (e) components without a seed stay 0; seeds adjacent to each other; smooth affinities
    shp = (12, 32, 32)
    f = smooth_field(shp, 12, 2.0)
    mask = np.pad(f > 0.55, 1, constant_values=0)
    rng = np.random.default_rng(12)
    aff = np.pad(np.stack([smooth_field(shp, 20 + i, 1.0) for i in range(3)]), ((0, 0), (1, 1), (1, 1), (1, 1)))
    lab_cc, ncc = ndi.label(mask)
    seeds = random_seeds(lab_cc == np.argmax(np.bincount(lab_cc.ravel())[1:]) + 1, 6, rng)
    seeds = np.concatenate([seeds, seeds[:2] + np.array([[0, 0, 1], [0, 1, 0]])])
    seeds = seeds[mask[tuple(seeds.T)]]
    seeds = np.unique(seeds, axis=0)[::-1].copy()          # reversed order: label ids follow list order
    run_kernel("ws_k_noseed_comp", aff, seeds, mask, out_dtype=np.uint32)
    print("  ncc", ncc)

    # (f) scale argument (watershed.py:23-24): per-axis (3,1) and a negative scalar
    run_kernel("ws_k_scale_axis", aff, seeds, mask, scale=np.array([[2.0], [1.0], [0.25]], np.float32),
               out_dtype=np.uint32)
    run_kernel("ws_k_scale_neg", aff, seeds, mask, scale=-3.0, out_dtype=np.uint32)

    # (g) negative / signed-zero affinities (Scharr-like), int32 output
    rng = np.random.default_rng(13)
    shp = (8, 20, 22)
    mask = np.pad(rng.random(shp) < 0.7, 1, constant_values=0)
    a = (rng.integers(-3, 4, (3,) + shp) / 2).astype(np.float32)
    a[a == 0] = np.where(rng.random((a == 0).sum()) < 0.5, np.float32(-0.0), np.float32(0.0))
    aff = np.pad(a, ((0, 0), (1, 1), (1, 1), (1, 1)))
    seeds = random_seeds(mask, 12, rng)
    run_kernel("ws_k_negvals", aff, seeds, mask, out_dtype=np.int32)

    # ------------------------------------------------------------------ W1-W3 stage pins on synthetic maps
    for i, (shp, sig) in enumerate((((12, 40, 44), 1.5), ((6, 70, 33), 3.0), ((20, 20, 20), 1.0), ((5, 64, 48), 1.2))):
        fm = np.stack([smooth_field(shp, 100 + 10 * i + c, sig) for c in range(5)])
        fm[4] = fm[4] ** 2                      # sparser centre map
        inter = stage_intermediates(fm)
        try:
            cur, _, _ = plugin_path(fm)
            err = ""
        except Exception as e:       # zero surviving seeds: np.apply_along_axis raises (watershed.py:50)
            cur = np.pad(np.zeros(fm.shape[1:], np.uint32), 1)
            err = type(e).__name__
        save(f"ws_stages_{i}", feature_map=fm, labels_padded=cur, error=np.asarray(err), **inter)
        print("  stages", i, "seeds", len(inter["seeds_raw"]), "->", len(inter["seeds_filtered_padded"]),
              "thresh", float(inter["thresh"]))

    # size filter with a small upper bound (exercises the XOR branch, watershed.py:244-247)
    rng = np.random.default_rng(14)
    m = np.pad(rng.random((10, 30, 30)) < 0.3, 1, constant_values=0)
    c = random_seeds(m, 200, rng)
    m2, c2 = ws._remove_unwanted_objects(m, c, min_area=10, max_area=60)
    save("ws_sizefilter", mask_padded=m, seeds_padded=c.astype(np.int64), mask_out=m2,
         seeds_out=c2.astype(np.int64), min_area=np.array(10), max_area=np.array(60))

    # plateaus / trivial images for W2 and constant image for W1
    plate = np.zeros((6, 12, 12), np.float32)
    plate[2:4, 3:6, 3:6] = 0.5
    plate[3, 8, 8] = 0.9
    plate[1, 1, 10] = 0.03
    pk = ws._get_centroids(plate, gaussian=False)
    const = np.full((5, 9, 9), 0.25, np.float32)
    pk_const = ws._get_centroids(const, gaussian=False)
    save("ws_plateau", image=plate, peaks=pk.astype(np.int64), const=const,
         peaks_const=pk_const.astype(np.int64).reshape(-1, 3),
         otsu_const=np.asarray(filters.threshold_otsu(const)))


if __name__ == "__main__":
    main()
