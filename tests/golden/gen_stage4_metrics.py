#!/usr/bin/env python3
"""Golden-vector generator, stage 4: segmentation assessment (the part that can be pinned).

Run ONLY in the build container (scikit-image 0.18.3 interpreter):

    /opt/conda/bin/python3.9 -W ignore tests/golden/gen_stage4_metrics.py

The reference's ``get_accuracy_metrics`` (src/iterseg/metrics.py:45-143) computes, per chunk, scikit-image's
``variation_of_information(gt, mr)`` (:105-108) and the object-count difference from ``np.unique`` (:111-119); its
average-precision part calls ``umetrix``, which is in neither interpreter of this container, so it is not captured.
"""
import os

import numpy as np
from scipy import ndimage as ndi
from skimage.metrics import variation_of_information

HERE = os.path.dirname(os.path.abspath(__file__))


def labels(shape, seed, density):
    rng = np.random.default_rng(seed)
    f = ndi.gaussian_filter(rng.random(shape), (1.0, 2.0, 2.0))
    return ndi.label(f > np.quantile(f, 1.0 - density))[0].astype(np.int32)


def main():
    out = {}
    for tag, shape, s0, s1, d0, d1 in (("a", (10, 48, 48), 1, 1, 0.3, 0.33), ("b", (7, 33, 50), 2, 3, 0.25, 0.4),
                                      ("same", (6, 24, 24), 5, 5, 0.3, 0.3)):
        gt, mr = labels(shape, s0, d0), labels(shape, s1, d1)
        vi = variation_of_information(gt, mr)
        n_gt, n_mr = np.unique(gt).size, np.unique(mr).size
        out[f"{tag}.gt"], out[f"{tag}.mr"] = gt, mr
        out[f"{tag}.vi"] = np.asarray(vi, np.float64)
        out[f"{tag}.counts"] = np.array([n_gt, n_mr, n_mr - n_gt], np.int64)
        out[f"{tag}.count_diff_pct"] = np.float64((n_mr - n_gt) / n_gt * 100)
        print(tag, vi, n_gt, n_mr)
    np.savez_compressed(os.path.join(HERE, "metrics_vi.npz"), **out)


if __name__ == "__main__":
    main()
