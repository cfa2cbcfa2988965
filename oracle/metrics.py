"""CPU oracle for the segmentation-assessment scores that can be pinned here.
TEST INFRASTRUCTURE ONLY.

numpy restatement of scikit-image 0.18.3 ``variation_of_information`` (skimage/metrics/_variation_of_information.py:
contingency table normalised to probabilities, ``H(1|0) = -sum_a px[a] sum_b q log2 q`` with ``q = p_ab / px[a]``, and
the symmetric term; returned as ``[h1g0, h0g1]``) and of the object-count difference of the reference's
``get_accuracy_metrics`` (src/iterseg/metrics.py:111-119).  Pinned by tests/golden/metrics_vi.npz
(tests/golden/gen_stage4_metrics.py).
"""
import numpy as np


def contingency(gt, mr):
    """unique (gt, mr) label pairs and their voxel counts."""
    key = (np.asarray(gt).astype(np.int64).ravel() << 32) | (np.asarray(mr).astype(np.int64).ravel() & 0xFFFFFFFF)
    u, c = np.unique(key, return_counts=True)
    return (u >> 32).astype(np.int64), (u & 0xFFFFFFFF).astype(np.int64), c.astype(np.int64)


def vi_from_pairs(a, b, c):
    n = float(c.sum())
    p = c / n
    ua, ia = np.unique(a, return_inverse=True)
    ub, ib = np.unique(b, return_inverse=True)
    px = np.bincount(ia, weights=p)
    py = np.bincount(ib, weights=p)
    qa = p / px[ia]
    qb = p / py[ib]
    h1g0 = -np.sum(px * np.bincount(ia, weights=qa * np.log2(qa), minlength=len(ua)))
    h0g1 = -np.sum(py * np.bincount(ib, weights=qb * np.log2(qb), minlength=len(ub)))
    return np.array([h1g0, h0g1])


def variation_of_information(gt, mr):
    return vi_from_pairs(*contingency(gt, mr))


def object_counts(gt, mr):
    n_gt, n_mr = np.unique(gt).size, np.unique(mr).size
    return n_gt, n_mr, n_mr - n_gt, (n_mr - n_gt) / n_gt * 100
