"""Oracle of the VI / object-count scores against scikit-image 0.18.3 (fixtures from gen_stage4_metrics.py)."""
import numpy as np

from oracle import metrics as om


def test_vi_and_counts_match_skimage(golden):
    d = golden("metrics_vi")
    for tag in ("a", "b", "same"):
        vi = om.variation_of_information(d[f"{tag}.gt"], d[f"{tag}.mr"])
        assert np.allclose(vi, d[f"{tag}.vi"], rtol=1e-12, atol=1e-12), (tag, vi, d[f"{tag}.vi"])
        n_gt, n_mr, nd, pct = om.object_counts(d[f"{tag}.gt"], d[f"{tag}.mr"])
        assert [n_gt, n_mr, nd] == list(d[f"{tag}.counts"]) and pct == float(d[f"{tag}.count_diff_pct"])
