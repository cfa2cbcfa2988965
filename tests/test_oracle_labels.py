"""Oracle of the training-target generation against the vectors captured from the reference's labels.py."""
import numpy as np

from oracle import labels as ol


def test_training_labels_match_reference(golden):
    d = golden("training_labels")
    for tag in ("a", "b"):
        chans = tuple(str(c) for c in d[f"{tag}.channels"])
        got = ol.get_training_labels(d[f"{tag}.labels"].astype(np.int64), chans, tuple(d[f"{tag}.scale"]))
        ref = d[f"{tag}.targets"]
        assert got.shape == ref.shape
        for i, c in enumerate(chans):
            if c == "centreness":        # float64 distances through np.linalg.norm upstream: allow the last float32 bit
                assert np.abs(got[i] - ref[i]).max() <= 2e-7, (tag, c, np.abs(got[i] - ref[i]).max())
            else:
                assert np.array_equal(got[i], ref[i]), (tag, c)


def test_offsets_and_log_centreness_match_reference(golden):
    """labels.py:212-276 (centre offsets) and labels.py:178-205 with log=True; offsets are exact (float32 values of
    float64 arithmetic), the log variant goes through np.log / np.linalg.norm upstream: last float32 bit."""
    d = golden("training_labels_ext")
    for tag in ("c", "d", "e", "f"):
        chans = tuple(str(c) for c in d[f"{tag}.channels"])
        got = ol.get_training_labels(d[f"{tag}.labels"].astype(np.int64), chans, tuple(d[f"{tag}.scale"]))
        ref = d[f"{tag}.targets"]
        assert got.shape == ref.shape
        for i, c in enumerate(chans):
            tol = 0.0 if c.startswith("offsets-") or c in ("x-1", "mask") else 2e-7
            assert np.abs(got[i] - ref[i]).max() <= tol, (tag, c, np.abs(got[i] - ref[i]).max())
