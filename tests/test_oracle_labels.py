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
