"""GPU parity of the training-target generation (isx_training_labels) against the vectors captured from the
reference's labels.py and against the numpy oracle on a larger seeded label volume.
Tolerance: affinities, mask and centre offsets bit-exact; centreness (plain and log) 2e-7 absolute (float64 distances,
float32 result)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_training_labels_match_reference(golden):
    from iterseg_amd import labels as gl
    d = golden("training_labels")
    for tag in ("a", "b"):
        chans = tuple(str(c) for c in d[f"{tag}.channels"])
        got = gl.get_training_labels(d[f"{tag}.labels"], chans, tuple(d[f"{tag}.scale"]))
        ref = d[f"{tag}.targets"]
        assert got.shape == ref.shape and got.dtype == np.float32
        for i, c in enumerate(chans):
            if c == "centreness":
                assert np.abs(got[i] - ref[i]).max() <= 2e-7, (tag, c, np.abs(got[i] - ref[i]).max())
            else:
                assert np.array_equal(got[i], ref[i]), (tag, c)


def test_training_labels_large_volume_vs_oracle():
    from scipy import ndimage as ndi
    from oracle import labels as ol
    from iterseg_amd import labels as gl
    rng = np.random.default_rng(4)
    f = ndi.gaussian_filter(rng.random((20, 160, 144)), (1.0, 2.5, 2.5))
    lab, n = ndi.label(f > np.quantile(f, 0.8))
    chans = ('z-1', 'y-1', 'x-1', 'centreness', 'mask', 'x-4')
    ref = ol.get_training_labels(lab.astype(np.int64), chans, (4, 1, 1))
    got = gl.get_training_labels(torch.from_numpy(lab.astype(np.int32)).cuda(), chans, (4, 1, 1)).cpu().numpy()
    assert n > 50
    for i, c in enumerate(chans):
        tol = 2e-7 if c == "centreness" else 0.0
        assert np.abs(got[i] - ref[i]).max() <= tol, (c, np.abs(got[i] - ref[i]).max())


def test_offsets_and_log_centreness_match_reference(golden):
    """Centre offsets (labels.py:212-276) bit-exact, centreness-log (labels.py:178-205, log=True) within 2e-7."""
    from iterseg_amd import labels as gl
    d = golden("training_labels_ext")
    for tag in ("c", "d", "e", "f"):         # e: '<axis>-<n>-smooth' (float64 Gaussian upstream, float32 here: 2e-7)
        chans = tuple(str(c) for c in d[f"{tag}.channels"])
        got = gl.get_training_labels(d[f"{tag}.labels"], chans, tuple(d[f"{tag}.scale"]))
        ref = d[f"{tag}.targets"]
        assert got.shape == ref.shape and got.dtype == np.float32
        for i, c in enumerate(chans):
            tol = 0.0 if c.startswith("offsets-") or c in ("x-1", "mask") else 2e-7
            if c == "centroid-gauss":        # float32 Gaussian passes, then divided by the maximum (about 0.16)
                tol = 1e-6
            assert np.abs(got[i] - ref[i]).max() <= tol, (tag, c, np.abs(got[i] - ref[i]).max())


def test_offsets_and_log_centreness_large_volume_vs_oracle():
    from scipy import ndimage as ndi
    from oracle import labels as ol
    from iterseg_amd import labels as gl
    rng = np.random.default_rng(11)
    f = ndi.gaussian_filter(rng.random((16, 120, 136)), (1.0, 2.5, 2.5))
    lab, n = ndi.label(f > np.quantile(f, 0.8))
    chans = ('offsets-z', 'centreness-log', 'offsets-x', 'centreness', 'offsets-y')
    ref = ol.get_training_labels(lab.astype(np.int64), chans, (4, 1, 1))
    got = gl.get_training_labels(torch.from_numpy(lab.astype(np.int32)).cuda(), chans, (4, 1, 1)).cpu().numpy()
    assert n > 40
    for i, c in enumerate(chans):
        tol = 0.0 if c.startswith("offsets-") else 2e-7
        assert np.abs(got[i] - ref[i]).max() <= tol, (c, np.abs(got[i] - ref[i]).max())
    assert np.array_equal(gl.get_centre_offsets(lab.astype(np.int32), (4, 1, 1)), ref[[0, 4, 2]].astype(np.float32))


def test_errors_like_reference():
    from iterseg_amd import labels as gl
    lab = np.zeros((4, 8, 8), np.int32)
    with pytest.raises(ValueError):
        gl.get_training_labels(lab, ('bogus',))
    with pytest.raises(ValueError):
        gl.get_training_labels(lab, ('offsets-q',))


def test_single_channel_helpers(golden):
    from iterseg_amd import labels as gl
    d = golden("training_labels")
    lab, ref = d["a.labels"], d["a.targets"]
    assert np.array_equal(gl.nth_affinity(lab, 1, 0), ref[0])
    assert np.abs(gl.get_centreness(lab, (4, 1, 1)) - ref[3]).max() <= 2e-7
    assert np.array_equal(gl.get_semantic_labels(lab), (lab > 1).astype(np.float32))
