"""CPU restatement of the patch sampler / augmentation (oracle/train_io.py): its own invariants (no fixture exists upstream)."""
import numpy as np

from oracle import train_io as otio


def test_sampler_crops_are_in_bounds_bright_and_reproducible():
    rng = np.random.default_rng(0)
    vol = rng.random((20, 64, 80)).astype(np.float32)
    vol[:, :20] *= 0.01
    a = otio.get_image_chunks(vol, (5, 16, 24), 8, 0.3, np.random.RandomState(4))
    b = otio.get_image_chunks(vol, (5, 16, 24), 8, 0.3, np.random.RandomState(4))
    assert a['slices'] == b['slices'] and a['n'] == 8
    for s, x in zip(a['slices'], a['x']):
        assert x.shape == (5, 16, 24) and x.mean() / x.max() > 0.3
        # upstream's randint upper bound leaves the last two start positions unused (train_io.py:243-244)
        assert all(sl.start <= dim - c - 2 for sl, dim, c in zip(s, vol.shape, (5, 16, 24)))


def test_order_and_intensity_restatements():
    a = np.arange(2 * 3 * 4).reshape(2, 3, 4)
    assert np.array_equal(otio.apply_order(a, 1, 0), a[:, ::-1, :])
    assert np.array_equal(otio.apply_order(a, 2, 1), np.swapaxes(a[:, :, ::-1], -2, -1))
    img = np.array([[[0.0, 2.0, 4.0]]], np.float32)
    assert np.allclose(otio.intensity(img, 1.0, 0.0), [[[0.0, 0.5, 1.0]]])
    assert np.allclose(otio.intensity(img, 1.2, 0.1), [[[0.1, 0.7, 1.0]]])
    assert np.allclose(otio.intensity(img, 1.0, 0.0, 2, np.full(img.shape, 0.5, np.float32)), [[[0.0, 0.75, 1.0]]])
