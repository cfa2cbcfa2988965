"""Patch sampler and augmentation (src/iterseg/train_io.py:224-341, augment.py:8-118) on the GPU against the CPU restatement
(oracle/train_io.py; parity unpinned -- the reference's module cannot be imported here) and through their properties."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class ScriptedRNG:
    """Replays a fixed script of decisions (binomial / randint) and forwards the field draws to a seeded RandomState."""

    def __init__(self, binom, ints, seed=0):
        self.binom, self.ints, self.rs = list(binom), list(ints), np.random.RandomState(seed)

    def binomial(self, n, p):
        return self.binom.pop(0)

    def randint(self, low, high=None):
        return self.ints.pop(0)

    def normal(self, loc=0.0, scale=1.0, size=None):
        return self.rs.normal(loc, scale, size)

    def poisson(self, lam):
        return self.rs.poisson(lam)


def _volume(seed, shape=(24, 96, 112)):
    rng = np.random.default_rng(seed)
    from scipy import ndimage as ndi
    v = ndi.gaussian_filter(rng.random(shape).astype(np.float32), 2.0)
    v -= v.min()
    v[:, :30] *= 0.05                                         # a dark band: candidates there fail the brightness test
    return np.ascontiguousarray(v / v.max(), np.float32)


def test_sampler_picks_the_crops_of_the_restated_reference_on_the_same_draws():
    from iterseg_amd import train_io
    from oracle import train_io as otio
    vol = _volume(1)
    shape = (6, 32, 48)
    for seed, prop in ((0, 0.3), (5, 0.45)):
        ref = otio.get_image_chunks(vol, shape, 12, prop, np.random.RandomState(seed))
        got = train_io.get_image_chunks(vol, shape, 12, prop, rng=np.random.RandomState(seed))
        assert got['slices'] == ref['slices'] and got['n'] == ref['n'] == 12
        assert list(got['df']['z_start']) == [s[0] for s in ref['starts']]
        assert list(got['df']['x_start']) == [s[2] for s in ref['starts']]
        for a, b in zip(got['x'], ref['x']):
            assert np.array_equal(a, b)
        dev = train_io.get_image_chunks(vol, shape, 12, prop, rng=np.random.RandomState(seed), device=True)
        for a, b in zip(dev['x'], ref['x']):
            assert a.is_cuda and np.array_equal(a.cpu().numpy(), b)
    # some candidates were rejected (the test exercises the loop, not only its first draw)
    rs = np.random.RandomState(5)
    ref = otio.get_image_chunks(vol, shape, 12, 0.45, rs)
    assert rs.randint(0, 1 << 30) != np.random.RandomState(5).randint(0, 1 << 30, 36)[-1]


def test_global_numpy_state_is_the_default_like_upstream():
    from iterseg_amd import train_io
    from oracle import train_io as otio
    vol = _volume(2)
    np.random.seed(11)
    ref = otio.get_image_chunks(vol, (6, 32, 48), 5, 0.3)
    np.random.seed(11)
    got = train_io.get_image_chunks(vol, (6, 32, 48), 5, 0.3)
    assert got['slices'] == ref['slices']


def test_label_chunks_are_crops_of_the_whole_volume_targets():
    from iterseg_amd import train_io
    from iterseg_amd.labels import get_training_labels
    from scipy import ndimage as ndi
    gt = ndi.label(np.random.default_rng(3).random((24, 96, 112)) > 0.6)[0].astype(np.int32)
    vol = _volume(3)
    cd = train_io.get_image_chunks(vol, (6, 32, 48), 6, 0.3, rng=np.random.RandomState(2))
    chans = ('z-1', 'y-1', 'x-1', 'centreness-log')
    cd = train_io.get_labels_chunks(cd, gt, chans, (4, 1, 1))
    whole = get_training_labels(gt, chans, (4, 1, 1))
    assert list(cd['ys'].keys()) == ['y'] and cd['channels'] == {'y': chans}
    for s, y, g in zip(cd['slices'], cd['ys']['y'], cd['ground_truth']):
        assert y.shape == (4, 6, 32, 48)
        assert np.array_equal(y, whole[(slice(None),) + s])
        assert np.array_equal(g, gt[s])


@pytest.mark.parametrize("mirror,transpose", [(0, 0), (1, 0), (2, 0), (0, 1), (1, 1), (2, 1)])
def test_order_augmentation_is_the_same_permutation_for_every_array(mirror, transpose):
    from iterseg_amd import augment
    from oracle import train_io as otio
    rng = np.random.default_rng(7)
    img = rng.random((5, 24, 40)).astype(np.float32)
    labs = {'y': rng.random((4, 5, 24, 40)).astype(np.float32)}
    gt = rng.integers(0, 9, (5, 24, 40)).astype(np.int32)
    binom = [1, 0] + [1 if mirror else 0] + [1 if transpose else 0]          # augment, no noise, mirror?, transpose?
    ints = ([mirror - 1] if mirror else []) + ([0] if transpose else [])
    out_img, out_labs, out_gt = augment.augment_images(img, labs, gt, rng=ScriptedRNG(binom, ints))
    assert np.array_equal(out_gt, otio.apply_order(gt, mirror, transpose)) and out_gt.dtype == gt.dtype
    assert np.array_equal(out_labs['y'], otio.apply_order(labs['y'], mirror, transpose))
    assert out_img.shape == otio.apply_order(img, mirror, transpose).shape
    assert out_img.min() >= 0.0 and out_img.max() <= 1.0


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_intensity_augmentation_matches_the_restated_formula(mode):
    from iterseg_amd import augment
    from oracle import train_io as otio
    rng = np.random.default_rng(9)
    img = (rng.random((5, 24, 40)) * 3.0).astype(np.float32)
    # noise: binomial -> 1, randint picks the mode ('gaussian' = 0, 'speckle' = 3)
    binom = [1 if mode else 0]
    ints = [0 if mode == 1 else 3] if mode else []
    r = ScriptedRNG(binom, ints, seed=4)
    got = augment.augment_intensity(img, rng=r)
    # replay the continuous draws to know scale / shift / noise
    rs = np.random.RandomState(4)
    def cc(lo, hi, sig, loc):
        while True:
            o = rs.normal(loc, sig, 1)
            if lo <= o <= hi:
                return float(o[0])
    scale, shift = cc(0.8, 1.2, .05, 1.0), cc(-0.1, 0.1, 0.02, 0.0)
    noise = rs.normal(0.0, 0.001 ** 0.5, img.shape).astype(np.float32) if mode else None
    ref = otio.intensity(img, scale, shift, mode, noise)
    assert np.abs(got - ref).max() <= 2e-6


def test_no_augmentation_returns_the_inputs():
    from iterseg_amd import augment
    img = np.random.default_rng(1).random((4, 16, 16)).astype(np.float32)
    lab = np.ones((2, 4, 16, 16), np.float32)
    a, b = augment.augment_images(img, lab, rng=ScriptedRNG([0], []))
    assert a is img and b is lab


def test_augment_chunks_walks_the_chunk_dict_like_upstream():
    from iterseg_amd import train_io
    rng = np.random.default_rng(2)
    cd = {'x': [rng.random((4, 16, 24)).astype(np.float32) for _ in range(3)],
          'ys': {'y': [rng.random((4, 4, 16, 24)).astype(np.float32) for _ in range(3)]},
          'ground_truth': [rng.integers(0, 5, (4, 16, 24)).astype(np.int32) for _ in range(3)], 'n': 3}
    np.random.seed(3)
    out = train_io.augment_chunks(cd)
    assert out['n'] == 3 and len(out['x']) == 3
    for x, y, g in zip(out['x'], out['ys']['y'], out['ground_truth']):
        assert x.shape[-2:] == y.shape[-2:] == g.shape[-2:] and x.shape[0] == 4
