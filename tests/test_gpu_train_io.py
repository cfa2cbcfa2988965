"""Patch sampler and augmentation (src/iterseg/train_io.py:224-341, augment.py:8-118) on the GPU against the REFERENCE's own
output: tests/golden/train_io_sampler.npz / augment.npz were produced by ``iterseg.train_io`` / ``iterseg.augment`` imported
unmodified under ``np.random.seed(k)`` (tests/golden/gen_stage5_train_io.py).  The product draws from the same global state
with upstream's calls, so the same seed must give the same crops, permutations and -- to float32 -- intensities, and leave
the generator where upstream leaves it (``next``)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 2e-6          # float32 intensities here, float64 upstream


@pytest.fixture(scope="module")
def samp():
    return np.load(os.path.join(GOLD, "train_io_sampler.npz"))


@pytest.fixture(scope="module")
def aug():
    return np.load(os.path.join(GOLD, "augment.npz"))


def sentinel():
    return int(np.random.randint(0, 1 << 30))


def test_sampler_picks_the_reference_crops_on_the_same_seed(samp):
    from iterseg_amd import train_io
    vol = samp["vol"]
    for k, (row, prop) in enumerate(zip(samp["cases"], samp["props"])):
        seed, n, shape = int(row[0]), int(row[1]), tuple(int(v) for v in row[2:])
        for device in (False, True):
            np.random.seed(seed)
            cd = train_io.get_image_chunks(vol, shape, n, float(prop), image_no=3, device=device)
            assert sentinel() == int(samp[f"s{k}.next"])              # same number of candidates drawn (rejections included)
            assert np.array_equal([[s.start for s in sl] for sl in cd["slices"]], samp[f"s{k}.starts"])
            assert np.array_equal([[s.stop for s in sl] for sl in cd["slices"]], samp[f"s{k}.stops"])
            for col in ("z_start", "y_start", "x_start"):
                assert np.array_equal(np.asarray(cd["df"][col]), samp[f"s{k}.{col}"])
            assert list(cd["df"].columns) == list(samp[f"s{k}.columns"])
            assert [i.split("_", 2)[2] for i in cd["ids"]] == list(samp[f"s{k}.id_suffix"])
            xs = [x.cpu().numpy() if device else np.asarray(x) for x in cd["x"]]
            assert all(x.is_cuda for x in cd["x"]) if device else True
            assert np.array_equal(np.stack(xs), samp[f"s{k}.x"]) and cd["n"] == n
    # an explicit generator draws the same crops as the seeded global state
    seed, n, shape = int(samp["cases"][1][0]), int(samp["cases"][1][1]), tuple(int(v) for v in samp["cases"][1][2:])
    cd = train_io.get_image_chunks(vol, shape, n, float(samp["props"][1]), rng=np.random.RandomState(seed))
    assert np.array_equal([[s.start for s in sl] for sl in cd["slices"]], samp["s1.starts"])


def test_label_chunks_and_augment_chunks_equal_the_reference(samp):
    from iterseg_amd import train_io
    vol, gt = samp["vol"], samp["gt"]
    seed, n, shape = int(samp["cases"][0][0]), int(samp["cases"][0][1]), tuple(int(v) for v in samp["cases"][0][2:])
    np.random.seed(seed)
    cd = train_io.get_image_chunks(vol, shape, n, float(samp["props"][0]))
    chans = tuple(str(c) for c in samp["l0.channels"])
    cd = train_io.get_labels_chunks(cd, gt, chans, (4, 1, 1))
    assert list(cd["ys"].keys()) == list(samp["l0.keys"]) and cd["channels"] == {'y': chans}
    assert np.array_equal(np.stack(cd["ground_truth"]), samp["l0.ground_truth"])
    ys = np.stack(cd["ys"]["y"])
    assert ys.shape == samp["l0.ys"].shape
    assert np.array_equal(ys[:, :3], samp["l0.ys"][:, :3])                     # affinities: exact
    assert np.abs(ys[:, 3] - samp["l0.ys"][:, 3]).max() <= 2e-7               # centreness-log
    # train_io.py:331-341 over the reference's own chunks, global state seeded like the generator
    cd["ys"]["y"] = [a for a in samp["l0.ys"]]
    np.random.seed(21)
    cd = train_io.augment_chunks(cd)
    assert sentinel() == int(samp["a0.next"])
    for i in range(n):
        assert cd["x"][i].shape == samp[f"a0.x{i}"].shape
        assert np.abs(np.asarray(cd["x"][i], np.float64) - samp[f"a0.x{i}"]).max() <= TOL
        assert np.array_equal(cd["ys"]["y"][i], samp[f"a0.y{i}"])
        assert np.array_equal(cd["ground_truth"][i], samp[f"a0.g{i}"]) and cd["ground_truth"][i].dtype == gt.dtype


def test_order_augmentation_every_mirror_and_transpose_case(aug):
    from iterseg_amd import augment
    import torch
    img, labs, gt = aug["img"], aug["labs"], aug["gt"]
    assert int(gt.max()) > 1 << 24                            # ids a float32 round trip would corrupt
    for k in range(int(aug["o.count"])):
        seed, mp, tp = (int(v) for v in aug[f"o{k}.cfg"])
        np.random.seed(seed)
        o = augment.augment_order([img, labs, gt], mirror_prob=float(mp), transpose_prob=float(tp))
        assert sentinel() == int(aug[f"o{k}.next"])
        for got, key in zip(o, ("img", "labs", "gt")):
            want = aug[f"o{k}.{key}"]
            assert got.dtype == want.dtype and np.array_equal(got, want), (k, key)
    # device tensors in -> device tensors out, int64 ground truth through two word planes
    np.random.seed(int(aug["o5.cfg"][0]))
    o = augment.augment_order([torch.from_numpy(img).cuda(), torch.from_numpy(gt.astype(np.int64) << 20).cuda()],
                              mirror_prob=float(aug["o5.cfg"][1]), transpose_prob=float(aug["o5.cfg"][2]))
    assert o[0].is_cuda and np.array_equal(o[0].cpu().numpy(), aug["o5.img"])
    assert o[1].dtype == torch.int64 and np.array_equal(o[1].cpu().numpy(), aug["o5.gt"].astype(np.int64) << 20)
    for k in range(int(aug["u.count"])):                      # used_axes = (-3, -2, -1): z takes part
        np.random.seed(int(aug[f"u{k}.seed"]))
        o = augment.augment_order([aug["cube"], aug["cube4"]], mirror_prob=0.7, transpose_prob=1.0, used_axes=(-3, -2, -1))
        assert sentinel() == int(aug[f"u{k}.next"])
        assert np.array_equal(o[0], aug[f"u{k}.cube"]) and np.array_equal(o[1], aug[f"u{k}.cube4"])


def test_intensity_augmentation_every_noise_mode_and_clip_range(aug):
    from iterseg_amd import augment
    modes = set()
    for k in range(int(aug["i.count"])):
        name, seed, noise_prob, mode, signed = aug[f"i{k}.cfg"]
        np.random.seed(int(seed))
        o = augment.augment_intensity(aug[str(name)], noise_prob=float(noise_prob))
        assert sentinel() == int(aug[f"i{k}.next"]), (k, mode)
        want = aug[f"i{k}.out"]
        assert o.shape == want.shape and o.dtype == np.float32
        assert np.abs(o.astype(np.float64) - want).max() <= TOL, (k, str(mode), float(np.abs(o - want).max()))
        modes.add((str(mode), int(signed)))
    assert len(modes) == 10                                   # none / gaussian / localvar / poisson / speckle x unsigned / signed


def test_augment_images_end_to_end_on_the_reference_seeds(aug):
    from iterseg_amd import augment
    img0, labs, gt = aug["img0"], aug["labs"], aug["gt"]
    for k in range(int(aug["e.count"])):
        np.random.seed(int(aug[f"e{k}.seed"]))
        a, l, g = augment.augment_images(img0, {'y': labs, 'w': labs[:2]}, gt)
        assert sentinel() == int(aug[f"e{k}.next"]), k
        assert np.abs(np.asarray(a, np.float64) - aug[f"e{k}.img"]).max() <= TOL, (k, str(aug[f"e{k}.mode"]))
        assert list(l.keys()) == ['y', 'w']
        assert np.array_equal(l['y'], aug[f"e{k}.y"]) and np.array_equal(l['w'], aug[f"e{k}.w"])
        assert np.array_equal(g, aug[f"e{k}.gt"]) and g.dtype == gt.dtype
    np.random.seed(3)
    cc = [float(augment.continuous_choice(0.8, 1.2, .05, loc=1.)[0]) for _ in range(6)] \
        + [float(augment.continuous_choice(-0.01, 0.01, .05)[0]) for _ in range(6)]
    assert np.array_equal(cc, aug["cc"])


def test_no_augmentation_returns_the_inputs():
    from iterseg_amd import augment

    class Never:
        def binomial(self, n, p):
            return 0
    img = np.random.default_rng(1).random((4, 16, 16)).astype(np.float32)
    lab = np.ones((2, 4, 16, 16), np.float32)
    a, b = augment.augment_images(img, lab, rng=Never())
    assert a is img and b is lab
