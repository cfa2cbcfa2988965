"""CPU restatement of the patch sampler / augmentation (oracle/train_io.py) against the reference's own output:
tests/golden/train_io_sampler.npz and augment.npz come from ``iterseg.train_io`` / ``iterseg.augment`` imported unmodified
under ``np.random.seed(k)`` (tests/golden/gen_stage5_train_io.py).  ``next`` = the draw the global state gives after the
call: equal sentinels mean the restatement consumed exactly the reference's draws."""
import os

import numpy as np
import pytest

from oracle import train_io as otio

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def samp():
    return np.load(os.path.join(GOLD, "train_io_sampler.npz"))


@pytest.fixture(scope="module")
def aug():
    return np.load(os.path.join(GOLD, "augment.npz"))


def sentinel():
    return int(np.random.randint(0, 1 << 30))


def test_sampler_draws_crops_and_columns(samp):
    vol = samp["vol"]
    rejected = 0
    for k, (row, prop) in enumerate(zip(samp["cases"], samp["props"])):
        seed, n, shape = int(row[0]), int(row[1]), tuple(int(v) for v in row[2:])
        np.random.seed(seed)
        cd = otio.get_image_chunks(vol, shape, n, float(prop), image_no=3)
        assert sentinel() == int(samp[f"s{k}.next"])
        assert np.array_equal([[s.start for s in sl] for sl in cd["slices"]], samp[f"s{k}.starts"])
        assert np.array_equal([[s.stop for s in sl] for sl in cd["slices"]], samp[f"s{k}.stops"])
        for col in ("z_start", "y_start", "x_start"):
            assert np.array_equal(cd["df"][col], samp[f"s{k}.{col}"])
        assert list(cd["df"].keys()) == list(samp[f"s{k}.columns"])
        assert cd["ids"] == list(samp[f"s{k}.id_suffix"])
        assert np.array_equal(np.stack(cd["x"]), samp[f"s{k}.x"])
        rejected += int(samp[f"s{k}.candidates"]) - n
    assert rejected > 0                                       # the fixture exercises the rejection loop


def test_label_chunks_and_augment_chunks(samp):
    vol, gt = samp["vol"], samp["gt"]
    seed, n, shape = int(samp["cases"][0][0]), int(samp["cases"][0][1]), tuple(int(v) for v in samp["cases"][0][2:])
    np.random.seed(seed)
    cd = otio.get_image_chunks(vol, shape, n, float(samp["props"][0]))
    cd = otio.get_labels_chunks(cd, gt, tuple(samp["l0.channels"]), (4, 1, 1))
    assert list(cd["ys"].keys()) == list(samp["l0.keys"])
    assert np.array_equal(np.stack(cd["ground_truth"]), samp["l0.ground_truth"])
    ys = np.stack(cd["ys"]["y"]).astype(np.float32)
    assert np.array_equal(ys[:, :3], samp["l0.ys"][:, :3])                     # affinities: exact
    assert np.abs(ys[:, 3] - samp["l0.ys"][:, 3]).max() <= 2e-7               # centreness-log (float64 distances -> f32)
    for y in cd["ys"]["y"]:                                                   # the fixture's own crops from here on
        pass
    cd["ys"]["y"] = [a for a in samp["l0.ys"]]
    np.random.seed(21)
    cd = otio.augment_chunks(cd)
    assert sentinel() == int(samp["a0.next"])
    for i in range(n):
        assert np.abs(np.asarray(cd["x"][i], np.float64) - samp[f"a0.x{i}"]).max() <= 1e-12
        assert np.array_equal(cd["ys"]["y"][i], samp[f"a0.y{i}"])
        assert np.array_equal(cd["ground_truth"][i], samp[f"a0.g{i}"])


def test_order_augmentation_every_case(aug):
    img, labs, gt = aug["img"], aug["labs"], aug["gt"]
    seen = set()
    for k in range(int(aug["o.count"])):
        seed, mp, tp = (int(v) for v in aug[f"o{k}.cfg"])
        np.random.seed(seed)
        o = otio.augment_order([img, labs, gt], mirror_prob=float(mp), transpose_prob=float(tp))
        assert sentinel() == int(aug[f"o{k}.next"])
        for got, key in zip(o, ("img", "labs", "gt")):
            want = aug[f"o{k}.{key}"]
            assert got.dtype == want.dtype and np.array_equal(got, want)
        seen.add((np.array_equal(o[0], img[:, ::-1]) or np.array_equal(o[0], np.swapaxes(img[:, ::-1], 1, 2)),
                  np.array_equal(o[0], img[:, :, ::-1]) or np.array_equal(o[0], np.swapaxes(img[:, :, ::-1], 1, 2)), tp))
    assert {(True, False, 0), (False, True, 0), (True, False, 1), (False, True, 1), (False, False, 0), (False, False, 1)} <= seen
    for k in range(int(aug["u.count"])):                      # used_axes = (-3, -2, -1)
        np.random.seed(int(aug[f"u{k}.seed"]))
        o = otio.augment_order([aug["cube"], aug["cube4"]], mirror_prob=0.7, transpose_prob=1.0, used_axes=(-3, -2, -1))
        assert sentinel() == int(aug[f"u{k}.next"])
        assert np.array_equal(o[0], aug[f"u{k}.cube"]) and np.array_equal(o[1], aug[f"u{k}.cube4"])


def test_intensity_augmentation_every_noise_mode(aug):
    modes = set()
    for k in range(int(aug["i.count"])):
        name, seed, noise_prob, mode, signed = aug[f"i{k}.cfg"]
        info = {}
        np.random.seed(int(seed))
        o = otio.augment_intensity(aug[str(name)], noise_prob=float(noise_prob), info=info)
        assert sentinel() == int(aug[f"i{k}.next"]), (k, mode)
        assert info["mode"] == str(mode)
        want = aug[f"i{k}.out"]
        assert o.dtype == want.dtype == np.float64
        assert np.abs(o - want).max() <= 1e-12, (k, mode)
        if int(signed) and str(mode) != "none":
            assert want.min() < 0                             # skimage's signed clip range was in force
        modes.add((str(mode), int(signed)))
    assert len(modes) == 10


def test_augment_images_end_to_end(aug):
    img0, labs, gt = aug["img0"], aug["labs"], aug["gt"]
    modes = set()
    for k in range(int(aug["e.count"])):
        info = {}
        np.random.seed(int(aug[f"e{k}.seed"]))
        a, l, g = otio.augment_images(img0, {'y': labs, 'w': labs[:2]}, gt, info=info)
        assert sentinel() == int(aug[f"e{k}.next"])
        assert info.get("mode", "none") == str(aug[f"e{k}.mode"])
        assert np.abs(np.asarray(a, np.float64) - aug[f"e{k}.img"]).max() <= 1e-12
        assert np.array_equal(l['y'], aug[f"e{k}.y"]) and np.array_equal(l['w'], aug[f"e{k}.w"])
        assert np.array_equal(g, aug[f"e{k}.gt"]) and g.dtype == gt.dtype
        modes.add(info.get("mode", "none"))
    assert len(modes) >= 2
    np.random.seed(3)
    cc = [float(otio.continuous_choice(0.8, 1.2, .05, loc=1.)[0]) for _ in range(6)] \
        + [float(otio.continuous_choice(-0.01, 0.01, .05)[0]) for _ in range(6)]
    assert np.array_equal(cc, aug["cc"])
