#!/usr/bin/env python3
"""Golden-vector generator, stage 5: the patch sampler and the augmentation (second half).

Run ONLY in the build container (needs /root/reference and the scikit-image 0.18.3 interpreter):

    /opt/conda/bin/python3.9 -W ignore tests/golden/gen_stage5_train_io.py

Imports the reference's ``train_io.py``, ``augment.py`` and ``labels.py`` unmodified (``zarr`` / ``zarr.core.Array``,
``torch`` and ``napari.types`` are empty stubs: the functions called here never touch them; the package object is synthetic
so that ``iterseg/__init__.py`` -- napari widgets -- is not executed) and records, under ``np.random.seed(k)`` (upstream
draws from the global ``numpy.random`` state; scikit-image 0.18.3's ``random_noise`` does too):

* ``get_image_chunks`` (train_io.py:224-275): starts, data-frame columns, the crops, the number of candidates drawn (runs
  with rejected candidates included) and the next draw of the global state after the call (pins the consumed draws);
* ``get_labels_chunks`` (train_io.py:291-320) on those slices;
* ``augment_order`` (augment.py:75-118) for every mirror x transpose case on image / 4-D labels / ground truth, and for
  ``used_axes=(-3, -2, -1)``;
* ``augment_intensity`` (augment.py:35-72) without noise and with every ``random_noise`` mode upstream can pick
  (gaussian, localvar, poisson, speckle), on an image with zeros (negative shift -> skimage's signed clip range) and on a
  strictly positive one;
* ``augment_images`` (augment.py:8-32) end to end for a list of seeds, and ``augment_chunks`` (train_io.py:331-341).

Writes tests/golden/train_io_sampler.npz and tests/golden/augment.npz (data only).
"""
import importlib.util
import os
import sys
import types

import numpy as np
from scipy import ndimage as ndi

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/src/iterseg"


def import_reference():
    zarr = types.ModuleType("zarr")
    zcore = types.ModuleType("zarr.core")

    class Array:                                             # isinstance(..., Array) is the only use (augment.py:47,121)
        pass
    zcore.Array = Array
    zarr.core = zcore
    sys.modules["zarr"], sys.modules["zarr.core"] = zarr, zcore
    sys.modules["torch"] = types.ModuleType("torch")
    nap = types.ModuleType("napari")
    ntypes = types.ModuleType("napari.types")
    ntypes.ImageData = ntypes.LabelsData = object
    nap.types = ntypes
    sys.modules["napari"], sys.modules["napari.types"] = nap, ntypes
    pkg = types.ModuleType("iterseg")
    pkg.__path__ = [REF]
    sys.modules["iterseg"] = pkg
    mods = {}
    for name in ("helpers", "augment", "labels", "train_io"):
        spec = importlib.util.spec_from_file_location(f"iterseg.{name}", os.path.join(REF, f"{name}.py"))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[f"iterseg.{name}"] = mod
        spec.loader.exec_module(mod)
        mods[name] = mod
    return mods


def sentinel():
    return int(np.random.randint(0, 1 << 30))


def volume(seed, shape):
    rng = np.random.default_rng(seed)
    v = ndi.gaussian_filter(rng.random(shape).astype(np.float32), 2.0)
    v -= v.min()
    v[:, :22] *= 0.05                                        # a dark band: candidates there fail the brightness test
    return np.ascontiguousarray(v / v.max(), np.float32)


def label_volume(seed, shape):
    rng = np.random.default_rng(seed)
    f = ndi.gaussian_filter(rng.random(shape), (1.0, 2.0, 2.0))
    lab, _ = ndi.label(f > np.quantile(f, 0.7))
    return lab.astype(np.int32)


class Counter:
    """Counts calls of np.random.randint made by the reference (three per candidate crop)."""

    def __init__(self):
        self.n = 0
        self.orig = np.random.randint

    def __enter__(self):
        def counted(*a, **k):
            self.n += 1
            return self.orig(*a, **k)
        np.random.randint = counted
        return self

    def __exit__(self, *a):
        np.random.randint = self.orig


def sampler(mods):
    tio = mods["train_io"]
    out = {}
    vol = volume(1, (20, 72, 88))
    gt = label_volume(2, (20, 72, 88))
    out["vol"], out["gt"] = vol, gt
    cases = [(0, (5, 16, 24), 8, 0.3), (5, (5, 16, 24), 8, 0.45), (9, (4, 20, 20), 6, 0.5)]
    out["cases"] = np.array([[s, n] + list(sh) for s, sh, n, _ in cases], np.int64)
    out["props"] = np.array([p for *_, p in cases], np.float64)
    for k, (seed, shape, n, prop) in enumerate(cases):
        np.random.seed(seed)
        with Counter() as c:
            cd = tio.get_image_chunks(vol, shape=shape, n=n, min_brightness_prop=prop, image_no=3)
        out[f"s{k}.next"] = np.int64(sentinel())
        out[f"s{k}.candidates"] = np.int64(c.n // 3)
        out[f"s{k}.starts"] = np.array([[s.start for s in sl] for sl in cd["slices"]], np.int64)
        out[f"s{k}.stops"] = np.array([[s.stop for s in sl] for sl in cd["slices"]], np.int64)
        for col in ("z_start", "y_start", "x_start"):
            out[f"s{k}.{col}"] = np.array(cd["df"][col], np.int64)
        out[f"s{k}.columns"] = np.array(list(cd["df"].columns))
        out[f"s{k}.id_suffix"] = np.array([i.split("_", 2)[2] for i in cd["ids"]])
        out[f"s{k}.x"] = np.stack(cd["x"])
        assert cd["n"] == n
        print("sampler", k, "candidates", c.n // 3, "for", n)
        if k == 0:                                           # train_io.py:291-320 on the same chunk dict
            chans = ('z-1', 'y-1', 'x-1', 'centreness-log')
            cd = tio.get_labels_chunks(cd, gt, channels=chans, scale=(4, 1, 1))
            out["l0.channels"] = np.array(chans)
            out["l0.keys"] = np.array(list(cd["ys"].keys()))
            out["l0.ys"] = np.stack(cd["ys"]["y"]).astype(np.float32)
            out["l0.ys_dtype"] = np.array(str(cd["ys"]["y"][0].dtype))
            out["l0.ground_truth"] = np.stack(cd["ground_truth"])
            # train_io.py:331-341: the whole chunk dict through augment_images, global state seeded
            np.random.seed(21)
            cd = tio.augment_chunks(cd)
            out["a0.next"] = np.int64(sentinel())
            for i in range(cd["n"]):
                out[f"a0.x{i}"] = np.asarray(cd["x"][i])
                out[f"a0.y{i}"] = np.asarray(cd["ys"]["y"][i]).astype(np.float32)
                out[f"a0.g{i}"] = np.asarray(cd["ground_truth"][i])
    np.savez_compressed(os.path.join(HERE, "train_io_sampler.npz"), **out)


def augmentation(mods):
    aug = mods["augment"]
    out = {}
    rng = np.random.default_rng(7)
    img = rng.random((5, 24, 40)).astype(np.float32) + np.float32(0.05)           # strictly positive
    img0 = img.copy()
    img0[rng.random(img.shape) < 0.2] = 0.0                                         # with zeros: shift < 0 -> signed range
    labs = rng.random((4, 5, 24, 40)).astype(np.float32)
    gt = rng.integers(0, 1 << 26, (5, 24, 40)).astype(np.int32)                     # ids beyond 2**24 (float32 would corrupt)
    cube = rng.random((6, 6, 6)).astype(np.float32)                                 # equal axes: any used_axes permutation is legal
    cube4 = rng.random((2, 6, 6, 6)).astype(np.float32)
    out.update(img=img, img0=img0, labs=labs, gt=gt, cube=cube, cube4=cube4)

    # --- augment_order: forced decisions (probabilities 0 / 1), seeds that pick either mirror axis
    k = 0
    for mp, tp in ((0, 0), (1, 0), (0, 1), (1, 1)):
        for seed in (0, 1, 2, 3):
            np.random.seed(seed)
            o = aug.augment_order([img, labs, gt], mirror_prob=float(mp), transpose_prob=float(tp))
            out[f"o{k}.cfg"] = np.array([seed, mp, tp], np.int64)
            out[f"o{k}.next"] = np.int64(sentinel())
            out[f"o{k}.img"], out[f"o{k}.labs"], out[f"o{k}.gt"] = (np.ascontiguousarray(a) for a in o)
            k += 1
    out["o.count"] = np.int64(k)
    k = 0
    for seed in range(8):                                     # used_axes = (-3, -2, -1): 3 mirror axes, 5 permutations
        np.random.seed(seed)
        o = aug.augment_order([cube, cube4], mirror_prob=0.7, transpose_prob=1.0, used_axes=(-3, -2, -1))
        out[f"u{k}.seed"] = np.int64(seed)
        out[f"u{k}.next"] = np.int64(sentinel())
        out[f"u{k}.cube"], out[f"u{k}.cube4"] = (np.ascontiguousarray(a) for a in o)
        k += 1
    out["u.count"] = np.int64(k)

    # --- augment_intensity: record the mode random_noise was called with (harness wrapper around the imported name)
    seen = {}
    real = aug.random_noise

    def spy(image, mode='gaussian', **kw):
        seen["mode"] = mode
        seen["signed"] = bool(image.min() < 0)
        return real(image, mode=mode, **kw)
    aug.random_noise = spy
    k = 0
    want = {(m, s) for m in ("none", "gaussian", "localvar", "poisson", "speckle") for s in (False, True)}
    for name, im in (("img", img), ("img0", img0)):
        for noise_prob in (0.0, 1.0):
            for seed in range(40):
                seen.clear()
                np.random.seed(seed)
                o = aug.augment_intensity(im, noise_prob=noise_prob)
                nxt = sentinel()
                mode = seen.get("mode", "none")
                signed = seen.get("signed", bool(o.min() < 0)) if mode != "none" else (name == "img0")
                if (mode, signed) not in want:
                    continue
                want.discard((mode, signed))
                out[f"i{k}.cfg"] = np.array([name, str(seed), str(noise_prob), mode, str(int(signed))])
                out[f"i{k}.next"] = np.int64(nxt)
                out[f"i{k}.out"] = np.asarray(o)
                print("intensity", k, name, seed, noise_prob, mode, "signed" if signed else "", o.dtype, float(o.min()), float(o.max()))
                k += 1
    out["i.count"] = np.int64(k)
    print("intensity cases not reached:", sorted(want))

    # --- augment_images end to end (default probabilities), dict labels + ground truth
    k = 0
    for seed in range(12):
        seen.clear()
        np.random.seed(seed)
        a, l, g = aug.augment_images(img0, {'y': labs, 'w': labs[:2]}, gt)
        out[f"e{k}.seed"] = np.int64(seed)
        out[f"e{k}.next"] = np.int64(sentinel())
        out[f"e{k}.mode"] = np.array(seen.get("mode", "none"))
        out[f"e{k}.img"], out[f"e{k}.y"], out[f"e{k}.w"], out[f"e{k}.gt"] = (np.ascontiguousarray(x) for x in (a, l['y'], l['w'], g))
        k += 1
    out["e.count"] = np.int64(k)
    aug.random_noise = real
    # continuous_choice (augment.py:170-185) on its own
    np.random.seed(3)
    out["cc"] = np.array([float(aug.continuous_choice(0.8, 1.2, .05, loc=1.)[0]) for _ in range(6)]
                         + [float(aug.continuous_choice(-0.01, 0.01, .05)[0]) for _ in range(6)])
    np.savez_compressed(os.path.join(HERE, "augment.npz"), **out)


if __name__ == "__main__":
    m = import_reference()
    sampler(m)
    augmentation(m)
