#!/usr/bin/env python3
"""Golden-vector generator, stage 3: training-target generation.

Run ONLY in the build container (needs /root/reference and the scikit-image 0.18.3 interpreter):

    /opt/conda/bin/python3.9 -W ignore tests/golden/gen_stage3_labels.py

Imports the reference's ``labels.py`` unmodified (torch / zarr / the package-relative helpers are stubbed: the
functions called here never touch them) and records ``get_training_labels`` (labels.py:23-68) for seeded label
volumes: nth affinities (labels.py:87-108), centreness (labels.py:142-205) and the semantic mask (labels.py:283-285);
second fixture (training_labels_ext.npz): centre offsets (labels.py:212-276) and centreness-log.
"""
import importlib.util
import os
import sys
import types

import numpy as np
from scipy import ndimage as ndi

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/src/iterseg"


def import_labels():
    for name in ("torch", "zarr"):
        sys.modules[name] = types.ModuleType(name)
    pkg = types.ModuleType("iterseg")
    pkg.__path__ = [REF]
    sys.modules["iterseg"] = pkg
    aug = types.ModuleType("iterseg.augment"); aug.augment_images = None
    hlp = types.ModuleType("iterseg.helpers")
    hlp.get_files = hlp.log_dir_or_None = hlp.write_log = None
    hlp.LINE = "-" * 10
    sys.modules["iterseg.augment"], sys.modules["iterseg.helpers"] = aug, hlp
    spec = importlib.util.spec_from_file_location("iterseg.labels", os.path.join(REF, "labels.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["iterseg.labels"] = mod
    spec.loader.exec_module(mod)
    return mod


def label_volume(shape, seed, density):
    """Seeded instance labels: thresholded smooth noise, 6-connected components (ids 1..n), a few ids removed so
    that the id range has gaps, one single-voxel object."""
    rng = np.random.default_rng(seed)
    f = ndi.gaussian_filter(rng.random(shape), (1.0, 2.0, 2.0))
    lab, n = ndi.label(f > np.quantile(f, 1.0 - density))
    lab = lab.astype(np.int64)
    lab[lab == 3] = 0
    lab[lab == 7] = 0
    lab[0, 0, 0] = n + 5                                   # isolated voxel, label beyond the range
    return lab


def main():
    labels = import_labels()
    out = {}
    cases = {"a": ((12, 40, 40), 1, 0.25, ('z-1', 'y-1', 'x-1', 'centreness'), (4, 1, 1)),
             "b": ((9, 33, 47), 2, 0.4, ('z-2', 'y-3', 'x-1', 'mask', 'centreness'), (2, 1, 0.5))}
    for tag, (shape, seed, dens, chans, scale) in cases.items():
        l = label_volume(shape, seed, dens)
        t = labels.get_training_labels(l, channels=chans, scale=scale)
        out[f"{tag}.labels"] = l.astype(np.int32)
        out[f"{tag}.targets"] = t.astype(np.float32) if t.dtype != np.float32 else t
        out[f"{tag}.targets_dtype"] = np.array(str(t.dtype))
        out[f"{tag}.channels"] = np.array(chans)
        out[f"{tag}.scale"] = np.array(scale, np.float64)
        print(tag, l.shape, int(l.max()), t.shape, t.dtype, [float(t[i].sum()) for i in range(t.shape[0])])
    np.savez_compressed(os.path.join(HERE, "training_labels.npz"), **out)
    # second fixture: the per-axis centre offsets (labels.py:212-276) and the log variant of the centreness
    # (labels.py:178-205 with log=True); small volumes, the reference loops over voxels in Python
    ext = {}
    cases = {"c": ((8, 32, 36), 3, 0.3, ('offsets-z', 'offsets-y', 'offsets-x', 'centreness-log'), (4, 1, 1)),
             "d": ((6, 25, 31), 5, 0.45, ('offsets-x', 'centreness-log', 'centreness', 'offsets-z'), (2, 1, 0.5)),
             # '<axis>-<n>-smooth' (labels.py:64-65,312-325): the only channel kind the '-smooth' suffix parses for
             "e": ((5, 30, 28), 7, 0.35, ('z-1-smooth', 'y-2-smooth', 'x-1'), (4, 1, 1)),
             "f": ((7, 36, 30), 9, 0.3, ('centroid-gauss', 'mask'), (4, 1, 1))}      # labels.py:293-309
    for tag, (shape, seed, dens, chans, scale) in cases.items():
        l = label_volume(shape, seed, dens)
        t = labels.get_training_labels(l, channels=chans, scale=scale)
        ext[f"{tag}.labels"] = l.astype(np.int32)
        ext[f"{tag}.targets"] = t.astype(np.float32)
        ext[f"{tag}.targets_dtype"] = np.array(str(t.dtype))
        ext[f"{tag}.channels"] = np.array(chans)
        ext[f"{tag}.scale"] = np.array(scale, np.float64)
        print(tag, l.shape, int(l.max()), t.shape, t.dtype, [float(t[i].sum()) for i in range(t.shape[0])])
    np.savez_compressed(os.path.join(HERE, "training_labels_ext.npz"), **ext)


if __name__ == "__main__":
    main()
