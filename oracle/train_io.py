"""TEST INFRASTRUCTURE -- CPU restatement of the reference's patch sampler and augmentation; never imported by the product.

* ``get_image_chunks`` follows src/iterseg/train_io.py:224-275 line by line (numpy only; the ``datetime`` ids are omitted).
* ``augment_order_plan`` / ``apply_order`` follow src/iterseg/augment.py:76-118; ``intensity`` follows :36-73 up to the noise
  (skimage's ``random_noise`` is a third-party dependency, scikit-image pinned by the reference's environment; its
  "gaussian" mode is ``clip(image + N(0, var))`` and "speckle" ``clip(image + image * N(0, var))`` with clip range [0, 1]
  for non-negative input -- restated here with the noise field given by the caller).

PARITY UNPINNED: ``train_io`` imports zarr / dask / napari at module level and none of them is in this image, so the
reference itself cannot be run here and its tests hold no fixture for these functions; this restatement is what the GPU
path is compared with (same draws -> same crops; augmentation: properties).
"""
import numpy as np


def get_image_chunks(image, shape=(10, 256, 256), n=25, min_brightness_prop=0.3, rng=np.random):
    im = np.array(image)                                        # train_io.py:231
    assert len(im.shape) == len(shape)
    xs, slices, starts = [], [], []
    i = 0
    while i < n:                                                # :240
        dim_randints = []
        for j, dim in enumerate(shape):
            max_ = im.shape[j] - dim - 1                        # :243
            dim_randints.append(rng.randint(0, max_))           # :244
        s_ = tuple(slice(dim_randints[j], dim_randints[j] + shape[j]) for j in range(len(shape)))
        x = im[s_]
        mbrightprop = x.mean() / x.max()                        # :249
        if mbrightprop > min_brightness_prop:                   # :250
            slices.append(s_)
            starts.append(tuple(dim_randints))
            xs.append(x)
            i += 1
    return {'x': xs, 'slices': slices, 'starts': starts, 'n': len(xs)}


def apply_order(a, mirror_axis, transpose):
    """augment.py:84-117 for used_axes = (-2, -1): mirror along y (1) or x (2), then swap the last two axes."""
    a = np.asarray(a)
    if mirror_axis == 1:
        a = a[..., ::-1, :]
    elif mirror_axis == 2:
        a = a[..., :, ::-1]
    if transpose:
        a = np.swapaxes(a, -2, -1)
    return a


def intensity(image, scale, shift, noise_mode=0, noise=None):
    """augment.py:47-72: image / image.max() * scale + shift, then the noise (restated random_noise) or the clip."""
    out = image.astype(np.float32) / np.float32(image.max())
    out = out * np.float32(scale) + np.float32(shift)
    if noise_mode == 1:
        out = out + noise
    elif noise_mode == 2:
        out = out + out * noise
    return np.clip(out, 0.0, 1.0)
