"""TEST INFRASTRUCTURE -- CPU restatement of the reference's patch sampler and augmentation; never imported by the product.

* ``get_image_chunks`` follows src/iterseg/train_io.py:224-275 line by line (numpy only; the ``datetime`` part of the ids
  is omitted), ``get_labels_chunks`` :291-320 (over ``oracle.labels.get_training_labels``), ``augment_chunks`` :331-341.
* ``augment_images`` / ``augment_intensity`` / ``augment_order`` / ``continuous_choice`` follow src/iterseg/augment.py:8-118,
  170-185, any ``used_axes``.
* ``random_noise`` restates the part of scikit-image 0.18.3's ``skimage.util.random_noise`` (third-party; the reference's
  environment leaves it unpinned) that ``augment_intensity`` can reach: modes gaussian / localvar / poisson /
  speckle, ``clip=True``, drawing from the GLOBAL ``numpy.random`` state like that version does (signed input -- any
  value below 0 -- is clipped to [-1, 1] instead of [0, 1], and poisson goes through ``(image + 1) / (max + 1)``).

PINNED: tests/golden/train_io_sampler.npz and tests/golden/augment.npz were produced by the reference's own modules,
imported unmodified under ``np.random.seed(k)`` (tests/golden/gen_stage5_train_io.py, scikit-image 0.18.3 interpreter);
tests/test_oracle_train_io.py checks every function here against them (crops, permutations and the consumed draws exact,
float64 intensities to 1e-12).
"""
from itertools import permutations

import numpy as np


def get_image_chunks(image, shape=(10, 256, 256), n=25, min_brightness_prop=0.3, image_no=0, rng=np.random):
    im = np.array(image)                                        # train_io.py:231
    assert len(im.shape) == len(shape)
    xs, slices, starts, ids = [], [], [], []
    df = {'z_start': [], 'y_start': [], 'x_start': []}
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
            for key, r in zip(('z_start', 'y_start', 'x_start'), dim_randints):    # :254-255, 278-284
                df[key].append(r)
            xs.append(x)
            ids.append(f'img-{image_no}_chunk-{i}')             # :261 without the time stamp
            i += 1
    df['data_ids'] = ids                                        # :265
    return {'x': xs, 'slices': slices, 'starts': starts, 'ids': ids, 'df': df, 'n': len(xs)}


def get_labels_chunks(chunk_dict, ground_truth, channels=('z-1', 'y-1', 'x-1', 'centreness-log'), scale=(4, 1, 1)):
    from . import labels as olabels
    if not isinstance(channels, dict):                          # train_io.py:297-298
        channels = {'y': channels}
    chunk_dict['channels'] = channels
    labels = {key: olabels.get_training_labels(ground_truth, channels[key], scale) for key in channels}
    chunk_dict['ys'] = {key: [] for key in labels}
    chunk_dict['ground_truth'] = []
    for s_ in chunk_dict['slices']:                             # :309-319
        chunk_dict['ground_truth'].append(ground_truth[s_])
        for key in labels:
            chunk_dict['ys'][key].append(labels[key][(slice(None),) + tuple(s_)])
    return chunk_dict


def augment_chunks(chunk_dict, rng=np.random):
    n = chunk_dict['n']                                         # train_io.py:331-341
    for i in range(n):
        labels_dict = {key: chunk_dict['ys'][key][i] for key in chunk_dict['ys']}
        image, labels_dict, gt = augment_images(chunk_dict['x'][i], labels_dict, chunk_dict['ground_truth'][i], rng=rng)
        chunk_dict['x'][i] = image
        for key in labels_dict:
            chunk_dict['ys'][key][i] = labels_dict[key]
        chunk_dict['ground_truth'][i] = gt
    return chunk_dict


def continuous_choice(min_, max_, sigma, loc=0., size=1, rng=np.random):
    while True:                                                 # augment.py:170-185
        out = rng.normal(loc=loc, scale=sigma, size=size)
        val = out if size == 1 else out.mean()
        if val >= min_ and val <= max_:
            return out


def random_noise(image, mode, rng=np.random, **kwargs):
    """scikit-image 0.18.3 ``skimage/util/noise.py:random_noise`` for the four modes and the defaults upstream uses."""
    low_clip = -1. if image.min() < 0 else 0.
    image = np.asarray(image, np.float64) if image.dtype != np.float32 else image      # img_as_float keeps float32 / float64
    if mode in ('gaussian', 'speckle'):
        noise = rng.normal(kwargs.get('mean', 0.), kwargs.get('var', 0.01) ** 0.5, image.shape)
        out = image + noise if mode == 'gaussian' else image + image * noise
    elif mode == 'localvar':
        out = image + rng.normal(0, (np.zeros_like(image) + 0.01) ** 0.5)
    elif mode == 'poisson':
        vals = 2 ** np.ceil(np.log2(len(np.unique(image))))
        if low_clip == -1.:
            old_max = image.max()
            image = (image + 1.) / (old_max + 1.)
        out = rng.poisson(image * vals) / float(vals)
        if low_clip == -1.:
            out = out * (old_max + 1.) - 1.
    else:
        raise ValueError(mode)
    return np.clip(out, low_clip, 1.0)


NOISE_OPTIONS = ['gaussian', 'localvar', 'poisson', 'speckle', 'gaussian', 'speckle']     # augment.py:58


def augment_intensity(image, min_shift=-0.1, max_shift=0.1, min_scale=0.8, max_scale=1.2, shift_sigma=0.02, scale_sigma=.05,
                      noise_prob=0.3, rng=np.random, info=None):
    out = image.copy() / image.max()                            # augment.py:49
    scale = continuous_choice(min_scale, max_scale, scale_sigma, loc=1., rng=rng)
    shift = continuous_choice(min_shift, max_shift, shift_sigma, rng=rng)
    out = (out * scale) + shift                                 # :54 -- scale / shift are float64 arrays of one element
    mode = 'none'
    if rng.binomial(1, noise_prob):                             # :56
        mode = NOISE_OPTIONS[rng.randint(len(NOISE_OPTIONS))]
        kwargs = {'var': 0.001} if mode in ('gaussian', 'speckle') else {}
        out = random_noise(out, mode, rng=rng, **kwargs)
    else:
        out = np.where(out < 0, 0, out)                         # :70-71
        out = np.where(out > 1, 1, out)
    if info is not None:
        info['mode'] = mode
    return out


def order_plan(used_axes=(-2, -1), mirror_prob=0.2, transpose_prob=0.2, rng=np.random):
    """The decisions of augment.py:84-104 in upstream's draw order: (mirror axis or None, permutation p or None)."""
    axis = p = None
    if rng.binomial(1, mirror_prob):
        axis = used_axes[rng.randint(0, len(used_axes))]
    if rng.binomial(1, transpose_prob):
        ps = [q for q in permutations(used_axes) if q != tuple(used_axes)]
        p = ps[rng.randint(0, len(ps))]
    return axis, p


def apply_order(a, axis, p, used_axes=(-2, -1)):
    """augment.py:89-117 for one array: mirror along ``axis``, then ``axes[used_axes[i]] = p[i]`` and np.transpose."""
    a = np.asarray(a).copy()
    if axis is not None:
        s_ = [slice(None)] * a.ndim
        s_[axis] = slice(None, None, -1)
        a = a[tuple(s_)]
    if p is not None:
        axes = list(range(a.ndim))
        for i, ax in enumerate(used_axes):
            na = p[i]
            axes[ax] = na + len(axes) if na < 0 else na
        a = np.transpose(a, axes)
    return a


def augment_order(images, mirror_prob=0.2, transpose_prob=0.2, used_axes=(-2, -1), rng=np.random):
    axis, p = order_plan(used_axes, mirror_prob, transpose_prob, rng)
    return [apply_order(img, axis, p, used_axes) for img in images]


def augment_images(image, labels, ground_truth=None, augment_prob=0.9, rng=np.random, info=None):
    augment = rng.binomial(1, augment_prob)                     # augment.py:9
    if augment:
        image = augment_intensity(image, rng=rng, info=info)
    imgs = [image]
    if isinstance(labels, dict):
        imgs += [labels[key] for key in labels]
    else:
        imgs.append(labels)
    if ground_truth is not None:
        imgs.append(ground_truth)
    if augment:
        imgs = augment_order(imgs, rng=rng)
    result = [imgs[0]]
    if isinstance(labels, dict):
        result.append({key: imgs[i + 1] for i, key in enumerate(labels)})
    else:
        result.append(imgs[1])
    if ground_truth is not None:
        result.append(imgs[-1])
    return tuple(result)
