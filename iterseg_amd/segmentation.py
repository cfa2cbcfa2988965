"""Mirror of the affinity-U-Net-watershed operator of ``iterseg.segmentation``.

Reference: src/iterseg/segmentation.py -- ``affinity_unet_watershed`` (:24-73),
``affinity_watershed_prep_config`` (:80-135), ``a_w_output_volume`` (:138-140),
``affinity_watershed_for_chunks`` (:147-195), ``segmentation_wrapper`` (:700-830),
``segmentation_loop`` (:833-882), ``segment_single_volume`` (:885-900),
``remove_sum_zero_slices`` (:903-916) and the ``segmenters`` registry (:924-930).

Same names, signatures, in-place output convention and exceptions; the processing function runs
the whole frame (chunked U-Net + watershed) on the MI355X through ``libitersegx.so``.  napari and
zarr are optional here: any object with ``.data`` (and ``.scale`` / ``.translate`` if a viewer is
given) works as the image layer, and without zarr the label store is a numpy array.
"""
from __future__ import annotations

import collections
import ctypes as C
import json
import os
import pathlib
import weakref

import numpy as np
import torch

from . import _lib
from . import unet as unet_mod
from . import watershed as ws
from .predict import load_unet, predict_chunk_feature_map, process_chunks  # noqa: F401


# ------------------------
# Affinity U-net Watershed
# ------------------------

def affinity_unet_watershed(napari_viewer, input_volume_layer, save_dir=None, name='my-segmentation',
                            unet_or_config_file=None, layer_reference=None, chunk_size=(10, 256, 256),
                            margin=(1, 64, 64), debug=False):
    """Operator entry point, same arguments as the reference (segmentation.py:24-73)."""
    return segmentation_wrapper(affinity_watershed_for_chunks, affinity_watershed_prep_config, napari_viewer,
                                input_volume_layer, save_dir, name, unet_or_config_file, layer_reference,
                                chunk_size, margin, debug)


def affinity_watershed_prep_config(input_volume_layer, unet_or_config_file, reference_layer):
    """Builds ``{'unet': UNet, 'output_volume': float32 (5, Z, Y, X)}`` (segmentation.py:80-135).

    Accepted forms, as in the reference: ``None`` (default network -- not distributed upstream, so
    this raises), a ``.pt``/``.pth`` path (str or PurePath).  The reference's ``.json`` branch never
    assigns ``unet`` (UnboundLocalError, segmentation.py:98-107); here it is read as the documented
    intent: ``{"unet": <path> | "default" | "labels layer"}``.
    """
    unet = None
    if isinstance(unet_or_config_file, pathlib.PurePath):
        unet_or_config_file = str(unet_or_config_file)
    if isinstance(unet_or_config_file, unet_mod.UNet):          # already-loaded network (device resident)
        return {'unet': unet_or_config_file,
                'output_volume': a_w_output_volume(input_volume_layer.data, 5)}
    if isinstance(unet_or_config_file, str):
        if unet_or_config_file.endswith('.json'):
            config = read_config_json(unet_or_config_file)
            unet = config.get('unet')
            if unet == 'labels layer':
                unet = reference_layer.metadata['unet']
            if unet == 'default':
                unet = None
        elif unet_or_config_file.endswith('.pt') or unet_or_config_file.endswith('.pth'):
            unet = unet_or_config_file
    if unet is not None:
        if isinstance(unet, str):
            m = f'There was not file at the provided location: {unet}\nMake sure a pytorch unet lives here...'
            assert os.path.exists(unet), m
            assert unet.endswith('.pt') or unet.endswith('.pth')
        else:
            raise ValueError('Please provide a valid path to a pytorch unet - must end with .pt or .pth')
    affinities_extent = 1
    num_pred_channels = 3 * affinities_extent + 2
    output_volume = a_w_output_volume(input_volume_layer.data, num_pred_channels)
    unet = load_unet(unet)
    return {'unet': unet, 'output_volume': output_volume}


# Feature-map buffers handed out by a_w_output_volume that nothing has written yet.  The reference's processing function
# fills `output_volume` chunk by chunk and zeroes it when the frame is done (segmentation.py:183-195); here the feature map
# lives in HBM, the host array is never written, and re-zeroing 2.7 GB of untouched host memory per 512^3 frame would only
# commit its pages.  The arrays are handed out read-only, so "still all zero" holds as long as they are registered here; the
# generic branch (a foreign model object) makes its array writable again and drops it from the registry.
_PRISTINE_ZEROS = {}


def a_w_output_volume(data, num_pred_channels, **kwargs):
    out = np.zeros((num_pred_channels,) + tuple(data.shape[-3:]), dtype=np.float32)
    out.flags.writeable = False
    key = id(out)
    _PRISTINE_ZEROS[key] = weakref.ref(out, lambda _r, k=key: _PRISTINE_ZEROS.pop(k, None))
    return out


def _is_pristine_zeros(a):
    r = _PRISTINE_ZEROS.get(id(a))
    return r is not None and r() is a and not a.flags.writeable


def _make_writable(a):
    if _PRISTINE_ZEROS.pop(id(a), None) is not None:
        a.flags.writeable = True


def affinity_watershed_for_chunks(input_volume, current_output, chunk_size, margin, unet=None,
                                  output_volume=None, **kwargs):
    """Processing function of the plugin pair (segmentation.py:147-195): labels are written in place
    into ``current_output`` (uint32, padded by one voxel; the border stays 0)."""
    if output_volume is None:
        raise ValueError('output_volume must not be None. Please ensure the output volume is supplied in the config dict')
    if unet is None:
        raise ValueError('unet must not be none. Please ensure a unet was loaded and added to the config dict in the config function')
    if not isinstance(unet, unet_mod.UNet):
        # a foreign model object: keep the reference's generic control flow
        _make_writable(output_volume)
        config = {'unet': unet, 'output_volume': output_volume}
        process_chunks(input_volume, chunk_size, output_volume, margin, predict_chunk_feature_map, config=config)
        ws.segment_output_image(output_volume, affinities_channels=(0, 1, 2), thresholding_channel=3,
                                centroids_channel=4, out=current_output.ravel())
        output_volume[:] = 0
        return
    vol = np.ascontiguousarray(input_volume, dtype=np.float32)
    out = current_output if current_output.flags.c_contiguous else np.ascontiguousarray(current_output)
    assert out.shape == tuple(s + 2 for s in vol.shape), "current_output must be the volume padded by 1"
    unet._bind_stream()
    rc = unet.ctx.lib.isx_segment_volume_host(
        unet.ctx.handle, vol.ctypes.data_as(C.c_void_p), _lib._dims(vol.shape), _lib._dims(chunk_size),
        _lib._dims(margin if not isinstance(margin, int) else (margin,) * 3),
        out.view(np.uint32).ctypes.data_as(C.c_void_p))
    if out is not current_output:
        current_output[...] = out
    if not _is_pristine_zeros(output_volume):      # post-condition of segmentation.py:195: all zeros
        output_volume[:] = 0
    _lib.check(rc)


class FramePipe:
    """Two-deep frame pipeline over ``isx_pipe_*`` for host-resident stacks: ``submit(frame)`` stages and uploads a
    float32 (Z,Y,X) frame while earlier frames compute, ``wait(out)`` returns the labels of the oldest frame in flight
    (the (Z,Y,X) interior, 32-bit).  Used by :func:`segmentation_loop`; the U-Net's context belongs to the pipe until
    ``close()``."""

    def __init__(self, unet, shape, chunk_size, margin, depth=2):
        # the U-Net owns the pipe (UNet._frame_pipe) and is held weakly here: no reference cycle, so the 9 GB of device and
        # pinned memory go when the U-Net goes, not when the cycle collector runs
        self._unet = weakref.ref(unet)
        self._lib = unet.ctx.lib
        self.shape, self.depth = tuple(int(s) for s in shape), int(depth)
        self._h = C.c_void_p()
        self._keep = collections.deque()
        unet._bind_stream()
        _lib.check(unet.ctx.lib.isx_pipe_create(unet.ctx.handle, _lib._dims(self.shape), _lib._dims(chunk_size),
                                                _lib._dims(margin if not isinstance(margin, int) else (margin,) * 3),
                                                self.depth, C.byref(self._h)))

    @property
    def unet(self):
        u = self._unet()
        if u is None:
            raise RuntimeError("the U-Net this frame pipe belonged to is gone")
        return u

    @property
    def in_flight(self):
        return len(self._keep)

    def submit(self, frame, normalise=False):
        assert frame.dtype == np.float32 and frame.flags.c_contiguous and tuple(frame.shape) == self.shape
        _lib.check(self.unet.ctx.lib.isx_pipe_submit(self._h, frame.ctypes.data_as(C.c_void_p), 1 if normalise else 0))
        self._keep.append(frame)                   # the library reads the frame until its wait()

    def wait(self, out=None):
        """-> labels (Z,Y,X); ``out``: C-contiguous 32-bit array of that shape (e.g. a row of the int32 label store)."""
        if out is None:
            out = np.empty(self.shape, np.uint32)
        assert out.flags.c_contiguous and out.itemsize == 4 and tuple(out.shape) == self.shape
        if not self._keep:
            raise ValueError("FramePipe.wait() without a frame in flight")
        rc = self.unet.ctx.lib.isx_pipe_wait(self._h, out.ctypes.data_as(C.c_void_p), 2)
        self._keep.popleft()
        _lib.check(rc)
        return out

    def discard(self):
        """Finish the oldest frame in flight and throw its result away, whatever it was (a frame that is redone through
        the one-frame path: a NoSeedsError of the void full-size frame must not end the stack)."""
        scratch = np.empty(self.shape, np.uint32)
        self._lib.isx_pipe_wait(self._h, scratch.ctypes.data_as(C.c_void_p), 2)
        self._keep.popleft()

    def drain(self):
        """Discard whatever is still in flight (after an error): the pipe is reusable afterwards."""
        while self._keep:
            self.discard()

    def close(self):
        if self._h:
            self._lib.isx_pipe_destroy(self._h)
            self._h = C.c_void_p()
            self._keep.clear()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def segment_volume_device(unet, volume, chunk_size, margin, feature_map=None, labels=None):
    """Device-resident frame: CUDA float32 (Z,Y,X) already normalised -> CUDA int32 labels
    (Z+2,Y+2,X+2).  ``feature_map`` (5,Z,Y,X) is scratch and comes back zeroed."""
    shape = tuple(volume.shape)
    if feature_map is None:
        feature_map = torch.empty((unet.out_channels,) + shape, dtype=torch.float32, device=volume.device)
    if labels is None:
        labels = torch.zeros(tuple(s + 2 for s in shape), dtype=torch.int32, device=volume.device)
    unet._bind_stream()
    _lib.check(unet.ctx.lib.isx_segment_volume(
        unet.ctx.handle, C.c_void_p(volume.data_ptr()), _lib._dims(shape), _lib._dims(chunk_size),
        _lib._dims(margin), C.c_void_p(feature_map.data_ptr()), C.c_void_p(labels.data_ptr())))
    return labels


# ----------------
# common machinery
# ----------------

def read_config_json(path_to_json):
    with open(path_to_json, 'r') as f:
        return json.load(f)


def _new_label_store(shape, chunk_size, save_path, layer_meta):
    try:
        import zarr
        if save_path is not None:
            return zarr.open(save_path, mode='w', shape=shape, chunks=(1,) * (len(shape) - 3) + tuple(chunk_size),
                             dtype=np.int32)
        return zarr.zeros(shape=shape, chunks=(1,) * (len(shape) - 3) + tuple(chunk_size), dtype=np.int32)
    except ImportError:
        return np.zeros(shape, np.int32)


def segmentation_wrapper(processing_function, config_prep_function, napari_viewer, input_volume_layer,
                         save_dir, name, network_or_config_file, layer_reference, chunk_size, margin,
                         debug=False):
    """segmentation.py:700-830 without the Qt worker: the loop runs inline (the reference's
    ``debug=True`` behaviour, which is also its widget default)."""
    config = config_prep_function(input_volume_layer, network_or_config_file, layer_reference)
    if config is None:
        config = {}
    save_path = None
    if save_dir is not None and not debug:
        save_path = os.path.join(str(save_dir), name + '.ome.zarr')
    data = input_volume_layer.data
    layer_meta = {'scale': getattr(input_volume_layer, 'scale', None),
                  'translate': getattr(input_volume_layer, 'translate', None), 'name': name}
    output_labels = _new_label_store(data.shape, chunk_size, save_path, layer_meta)
    output_layer = output_labels
    if napari_viewer is not None:
        output_layer = napari_viewer.add_labels(output_labels, name=name, scale=layer_meta['scale'],
                                                translate=layer_meta['translate'])
    for t in segmentation_loop(napari_viewer, data, chunk_size, margin, output_labels, processing_function, config):
        print(f'Segmented frame {t}')
    return output_layer


def segmentation_loop(viewer, data, chunk_size, margin, output_labels, processing_function, config):
    """One 3-D frame at a time; yields the finished time index (segmentation.py:833-882)."""
    if data.ndim == 3:
        output = segment_single_volume(np.asarray(data).astype(np.float32), chunk_size, config, margin,
                                       processing_function)
        output_labels[...] = output
        yield 0
        return
    u = config.get('unet') if isinstance(config, dict) else None
    if (processing_function is affinity_watershed_for_chunks and isinstance(u, unet_mod.UNet) and data.shape[0] > 1
            and os.environ.get("ISX_FRAME_PIPE", "1") != "0"):
        yield from _pipelined_frames(data, chunk_size, margin, output_labels, config)
        return
    for t in range(data.shape[0]):
        if np.any(output_labels[t]):          # warm restart: frames that already have labels are skipped
            continue
        input_volume = np.asarray(data[t]).astype(np.float32)
        current_output = segment_single_volume(input_volume, chunk_size, config, margin, processing_function)
        output_labels[t, ...] = current_output
        yield t


def _pipelined_frames(data, chunk_size, margin, output_labels, config):
    """segmentation_loop over the accelerated processing function with frame t+1 crossing PCIe while frame t computes
    (``isx_pipe_*``): same frames, same per-frame results, same yield order.  ``frame /= max`` (segmentation.py:889) is the
    same IEEE float32 division on the device; the zero-slice branch (:887-888) is evaluated on the host while the frame is
    already in flight, and a frame whose shape it changes is redone through the one-frame path."""
    u = config['unet']
    shape = tuple(int(v) for v in data.shape[-3:])
    pipe = _pipe_for(u, shape, chunk_size, margin)
    direct = isinstance(output_labels, np.ndarray) and output_labels.dtype.itemsize == 4
    pending = collections.deque()                 # (t, reduced frame or None)

    def finish():
        t0, redo = pending.popleft()
        if redo is not None:                      # rare: the frame lost all-zero planes, its pipelined result is void
            pipe.discard()
            output_labels[t0, ...] = segment_single_volume(redo, chunk_size, config, margin, affinity_watershed_for_chunks)
        elif direct and output_labels[t0].flags.c_contiguous:
            pipe.wait(out=output_labels[t0])      # uint32 labels < 2^31: the bits are the int32 values
        else:
            output_labels[t0, ...] = pipe.wait()
        return t0

    try:
        for t in range(data.shape[0]):
            if np.any(output_labels[t]):          # warm restart
                continue
            frame = np.asarray(data[t])
            if frame.dtype != np.float32 or not frame.flags.c_contiguous:
                frame = np.ascontiguousarray(frame, dtype=np.float32)
            pipe.submit(frame, normalise=True)
            redo = None
            if frame.min() == 0:
                reduced = remove_sum_zero_slices(frame)
                if reduced.shape != frame.shape:
                    redo = reduced.astype(np.float32)
            pending.append((t, redo))
            if redo is not None:
                # the one-frame path needs the context: everything in flight is finished first, in order
                while pending:
                    yield finish()
            elif len(pending) == pipe.depth:
                yield finish()
        while pending:
            yield finish()
    finally:
        pipe.drain()          # only non-empty after an error; the pipe itself stays with the U-Net for the next stack


def _pipe_for(unet, shape, chunk_size, margin):
    """The U-Net's frame pipe for this geometry: its slots (two device frames + label blocks, the feature map, 2 x 2
    pinned staging buffers -- 9 GB at 512^3) are allocated once and kept until the geometry changes or the U-Net's
    context goes away (``UNet._drop_pipe``), like the context's own workspace cache."""
    m = tuple(margin) if not isinstance(margin, int) else (margin,) * 3
    key = (tuple(int(v) for v in shape), tuple(int(v) for v in chunk_size), tuple(int(v) for v in m))
    cached = getattr(unet, "_frame_pipe", None)
    if cached is not None and cached[0] == key and cached[1]._h:
        return cached[1]
    unet._drop_pipe()
    pipe = FramePipe(unet, key[0], key[1], key[2], depth=2)
    unet._frame_pipe = (key, pipe)
    return pipe


def segment_single_volume(input_volume, chunk_size, config, margin, processing_function):
    if input_volume.min() == 0:
        input_volume = remove_sum_zero_slices(input_volume)
    input_volume /= np.max(input_volume)
    # np.pad(np.zeros(shape, uint32), 1) upstream (segmentation.py:890): the same zeros, without writing them once
    current_output = np.zeros(tuple(s + 2 for s in input_volume.shape), dtype=np.uint32)
    crop = (slice(1, -1),) * current_output.ndim
    processing_function(input_volume, current_output, chunk_size, margin, **config)
    return current_output[crop]


def remove_sum_zero_slices(input_volume):
    for ax_i in range(input_volume.ndim):
        keep = [i for i in range(input_volume.shape[ax_i])
                if np.take(input_volume, i, axis=ax_i).sum() != 0]
        input_volume = np.take(input_volume, keep, axis=ax_i)
    return input_volume


segmenters = {
    'affinity-unet-watershed': affinity_unet_watershed,
}
