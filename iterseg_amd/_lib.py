"""ctypes binding of ``libitersegx.so`` (C ABI declared in ``include/itersegx.h``).

There is no CPU fallback: if the shared library is missing it must be built with
``python -m iterseg_amd._build`` (hipcc), and creating a :class:`Context` without a usable
MI355X raises.  PyTorch is used by callers only as the device-memory / stream container;
this module passes raw ``data_ptr()`` addresses.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# ISX_LIB: development only -- load another build of the library (an ablation / profiling build under scratch/libs/) instead of
# copying it over the product library, so that an interrupted experiment cannot leave the wrong code installed
LIB_PATH = os.environ.get("ISX_LIB") or os.path.join(HERE, "libitersegx.so")

ISX_OK = 0
ISX_ERR_NO_DEVICE, ISX_ERR_HIP, ISX_ERR_ARG, ISX_ERR_NO_SEEDS, ISX_ERR_STATE, ISX_ERR_NOMEM, ISX_ERR_INTERNAL = -1, -2, -3, -4, -5, -6, -7
PROF_CLASSES = {"conv_mfma": 0, "conv_direct": 1, "elementwise": 2, "ws_filters": 3, "ws_ccl": 4, "ws_flood": 5}

_vp, _i64, _i32, _f32 = C.c_void_p, C.c_int64, C.c_int32, C.c_float
_i64p = C.POINTER(C.c_int64)
_i32p = C.POINTER(C.c_int32)
_f32p = C.POINTER(C.c_float)

# name -> (restype, argtypes); every symbol include/itersegx.h declares
SIGNATURES = {
    "isx_version": (C.c_int, []),
    "isx_last_error": (C.c_char_p, []),
    "isx_device_count": (C.c_int, []),
    "isx_ctx_create": (_vp, [C.c_int]),
    "isx_ctx_destroy": (None, [_vp]),
    "isx_ctx_set_stream": (C.c_int, [_vp, _vp]),
    "isx_ctx_synchronize": (C.c_int, [_vp]),
    "isx_ctx_device": (C.c_int, [_vp]),
    "isx_ctx_workspace_bytes": (_i64, [_vp]),
    "isx_ctx_release_workspace": (None, [_vp]),
    "isx_ctx_reload_knobs": (C.c_int, [_vp]),
    "isx_prof_enable": (C.c_int, [_vp, C.c_int]),
    "isx_prof_get": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_double), _i64p]),
    "isx_make_chunks": (_i64, [_i64p, _i64p, _i64p, _i32p, _i64]),
    "isx_load_weight_host": (C.c_int, [_vp, C.c_char_p, _vp, _i64p, C.c_int]),
    "isx_weights_commit": (C.c_int, [_vp, C.c_int]),
    "isx_weights_commit_forked": (C.c_int, [_vp, _i32p, C.c_int]),
    "isx_get_weight_host": (C.c_int, [_vp, C.c_char_p, _vp, _i64]),
    "isx_unet_forward_chunks": (C.c_int, [_vp, _vp, _i64p, _i32p, _i64, _i64p, _vp]),
    "isx_predict_volume": (C.c_int, [_vp, _vp, _i64p, _i32p, _i64, _i64, _i64p, _vp]),
    "isx_unet_tap_host": (C.c_int, [_vp, C.c_char_p, _vp, _i64, _i64p]),
    "isx_unet_set_batch": (C.c_int, [_vp, C.c_int]),
    "isx_unet_set_precision": (C.c_int, [_vp, C.c_int]),
    "isx_unet_set_final_activations": (C.c_int, [_vp, C.POINTER(C.c_int32), C.c_int]),
    "isx_unet_precision_report": (C.c_int, [_vp, _i32p, C.c_int, _i32p]),
    "isx_train_param_count": (_i64, [_vp]),
    "isx_train_bind": (C.c_int, [_vp, _i64p, _vp, _vp, _vp, _vp, _i64]),
    "isx_train_step": (C.c_int, [_vp, _vp, _vp, C.c_int, _f32p, _f32, C.c_int, C.c_int, _f32p]),
    "isx_train_step_channelwise": (C.c_int, [_vp, _vp, _vp, _i32p, _f32p, _f32, C.c_int, C.c_int, _f32p]),
    "isx_adam_step": (C.c_int, [_vp, _f32, C.c_int]),
    "isx_train_refresh_weights": (C.c_int, [_vp]),
    "isx_train_tap_host": (C.c_int, [_vp, C.c_char_p, _vp, _i64, _i64p]),
    "isx_training_labels": (C.c_int, [_vp, _vp, _i64p, C.c_int, _i32p, _i32p, _i32p, C.POINTER(C.c_double), _vp]),
    "isx_patch_stats": (C.c_int, [_vp, _vp, _i64p, _i32p, C.c_int, _i64p, C.POINTER(C.c_double), C.POINTER(C.c_float),
                                  C.POINTER(C.c_float)]),
    "isx_patch_gather": (C.c_int, [_vp, _vp, _i64p, _i32p, C.c_int, _i64p, _vp]),
    "isx_augment_array": (C.c_int, [_vp, _vp, _i64p, C.c_int, C.c_float, C.c_float, C.c_int, _vp, C.c_float, C.c_int, _i32p, _vp]),
    "isx_label_contingency": (C.c_int, [_vp, _vp, _vp, _i64, _vp, _vp, _i64, _i64p]),
    "isx_ws_set_shard": (C.c_int, [_vp, C.c_int, C.c_int]),
    "isx_comm_unique_id": (C.c_int, [_vp]),
    "isx_comm_init": (C.c_int, [_vp, _vp, C.c_int, C.c_int]),
    "isx_comm_destroy": (C.c_int, [_vp]),
    "isx_allreduce_grads": (C.c_int, [_vp, _vp, _i64]),
    "isx_gather_tiles": (C.c_int, [_vp, _vp, _vp, _i64]),
    "isx_allreduce_max_labels": (C.c_int, [_vp, _vp, _i64]),
    "isx_ws_gaussian": (C.c_int, [_vp, _vp, _vp, _i64p, _i32p]),
    "isx_ws_threshold_otsu": (C.c_int, [_vp, _vp, _i64, _f32p]),
    "isx_ws_mask": (C.c_int, [_vp, _vp, _i64p, _vp, _f32p]),
    "isx_ws_seeds": (C.c_int, [_vp, _vp, _i64p, C.c_int, _f32, _vp, _i64, _i64p]),
    "isx_ws_filter": (C.c_int, [_vp, _vp, _i64p, _i64, _i64, _vp, _vp, _i64, _vp, _i64p]),
    "isx_ws_components": (C.c_int, [_vp, _vp, _i64p, _vp, _vp]),
    "isx_affinity_watershed": (C.c_int, [_vp, _vp, _vp, _i64, _vp, _vp, _i64p, _f32p]),
    "isx_segment_output_image": (C.c_int, [_vp, _vp, _i64p, _vp, _i64, _i64, _f32p, _vp, _i64, _i64p, _vp, _f32p]),
    "isx_segment_output_image_scaled": (C.c_int, [_vp, _vp, _i64p, _vp, _i64, _i64, _f32p, _vp, _i64, _i64p, _vp, _f32p, _f32p]),
    "isx_normalise_volume": (C.c_int, [_vp, _vp, _i64]),
    "isx_segment_volume": (C.c_int, [_vp, _vp, _i64p, _i64p, _i64p, _vp, _vp]),
    "isx_segment_volume_host": (C.c_int, [_vp, _vp, _i64p, _i64p, _i64p, _vp]),
    "isx_pipe_create": (C.c_int, [_vp, _i64p, _i64p, _i64p, C.c_int, C.POINTER(_vp)]),
    "isx_pipe_submit": (C.c_int, [_vp, _vp, C.c_int]),
    "isx_pipe_wait": (C.c_int, [_vp, _vp, C.c_int]),
    "isx_pipe_destroy": (None, [_vp]),
}

_LIB = None


class IsxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libitersegx error {code}: {msg}")
        self.code = code


def load(build_if_missing=False):
    """Load the shared library (once).  Raises if it is absent -- never falls back."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        if build_if_missing:
            from . import _build
            _build.build()
        else:
            raise ImportError(
                f"{LIB_PATH} not found: build the HIP extension first (python -m iterseg_amd._build); "
                "iterseg_amd has no CPU path")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _LIB = lib
    return lib


def _dims(shape):
    a = (C.c_int64 * 3)(*[int(s) for s in shape])
    return a


class NoSeedsError(ValueError):
    """Zero seeds survive (ISX_ERR_NO_SEEDS).  A ValueError like the reference's own failure at this point
    (np.apply_along_axis, watershed.py:50), as its own subclass so that callers can tell it from a bad argument."""


def check(rc):
    if rc == ISX_OK:
        return
    msg = load().isx_last_error().decode("utf-8", "replace")
    if rc == ISX_ERR_NO_SEEDS:
        # the reference fails here with this exact exception type (np.apply_along_axis on an
        # empty seed list, src/iterseg/watershed.py:50)
        raise NoSeedsError("Cannot apply_along_axis when any iteration dimensions are 0 [" + msg + "]")
    if rc == ISX_ERR_ARG:
        raise ValueError(msg)
    raise IsxError(rc, msg)


class Context:
    """Owns an ``isx_ctx`` (device workspace, weights, stream binding)."""

    def __init__(self, device=0, stream=None):
        self.lib = load()
        self._h = self.lib.isx_ctx_create(int(device))
        if not self._h:
            raise IsxError(ISX_ERR_NO_DEVICE, self.lib.isx_last_error().decode())
        self.device = int(device)
        if stream is not None:
            self.set_stream(stream)

    @property
    def handle(self):
        return self._h

    def set_stream(self, stream):
        check(self.lib.isx_ctx_set_stream(self._h, C.c_void_p(int(stream))))

    def use_torch_stream(self):
        import torch
        self.set_stream(torch.cuda.current_stream(self.device).cuda_stream)

    def synchronize(self):
        check(self.lib.isx_ctx_synchronize(self._h))

    def workspace_bytes(self):
        return int(self.lib.isx_ctx_workspace_bytes(self._h))

    def release_workspace(self):
        self.lib.isx_ctx_release_workspace(self._h)

    def reload_knobs(self):
        """Re-read the ISX_* environment knobs (they are read once, at creation): tests' stress modes."""
        check(self.lib.isx_ctx_reload_knobs(self._h))

    def set_flood_shard(self, rank=0, world=1):
        """Flood only the mask components of ``rank`` out of ``world`` (isx_ws_set_shard); (0, 1) = all."""
        check(self.lib.isx_ws_set_shard(self._h, int(rank), int(world)))

    def prof_enable(self, on=True):
        check(self.lib.isx_prof_enable(self._h, 1 if on else 0))

    def prof_get(self, cls):
        ms, n = C.c_double(), C.c_int64()
        check(self.lib.isx_prof_get(self._h, PROF_CLASSES[cls], C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def close(self):
        if getattr(self, "_h", None):
            self.lib.isx_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def make_chunks_table(shape, chunk, margin):
    """``isx_make_chunks``: the C-side chunk table (int32 [n, 9])."""
    lib = load()
    if isinstance(margin, int):
        margin = (margin,) * 3
    n = lib.isx_make_chunks(_dims(shape), _dims(chunk), _dims(margin), None, 0)
    if n < 0:
        check(int(n))
    tab = np.empty((n, 9), np.int32)
    n2 = lib.isx_make_chunks(_dims(shape), _dims(chunk), _dims(margin), tab.ctypes.data_as(_i32p), n)
    assert n2 == n
    return tab
