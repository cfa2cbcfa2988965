"""Host-side U-Net object: holds a reference-format state dict and runs it on the MI355X.

Mirror of ``iterseg.unet.UNet`` (reference: src/iterseg/unet.py:126-364) for the inference
path: same constructor arguments (single decoder), ``load_state_dict`` / ``state_dict`` with the
reference's key names and OIDHW float32 layouts, and ``__call__`` on a ``(1, 1, D, H, W)``
tensor/array returning ``(1, C, D, H, W)`` -- what ``predict_chunk_feature_map`` expects
(predict.py:120-123).  The arithmetic happens in ``libitersegx.so`` (HIP); this class never
computes on the host.  As in the reference's inference, BatchNorm runs in training mode: batch
statistics per chunk, running statistics updated as a side effect.
"""
from __future__ import annotations

import ctypes as C
import os
from collections import OrderedDict

import numpy as np
import torch

from . import _lib, weights as _weights


# Arithmetic of the wide inference convolutions when the caller does not choose: split-f16 is fp32-class (error against
# a float64 oracle at or below the fp32 kernel's own) and about twice as fast; "fp32" keeps exact fp32
# products.  Training always runs fp32.
DEFAULT_PRECISION = "split-f16"


class UNet:
    def __init__(self, in_channels=1, out_channels=5, down_factors=(1, 2, 2), up='convolution',
                 downsample_1_at_bottom=True, chan_final_activations=None, device=None, batch=32,
                 precision=None):
        # a tuple = the forked network (unet.py:152-157): one decoder c5_i..c8_i per entry on a shared encoder, outputs
        # concatenated (unet.py:313-325); the reference's decoder() serves forks 0 and 1 (unet.py:330-362)
        self.forked = isinstance(out_channels, (tuple, list))
        self.fork_channels = tuple(int(c) for c in out_channels) if self.forked else (int(out_channels),)
        if len(self.fork_channels) > 2:
            raise NotImplementedError("more than two decoders: the reference's decoder() only handles forks 0 and 1")
        if sum(self.fork_channels) > 8:
            raise NotImplementedError("at most 8 output channels over all decoders")
        if in_channels != 1 or tuple(down_factors) != (1, 2, 2) or up != 'convolution' or not downsample_1_at_bottom:
            raise NotImplementedError("only the configuration iterseg ships (in=1, (1,2,2), transposed conv) is built")
        # unet.py:205-209: chan_final_activations[i] is handed to ConvModule(final=...) of decoder i (an IndexError upstream too
        # when the list is shorter than the decoders; surplus entries are never read); ConvModule.forward (unet.py:98-106) knows
        # 'relu', 'softmax', 'sigmoid', 'tanh' and leaves the BatchNorm output alone for any other string
        if chan_final_activations is None:
            chan_final_activations = ['sigmoid'] * len(self.fork_channels)
        self.final_activations = tuple(chan_final_activations[i] for i in range(len(self.fork_channels)))
        self.in_channels, self.out_channels = in_channels, int(sum(self.fork_channels))
        self._schema_out = self.fork_channels if self.forked else self.out_channels
        self.device = torch.cuda.current_device() if device is None else int(device)
        self.ctx = _lib.Context(self.device)
        _lib.check(self.ctx.lib.isx_unet_set_batch(self.ctx.handle, int(batch)))
        self._set_final_activations()
        self._batch = int(batch)
        self._loaded = False
        self._sd = None
        self._train_params = None          # (layout, flat parameter tensor) once a Trainer is bound to this context
        self.precision = "fp32"
        self.set_precision(precision if precision is not None else os.environ.get("ISX_CONV_PRECISION", DEFAULT_PRECISION))

    _FINAL_CODES = {'sigmoid': 0, 'relu': 1, 'tanh': 2, 'softmax': 4}          # ISX_FINAL_*; anything else: 3 (none)

    def _set_final_activations(self):
        codes = (C.c_int32 * len(self.final_activations))(*[self._FINAL_CODES.get(a, 3) for a in self.final_activations])
        _lib.check(self.ctx.lib.isx_unet_set_final_activations(self.ctx.handle, codes, len(self.final_activations)))

    # ------------------------------------------------------------------ weights
    def load_state_dict(self, state_dict, strict=True):
        """Accepts torch tensors or numpy arrays keyed like the reference (predict.py:34)."""
        sd = OrderedDict()
        for k, v in state_dict.items():
            sd[k] = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
        _weights.check_state_dict(sd, self.in_channels, self._schema_out)      # strict key/shape check
        if self._loaded:                          # a committed context cannot be re-loaded: start a new one
            self._drop_pipe()
            self.ctx.close()
            # the trainer bound to the old context (if any) is gone with it: state_dict() must not read its buffers
            self._train_params = None
            self.ctx = _lib.Context(self.device)
            _lib.check(self.ctx.lib.isx_unet_set_batch(self.ctx.handle, self._batch))
            self._set_final_activations()
            self.set_precision(self.precision)
        lib, h = self.ctx.lib, self.ctx.handle
        for k, v in sd.items():
            if k.endswith("num_batches_tracked"):
                a = np.ascontiguousarray(v, np.int64).reshape(())
                _lib.check(lib.isx_load_weight_host(h, k.encode(), a.ctypes.data_as(C.c_void_p), None, 0))
            else:
                a = np.ascontiguousarray(v, np.float32)
                shp = (C.c_int64 * a.ndim)(*a.shape)
                _lib.check(lib.isx_load_weight_host(h, k.encode(), a.ctypes.data_as(C.c_void_p), shp, a.ndim))
        if self.forked:
            fc = (C.c_int32 * len(self.fork_channels))(*self.fork_channels)
            _lib.check(lib.isx_weights_commit_forked(h, fc, len(self.fork_channels)))
        else:
            _lib.check(lib.isx_weights_commit(h, self.out_channels))
        self._sd = sd
        self._loaded = True
        return self

    def state_dict(self):
        """Reference-format state dict (torch CPU tensors); running statistics are read back."""
        out = OrderedDict()
        for k, v in self._sd.items():
            if ".running_" in k or k.endswith("num_batches_tracked"):
                buf = np.empty(v.shape, v.dtype if not k.endswith("tracked") else np.int64)
                _lib.check(self.ctx.lib.isx_get_weight_host(self.ctx.handle, k.encode(),
                                                            buf.ctypes.data_as(C.c_void_p), max(buf.nbytes, 8)))
                out[k] = torch.from_numpy(buf.copy())
            else:
                out[k] = torch.from_numpy(np.array(v, copy=True))
        # a U-Net that is (or was) being trained: the live parameters are the trainer's flat buffer, not the arrays
        # this object was loaded from (``unet, _ = train_unet(...); torch.save(unet.state_dict(), ...)`` is the
        # reference idiom, train.py:183)
        if self._train_params is not None:
            layout, params = self._train_params[:2]
            for k, shape, off in layout:
                out[k] = params[off:off + int(np.prod(shape))].view(shape).detach().cpu().clone()
        return out

    def set_precision(self, precision):
        """Arithmetic of the wide convolutions: ``"fp32"`` (exact fp32 products), ``"split-f16"`` (fp32 operands split
        into two halves each, f16 matrix cores, fp32 accumulation: fp32-class results, the default), or the reduced-
        precision side modes ``"f16"`` / ``"bf16"`` (plain operands, one MFMA per product; see include/itersegx.h).
        Training runs fp32 except in ``"bf16"`` (mixed precision: bf16 forward / data-gradient convolutions, fp32
        master weights, weight gradients, BatchNorm and Adam)."""
        modes = {"fp32": 0, "split-f16": 1, "f16": 2, "bf16": 3}
        if precision not in modes:
            raise ValueError(f"precision must be one of {sorted(modes)}")
        _lib.check(self.ctx.lib.isx_unet_set_precision(self.ctx.handle, modes[precision]))
        self.precision = precision

    def precision_report(self):
        """How each convolution of the last forward ran: ``{"c0.conv0": "fp32" | "reduced" | "fp32-range-fallback", ...}``
        (``isx_unet_precision_report``).  A trained network with large BatchNorm weights can push layers of the split-f16 /
        f16 modes back to the fp32 kernel; this is where that shows."""
        modes = (C.c_int32 * 26)()
        nf = C.c_int32(0)
        n = self.ctx.lib.isx_unet_precision_report(self.ctx.handle, modes, 26, C.byref(nf))
        if n < 0:
            _lib.check(n)
        mods = ["c0", "c1", "c2", "c3", "c4", "c5_0", "c6_0", "c7_0", "c8_0", "c5_1", "c6_1", "c7_1", "c8_1"]
        names = [f"{m}.conv{j}" for m in mods for j in (0, 1)]
        kind = {0: "fp32", 1: "reduced", 2: "fp32-range-fallback"}
        return {names[i]: kind[modes[i]] for i in range(n)}

    def _drop_pipe(self):
        """Free the cached frame pipe (segmentation._pipe_for): it holds device and pinned memory of this context."""
        cached = getattr(self, "_frame_pipe", None)
        self._frame_pipe = None
        if cached is not None:
            cached[1].close()

    def __del__(self):
        try:
            self._drop_pipe()        # before the context: isx_pipe_destroy synchronises on it
        except Exception:
            pass

    def set_batch(self, batch):
        _lib.check(self.ctx.lib.isx_unet_set_batch(self.ctx.handle, int(batch)))
        self._batch = int(batch)

    # ------------------------------------------------------------------ forward
    def _bind_stream(self):
        self.ctx.set_stream(torch.cuda.current_stream(self.device).cuda_stream)

    def _require_idle(self):
        """The context belongs to the frame pipe's compute thread while frames are in flight (segmentation._pipelined_frames
        hands results out between yields): a forward from the consumer's side would race with it."""
        cached = getattr(self, "_frame_pipe", None)
        if cached is not None and cached[1].in_flight:
            raise RuntimeError("this U-Net has frames in flight in a frame pipe (segmentation_loop is mid-stack): finish or "
                               "drain the pipe before calling the U-Net directly")

    def forward_chunks(self, volume, starts, chunk):
        self._require_idle()
        """volume: CUDA float32 (Z,Y,X); starts: (n,3) ints -> CUDA (n, C, cz, cy, cx)."""
        assert self._loaded, 'Please ensure a unet is loaded and supplied'
        self._bind_stream()
        starts = np.ascontiguousarray(starts, np.int32).reshape(-1, 3)
        out = torch.empty((len(starts), self.out_channels) + tuple(int(c) for c in chunk), dtype=torch.float32,
                          device=volume.device)
        _lib.check(self.ctx.lib.isx_unet_forward_chunks(
            self.ctx.handle, C.c_void_p(volume.data_ptr()), _lib._dims(volume.shape),
            starts.ctypes.data_as(C.POINTER(C.c_int32)), len(starts), _lib._dims(chunk), C.c_void_p(out.data_ptr())))
        return out

    def __call__(self, x):
        """(1,1,D,H,W) tensor or array -> (1,C,D,H,W) tensor on the same kind of device."""
        was_np = not isinstance(x, torch.Tensor)
        t = torch.as_tensor(x, dtype=torch.float32)
        on_gpu = t.is_cuda
        assert t.dim() == 5 and t.shape[0] == 1 and t.shape[1] == 1, "expects a (1,1,D,H,W) chunk"
        vol = t[0, 0].contiguous().to(f"cuda:{self.device}")
        y = self.forward_chunks(vol, [(0, 0, 0)], vol.shape)
        if was_np:
            return y.cpu().numpy()
        return y if on_gpu else y.cpu()

    forward = __call__

    def predict_volume(self, volume, chunk_size, margin, feature_map=None, first=0, count=None):
        """P2-P4 on device: CUDA (Z,Y,X) volume -> CUDA (C,Z,Y,X) feature map (crops stitched)."""
        assert self._loaded, 'Please ensure a unet is loaded and supplied'
        self._require_idle()
        self._bind_stream()
        tab = _lib.make_chunks_table(volume.shape, chunk_size, margin)
        if feature_map is None:
            feature_map = torch.zeros((self.out_channels,) + tuple(volume.shape), dtype=torch.float32,
                                      device=volume.device)
        count = len(tab) - first if count is None else count
        _lib.check(self.ctx.lib.isx_predict_volume(
            self.ctx.handle, C.c_void_p(volume.data_ptr()), _lib._dims(volume.shape),
            tab.ctypes.data_as(C.POINTER(C.c_int32)), int(first), int(count), _lib._dims(chunk_size),
            C.c_void_p(feature_map.data_ptr())))
        return feature_map

    def tap(self, name):
        """Named intermediate of chunk 0 of the last forward, as a (C, D, H, W) numpy array."""
        cap = 1 << 25                        # the largest tensor of a (10, 256, 256) chunk: 32 channels at full resolution
        buf = np.empty(cap, np.float32)
        shp = (C.c_int64 * 4)()
        _lib.check(self.ctx.lib.isx_unet_tap_host(self.ctx.handle, name.encode(), buf.ctypes.data_as(C.c_void_p),
                                                  cap, shp))
        shape = tuple(shp)
        return buf[:int(np.prod(shape))].reshape(shape).copy()

    # torch.nn.Module look-alikes the reference's callers touch
    def cuda(self):
        return self

    def to(self, *a, **k):
        return self

    def train(self, mode=True):
        if not mode:
            raise NotImplementedError("iterseg runs its U-Net with train-mode BatchNorm; eval mode is not built")
        return self
