"""Mirror of ``iterseg.train`` (reference: src/iterseg/train.py) on the MI355X path.

* :func:`train_unet`   -- train.py:25-183 (same signature and return value ``(unet, unet_path)``)
* :class:`Trainer`     -- the device side of ``_train_step`` / ``_validate`` (train.py:344-389): forward,
  loss, backward and Adam run in ``libitersegx.so`` (``isx_train_step``); there is no torch.autograd.

Parameters, gradients and the Adam moments are four flat CUDA tensors owned here (PyTorch = memory
container), bound to the library with ``isx_train_bind``.  With ``torch.distributed`` initialised the
gradient buffer is averaged across ranks with ONE all-reduce per step (RCCL over xGMI) between backward
and the optimiser update -- data parallel, batch 1 per rank, BatchNorm statistics per rank like the
reference's single-GPU semantics.

Reference quirk kept and switchable: ``_train_loop`` calls ``optimiser.step()`` a second time on the same
gradients (train.py:310) -> ``double_step=True`` by default.
"""
from __future__ import annotations

import ctypes as C
import os
from collections import OrderedDict
from datetime import datetime

import numpy as np
import torch

from . import _lib, weights as _weights
from .unet import UNet

LOSS_KINDS = {"BCELoss": 0, "MSELoss": 1, "WeightedBCE": 2, "EpochWeightedBCE": 2, "DiceLoss": 3, "Channelwise": 4}


def channelwise_spec(losses, chan_losses, out_channels):
    """``ChannelwiseLoss(losses, chan_losses)`` (custom_loss.py:131-159) as per-channel (kind, weight):
    entry i of ``losses`` (an ``nn.BCELoss`` / ``nn.MSELoss`` instance, class or name) is applied to
    ``y_hat[:, chan_losses[i]]`` (an int or a slice), each is a mean over its elements and the result is the
    mean over entries -> channel weight 1 / (L * n_i)."""
    if losses is None or chan_losses is None or len(losses) != len(chan_losses) or not len(losses):
        raise ValueError("Channelwise needs `losses` and `chan_losses` of equal, non-zero length")
    kinds = [-1] * out_channels
    wts = [0.0] * out_channels
    L = len(losses)
    for lf, sel in zip(losses, chan_losses):
        name = lf if isinstance(lf, str) else (lf.__name__ if isinstance(lf, type) else type(lf).__name__)
        if name not in ("BCELoss", "MSELoss"):
            raise NotImplementedError(f"Channelwise: {name} per channel is not on the accelerated path")
        idx = list(range(out_channels))[sel] if isinstance(sel, slice) else [int(sel)]
        for c in idx:
            if kinds[c] != -1:
                raise ValueError(f"Channelwise: channel {c} appears in two entries")
            kinds[c] = LOSS_KINDS[name]
            wts[c] = 1.0 / (L * len(idx))
    kinds = [k if k >= 0 else 0 for k in kinds]          # channels no entry covers: weight 0
    return kinds, wts


def trainable_layout(out_channels=5):
    """``[(key, shape, offset)]`` of the flat parameter buffer: trainable state-dict entries, grouped per
    module as conv0.{weight,bias}, conv1.{weight,bias}, batch0.{weight,bias}, batch1.{weight,bias}, then the
    transposed convolutions (the order ``isx_train_bind`` documents)."""
    schema = _weights.unet_schema(1, out_channels)
    nforks = len(out_channels) if isinstance(out_channels, (tuple, list)) else 1
    mods = ["c0", "c1", "c2", "c3", "c4"] + [f"c{n}_{i}" for i in range(nforks) for n in (5, 6, 7, 8)]
    keys = []
    for m in mods:
        keys += [f"{m}.conv0.weight", f"{m}.conv0.bias", f"{m}.conv1.weight", f"{m}.conv1.bias",
                 f"{m}.batch0.weight", f"{m}.batch0.bias", f"{m}.batch1.weight", f"{m}.batch1.bias"]
    for k in range(4):
        keys += [f"up{k}.weight", f"up{k}.bias"]
    out, off = [], 0
    for k in keys:
        shape = tuple(schema[k])
        out.append((k, shape, off))
        off += int(np.prod(shape))
    return out, off


class Trainer:
    """One U-Net replica being trained on this rank's GPU."""

    def __init__(self, state_dict=None, out_channels=5, patch=(10, 256, 256), device=None, seed=0,
                 chan_final_activations=None, precision="fp32"):
        """``precision``: ``"fp32"`` (the reference's arithmetic) or ``"bf16"`` (mixed precision, BASELINE config 4: the wide
        convolutions of the forward pass and their data gradients on the bf16 matrix cores, everything else fp32)."""
        if precision not in ("fp32", "bf16"):
            raise ValueError("training precision must be 'fp32' or 'bf16'")
        self.unet = UNet(1, out_channels, device=device, batch=1, chan_final_activations=chan_final_activations,
                         precision=precision)
        self.precision = precision
        if isinstance(out_channels, list):
            out_channels = tuple(out_channels)
        if state_dict is None:
            state_dict = _weights.generate_state_dict(seed, 1, out_channels)
        self.unet.load_state_dict(state_dict)
        self.out_channels = self.unet.out_channels            # all decoders together (a tuple = the forked network)
        self.patch = tuple(int(p) for p in patch)
        dev = torch.device("cuda", self.unet.device)
        lib, h = self.unet.ctx.lib, self.unet.ctx.handle
        self.layout, n = trainable_layout(out_channels)
        n_lib = lib.isx_train_param_count(h)
        assert n_lib == n, (n_lib, n)
        self.params = torch.zeros(n, dtype=torch.float32, device=dev)
        self.grads = torch.zeros_like(self.params)
        self.adam_m = torch.zeros_like(self.params)
        self.adam_v = torch.zeros_like(self.params)
        self.unet._bind_stream()
        _lib.check(lib.isx_train_bind(h, _lib._dims(self.patch), C.c_void_p(self.params.data_ptr()),
                                      C.c_void_p(self.grads.data_ptr()), C.c_void_p(self.adam_m.data_ptr()),
                                      C.c_void_p(self.adam_v.data_ptr()), n))
        # UNet.state_dict() reads the live parameters from here.  The UNet holds the parameter TENSOR, not the Trainer: no
        # reference cycle (contexts are destroyed promptly, in creation order), and the buffer the library's bias / BatchNorm
        # pointers alias stays alive as long as the context does -- also for the UNet that train_unet hands back
        self.unet._train_params = (self.layout, self.params, self.grads, self.adam_m, self.adam_v)

    # ------------------------------------------------------------------ steps
    def _call_lib(self, x, y, loss, chan_weights, lr, n_adam, backward, channelwise=None):
        dev = self.params.device
        if self.unet._train_params is None or self.unet._train_params[1] is not self.params:
            raise RuntimeError("this Trainer's U-Net was re-loaded (load_state_dict opens a new context): build a new "
                               "Trainer from the new state dict")
        x = torch.as_tensor(np.asarray(x) if not isinstance(x, torch.Tensor) else x, dtype=torch.float32).to(dev).contiguous()
        y = torch.as_tensor(np.asarray(y) if not isinstance(y, torch.Tensor) else y, dtype=torch.float32).to(dev).contiguous()
        assert tuple(x.shape) == self.patch and tuple(y.shape) == (self.out_channels,) + self.patch, (x.shape, y.shape)
        if loss not in LOSS_KINDS:
            raise ValueError('Valid loss options are BCELoss, WeightedBCE, and DiceLoss')      # train.py:214-216
        out = (C.c_float * (1 + self.out_channels))()
        self.unet._bind_stream()
        lib, h = self.unet.ctx.lib, self.unet.ctx.handle
        if LOSS_KINDS[loss] == 4:
            kinds, wts = channelwise if channelwise is not None else (None, None)
            if kinds is None:
                raise ValueError("Channelwise needs `losses` and `chan_losses`")
            ck = (C.c_int32 * self.out_channels)(*kinds)
            cwt = (C.c_float * self.out_channels)(*wts)
            _lib.check(lib.isx_train_step_channelwise(h, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), ck, cwt,
                                                      C.c_float(lr), int(n_adam), 1 if backward else 0, out))
        else:
            cw = None
            if LOSS_KINDS[loss] == 2:
                cw = (C.c_float * self.out_channels)(*[float(w) for w in chan_weights])
            _lib.check(lib.isx_train_step(h, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), LOSS_KINDS[loss], cw,
                                          C.c_float(lr), int(n_adam), 1 if backward else 0, out))
        return float(out[0]), [float(v) for v in out[1:]]

    def _call(self, *a, **k):
        try:
            return self._call_lib(*a, **k)
        except ValueError as e:
            if str(e).startswith("all elements of input should be between 0 and 1"):
                raise RuntimeError(str(e)) from None       # what nn.BCELoss raises upstream for relu / tanh / unactivated outputs
            raise

    def train_step(self, x, y, lr=0.01, loss="BCELoss", chan_weights=None, double_step=True, group=None,
                   channelwise=None):
        """zero_grad -> forward -> loss -> backward -> Adam (x2 if ``double_step``).  Returns
        ``(loss, per_channel_losses)``."""
        n_adam = 2 if double_step else 1
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            l, cl = self._call(x, y, loss, chan_weights, lr, 0, True, channelwise)
            from .distributed import average_gradients
            average_gradients(self.grads, group)                                     # one flat bucket: 39.9 MB fp32
            self.unet._bind_stream()
            _lib.check(self.unet.ctx.lib.isx_adam_step(self.unet.ctx.handle, C.c_float(lr), n_adam))
            return l, cl
        return self._call(x, y, loss, chan_weights, lr, n_adam, True, channelwise)

    def evaluate(self, x, y, loss="BCELoss", chan_weights=None, channelwise=None):
        """Validation forward (train-mode BN, no gradient), train.py:366-389."""
        return self._call(x, y, loss, chan_weights, 0.0, 0, False, channelwise)

    # ------------------------------------------------------------------ state
    def grad_tap(self, name):
        """A gradient tensor of the last backward pass as an (M, C) numpy array (``isx_train_tap_host``):
        ``"draw:<layer>"``, ``"gact:<layer>"``, ``"dpool:<k>"``, ``"dcat:<k>"``."""
        cap = 1 << 26
        buf = np.empty(cap, np.float32)
        shp = (C.c_int64 * 2)()
        self.unet._bind_stream()
        _lib.check(self.unet.ctx.lib.isx_train_tap_host(self.unet.ctx.handle, name.encode(), buf.ctypes.data_as(C.c_void_p),
                                                        cap, shp))
        return buf[:shp[0] * shp[1]].reshape(shp[0], shp[1]).copy()

    def grad(self, key):
        for k, shape, off in self.layout:
            if k == key:
                return self.grads[off:off + int(np.prod(shape))].view(shape)
        raise KeyError(key)

    def param(self, key):
        for k, shape, off in self.layout:
            if k == key:
                return self.params[off:off + int(np.prod(shape))].view(shape)
        raise KeyError(key)

    def state_dict(self):
        """Reference-format state dict: current parameters + running statistics."""
        return self.unet.state_dict()


# --------------------------------------------------------------------------------------------------
# train_unet: same signature as the reference (train.py:25-52)
# --------------------------------------------------------------------------------------------------
def _load(i, xs):
    return np.asarray(xs[i])               # load_tensor_from_zarr (train_io.py:437-440)


def train_unet(x, vx, y, vy, ids=None, vids=None, out_dir=None, name='my-unet', channels=None, validate=True,
               log=True, epochs=3, lr=0.01, loss_function='BCELoss', chan_weights=None, weights=None,
               update_every=20, losses=None, chan_losses=None, fork_channels=None, chan_final_activations=None,
               double_step=True, precision="fp32", **kwargs):
    """Train the affinity U-Net on lists of patches ``x[i]`` (Z,Y,X) / ``y[i]`` (C,Z,Y,X).  Returns
    ``(unet, unet_path)`` like the reference; ``unet_path`` is None when ``out_dir`` is None (the reference
    leaves it unbound there, train.py:177-183)."""
    import pandas as pd
    save_output = out_dir is not None
    print('Output will be saved: ', save_output)
    print('Save directory: ', out_dir)
    if ids is None:
        ids = [name + f'_{i}' for i in range(len(x))]
    if vids is None:
        vids = [name + f'_val_{i}' for i in range(len(vx))]
    n_out = len(channels) if channels is not None else int(np.asarray(y[0]).shape[0])
    channels = tuple(channels) if channels is not None else tuple('channel_' + str(i) for i in range(n_out))
    patch = tuple(np.asarray(x[0]).shape)
    sd = None
    if weights is not None:
        sd = {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in weights.items()}
    if fork_channels is not None:
        # train.py:139-141: UNet(out_channels=fork_channels) -- one decoder per entry, outputs concatenated in channel order
        fork_channels = tuple(int(c) for c in fork_channels)
        if sum(fork_channels) != n_out:
            raise ValueError(f"fork_channels {fork_channels} do not add up to the {n_out} target channels")
    trainer = Trainer(sd, out_channels=fork_channels if fork_channels is not None else n_out, patch=patch,
                      seed=kwargs.get('seed', 0), chan_final_activations=chan_final_activations, precision=precision)
    cwise = channelwise_spec(losses, chan_losses, n_out) if loss_function == 'Channelwise' else None
    loss_dict = {'epoch': [], 'batch_num': [], 'loss': [], 'data_id': []}
    for c in channels:
        loss_dict[c] = []
    validation_dict = {'epoch': [], 'validation_loss': [], 'data_id': [], 'batch_id': []}

    def epoch_weights(e):
        if loss_function == 'EpochWeightedBCE':
            return chan_weights[e]
        return chan_weights

    def run_validation(e, batch_no):
        total = 0.0
        for i in range(len(vx)):
            vl, _ = trainer.evaluate(_load(i, vx), _load(i, vy), loss_function, epoch_weights(e), cwise)
            total += vl
            validation_dict['epoch'].append(e)
            validation_dict['validation_loss'].append(vl)
            validation_dict['data_id'].append(vids[i])
            validation_dict['batch_id'].append(batch_no)
        if len(vx):
            print(f'Epoch {e} - validation loss: {total / len(vx)}')

    if log and save_output:
        # helpers.write_log (helpers.py:50-53): one line per call appended to <out_dir>/log.txt
        _write_log('Training U-Net (MI355X path)', out_dir)
        _write_log(f'Loss function: {loss_function}', out_dir)
        if chan_weights is not None:
            _write_log(f'Channel weights: {chan_weights}', out_dir)
        _write_log(f'Epochs: {epochs}; learning rate: {lr}; patches: {len(x)} training / {len(vx)} validation', out_dir)
        _write_log('Initial weights: ' + ('supplied state dict' if weights is not None else 'seeded generator'), out_dir)
        _write_log(f'Channels: {channels}', out_dir)
        if fork_channels is not None:
            _write_log(f'Channels per fork (according to channel order): {fork_channels}', out_dir)
    unet_path = None
    for e in range(epochs):
        if validate and e == 0:
            run_validation(e, 0)
        running_loss = 0.0
        for i in range(len(x)):
            l, cl = trainer.train_step(_load(i, x), _load(i, y), lr=lr, loss=loss_function,
                                       chan_weights=epoch_weights(e), double_step=double_step, channelwise=cwise)
            loss_dict['epoch'].append(e)
            loss_dict['batch_num'].append(i)
            loss_dict['loss'].append(l)
            loss_dict['data_id'].append(ids[i])
            for c, v in zip(channels, cl):
                loss_dict[c].append(v)
            running_loss += l
            if i % update_every == (update_every - 1):
                print(f'Epoch {e} - running loss: {running_loss / update_every}')
                running_loss = 0.0
        if validate:
            run_validation(e, (e + 1) * len(x))
        if save_output:
            _save_checkpoint(trainer.state_dict(), out_dir, f'{name}_epoch-{e}')
    if save_output:
        unet_path = _save_checkpoint(trainer.state_dict(), out_dir, name, r=True)
        pd.DataFrame(loss_dict).to_csv(os.path.join(out_dir, 'loss_' + name + '.csv'))
        if validate:
            # _save_output (train.py:425-432): the last validation predictions, one file per validation patch
            requested = trainer.unet.precision
            trainer.unet.set_precision("fp32")
            v_y_hats = [trainer.unet(np.asarray(_load(i, vx), np.float32)[None, None])[0] for i in range(len(vx))]
            trainer.unet.set_precision(requested)           # the network handed back keeps the arithmetic it was trained in
            _save_output(v_y_hats, vids, out_dir, name='_validation')
            pd.DataFrame(validation_dict).to_csv(os.path.join(out_dir, 'validation-loss_' + name + '.csv'))
    return trainer.unet, unet_path


def _write_log(string, out_dir, log_name='log.txt'):
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, log_name), 'a') as f:
        f.write(string + '\n')


def _save_output(y_hats, ids, out_dir, name=''):
    """``<id><name>_output.tif`` per prediction like the reference (tifffile); without tifffile in the environment
    the same array goes to ``<id><name>_output.npy``."""
    assert len(y_hats) == len(ids)
    os.makedirs(out_dir, exist_ok=True)
    try:
        from tifffile import TiffWriter
    except ImportError:
        TiffWriter = None
    for yh, i in zip(y_hats, ids):
        a = np.asarray(yh, np.float32)
        if TiffWriter is not None:
            with TiffWriter(os.path.join(out_dir, i + name + '_output.tif')) as tiff:
                tiff.write(a)
        else:
            np.save(os.path.join(out_dir, i + name + '_output.npy'), a)


def _save_checkpoint(checkpoint, out_dir, name, r=False):
    d = datetime.now().strftime("%y%d%m_%H%M%S")        # day-month order, as upstream (train.py:416)
    path = os.path.join(out_dir, d + '_unet_' + name + '.pt')
    os.makedirs(out_dir, exist_ok=True)
    torch.save(OrderedDict(checkpoint), path)
    if r:
        return path
