"""One-process-per-GPU sharding of the hot path (``torch.distributed``; ``nccl`` = RCCL over xGMI on the
GPU box, ``gloo`` in the CPU tests).

The reference has no parallelism of any kind; what shards naturally is

* **frames of a (t,z,y,x) stack** -- fully independent (per-frame normalisation, BatchNorm statistics,
  Otsu threshold, seeds and label ids; reference: src/iterseg/segmentation.py:873-882).  Frame ``t`` goes
  to rank ``t mod N``; there is no data-path collective, only one ``all_gather`` of the int32 label
  tiles at the end (BASELINE config 3);
* **chunks of one volume** for the U-Net stage -- independent because BatchNorm statistics are per chunk and
  overlap is recomputed, not exchanged (predict.py:81-95).  Each rank predicts a contiguous block of the
  chunk list into a zeroed feature map; crops are disjoint, so a sum all-reduce is an exact union.

* **mask components of one volume** for the flood (W4) -- exact, because the heap order inside one 6-connected
  component does not depend on the others.  Every rank runs the cheap W0-W3 stages on the
  replicated feature map, floods its own range of components and the label volumes are combined with an
  element-wise max (disjoint supports; seed voxels carry the same label everywhere).  One giant component
  cannot be split: that case degenerates to "replicas only".

The compute callables are injected, so the CPU tests drive exactly this code with the oracle.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

from .chunking import shard_range


def frames_for_rank(n_frames, rank, world_size):
    """Frame indices owned by ``rank``: t mod N == rank (2 frames each for 16 frames on 8 GPUs)."""
    return list(range(int(rank), int(n_frames), int(world_size)))


def segment_stack(data, segment_frame, group=None, gather=True, device=None):
    """Segment a ``(T, Z, Y, X)`` stack, frames sharded over the ranks of ``group``.

    ``segment_frame(frame: np.ndarray float32 (Z,Y,X)) -> labels (Z,Y,X)`` (numpy or tensor, int32).
    Returns the full ``(T, Z, Y, X)`` int32 label stack on every rank when ``gather`` (tensor on
    ``device``), else ``{t: labels}`` for the local frames.
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    T = int(data.shape[0])
    shape = tuple(int(s) for s in data.shape[1:])
    mine = frames_for_rank(T, rank, world)
    local = {}
    for t in mine:
        lab = segment_frame(np.asarray(data[t]).astype(np.float32))
        local[t] = torch.as_tensor(lab).to(torch.int32)
    if not gather:
        return local
    if device is None:
        device = next(iter(local.values())).device if local else torch.device("cpu")
    per_rank = (T + world - 1) // world
    send = torch.zeros((per_rank,) + shape, dtype=torch.int32, device=device)
    for i, t in enumerate(mine):
        send[i] = local[t].to(device)
    if world == 1:
        tiles = [send]
    else:
        tiles = [torch.empty_like(send) for _ in range(world)]
        dist.all_gather(tiles, send, group=group)
    out = torch.zeros((T,) + shape, dtype=torch.int32, device=device)
    for r in range(world):
        for i, t in enumerate(frames_for_rank(T, r, world)):
            out[t] = tiles[r][i]
    return out


def predict_volume_sharded(predict_chunks, n_chunks, feature_map, group=None):
    """U-Net stage of ONE volume, chunk list block-partitioned over the ranks.

    ``predict_chunks(first, count, feature_map)`` must write the centre crops of chunks
    ``[first, first+count)`` into ``feature_map`` (a zero-initialised ``(C, Z, Y, X)`` tensor).  Afterwards the
    feature map is summed across ranks (disjoint crops => exact union) so every rank can run the watershed.
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    first, end = shard_range(n_chunks, rank, world)
    feature_map.zero_()
    predict_chunks(first, end - first, feature_map)
    if world > 1:
        dist.all_reduce(feature_map, op=dist.ReduceOp.SUM, group=group)
    return feature_map


def average_gradients(flat_grads, group=None):
    """Data-parallel training (BASELINE config 4): ONE all-reduce of the flat gradient buffer
    (9 972 673 floats = 39.9 MB) per step, then divide by the world size.  With RCCL on the fully
    connected xGMI mesh this is a reduce-scatter + all-gather over all 7 links."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world > 1:
        dist.all_reduce(flat_grads, op=dist.ReduceOp.SUM, group=group)
        flat_grads.div_(world)
    return flat_grads


def combine_labels(labels, group=None):
    """Union of component-sharded label volumes: element-wise max all-reduce (int32, 0 = not mine)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world > 1:
        dist.all_reduce(labels, op=dist.ReduceOp.MAX, group=group)
    return labels


def segment_volume_sharded(unet, volume, chunk_size, margin, group=None, feature_map=None, labels=None):
    """ONE normalised CUDA volume on all ranks of ``group`` (BASELINE config 2 / the 512^3 north star as a strong-
    scaling job): chunk-sharded U-Net -> sum all-reduce of the feature map (2.68 GB at 512^3) -> W0-W3 replicated ->
    component-sharded flood -> max all-reduce of the int32 labels (0.54 GB).  Returns the padded label volume
    (Z+2, Y+2, X+2), identical on every rank and identical to the single-GPU result."""
    from . import watershed as _ws
    from .chunking import chunk_table
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    shape = tuple(int(s) for s in volume.shape)
    if feature_map is None:
        feature_map = torch.zeros((unet.out_channels,) + shape, dtype=torch.float32, device=volume.device)
    if labels is None:
        labels = torch.zeros(tuple(s + 2 for s in shape), dtype=torch.int32, device=volume.device)
    n_chunks = len(chunk_table(shape, chunk_size, margin))
    predict_volume_sharded(
        lambda first, count, fm: unet.predict_volume(volume, chunk_size, margin, feature_map=fm, first=first, count=count),
        n_chunks, feature_map, group)
    unet._bind_stream()
    unet.ctx.set_flood_shard(rank, world)
    try:
        _ws.segment_output_image(feature_map, (0, 1, 2), 4, 3, out=labels, ctx=unet.ctx)
    finally:
        unet.ctx.set_flood_shard(0, 1)
    return combine_labels(labels, group)
