"""One-process-per-GPU sharding of the hot path (``torch.distributed``; ``nccl`` = RCCL over xGMI on the
GPU box, ``gloo`` in the CPU tests).

The reference has no parallelism of any kind; what shards naturally is

* **frames of a (t,z,y,x) stack** -- fully independent (per-frame normalisation, BatchNorm statistics,
  Otsu threshold, seeds and label ids; reference: src/iterseg/segmentation.py:873-882).  Frame ``t`` goes
  to rank ``t mod N``; there is no data-path collective, only one ``all_gather`` of the int32 label
  tiles at the end (BASELINE config 3);
* **chunks of one volume** for the U-Net stage -- independent because BatchNorm statistics are per chunk and
  overlap is recomputed, not exchanged (predict.py:81-95).  The z-major chunk list is cut at z-row boundaries, so
  every rank's centre crops tile one z-slab of the feature map; the slabs are exchanged with ONE all-gather
  (335 MB per rank at 512^3 on 8 ranks: 1x the feature map on the wire, where a sum all-reduce of zero-filled maps
  moved 2x).

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
        # every rank must hand the collective a buffer on the same kind of device, also a rank that owns no frame
        # (T < world size): RCCL wants the rank's GPU, gloo the host
        if dist.is_initialized() and dist.get_backend(group) == "nccl":
            device = torch.device("cuda", torch.cuda.current_device())
        else:
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


def z_rows(table):
    """Rows of the z-major chunk table: ``[(first_chunk, n_chunks, z_lo, z_hi)]`` -- the chunks that share one z start;
    their centre crops cover the slab ``[z_lo, z_hi)`` of the volume completely and nothing else."""
    table = np.asarray(table)
    rows, i, n = [], 0, len(table)
    while i < n:
        j = i
        while j < n and table[j, 0] == table[i, 0]:
            j += 1
        rows.append((i, j - i, int(table[i, 0] + table[i, 3]), int(table[i, 0] + table[i, 4])))
        i = j
    return rows


def slab_of_rank(table, rank, world_size):
    """``(first_chunk, n_chunks, z_lo, z_hi)`` of the block of z-rows owned by ``rank`` (possibly empty)."""
    rows = z_rows(table)
    r0, r1 = shard_range(len(rows), rank, world_size)
    if r0 == r1:
        z = rows[r0][2] if r0 < len(rows) else (rows[-1][3] if rows else 0)
        return (rows[r0][0] if r0 < len(rows) else len(table)), 0, z, z
    return rows[r0][0], sum(r[1] for r in rows[r0:r1]), rows[r0][2], rows[r1 - 1][3]


def predict_volume_sharded(predict_chunks, table, feature_map, group=None):
    """U-Net stage of ONE volume, the chunk table (``chunking.chunk_table``) block-partitioned by z-rows over the ranks.

    ``predict_chunks(first, count, feature_map)`` must write the centre crops of chunks ``[first, first+count)`` into
    ``feature_map`` (``(C, Z, Y, X)``).  Afterwards every rank holds the whole feature map: the z-slabs are exchanged
    with one all-gather (RCCL over xGMI on the GPU box), so every rank can run the watershed.
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    first, count, z0, z1 = slab_of_rank(table, rank, world)
    if count:
        predict_chunks(first, count, feature_map)
    if world > 1:
        slabs = [slab_of_rank(table, r, world) for r in range(world)]
        zmax = max(s[3] - s[2] for s in slabs)
        C = feature_map.shape[0]
        send = feature_map.new_zeros((C, zmax) + tuple(feature_map.shape[2:]))
        send[:, :z1 - z0] = feature_map[:, z0:z1]
        recv = feature_map.new_empty((world,) + tuple(send.shape))
        dist.all_gather(list(recv.unbind(0)), send, group=group)       # views of one buffer (gloo and RCCL both take the list form)
        for r, (_, _, a, b) in enumerate(slabs):
            if r != rank and b > a:
                feature_map[:, a:b] = recv[r, :, :b - a]
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
    """Union of component-sharded label volumes: element-wise max all-reduce (int32, 0 = not mine; label ids are
    seed positions + 1 < 2^31, so the signed maximum is the unsigned one)."""
    assert labels.dtype == torch.int32
    # label ids are 1..n_seeds and there are fewer seeds than voxels, so this bounds them below 2^31
    assert labels.numel() < 2 ** 31, "label ids could reach 2^31: the signed max would not be the unsigned one"
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world > 1:
        dist.all_reduce(labels, op=dist.ReduceOp.MAX, group=group)
    return labels


def segment_volume_sharded(unet, volume, chunk_size, margin, group=None, feature_map=None, labels=None):
    """ONE normalised CUDA volume on all ranks of ``group`` (BASELINE config 2 / the 512^3 north star as a strong-
    scaling job): chunk-sharded U-Net -> all-gather of the feature-map z-slabs (335 MB per rank at 512^3, N = 8) -> W0-W3 replicated ->
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
    table = chunk_table(shape, chunk_size, margin)
    predict_volume_sharded(
        lambda first, count, fm: unet.predict_volume(volume, chunk_size, margin, feature_map=fm, first=first, count=count),
        table, feature_map, group)
    unet._bind_stream()
    unet.ctx.set_flood_shard(rank, world)
    try:
        _ws.segment_output_image(feature_map, (0, 1, 2), 4, 3, out=labels, ctx=unet.ctx)
    finally:
        unet.ctx.set_flood_shard(0, 1)
    return combine_labels(labels, group)
