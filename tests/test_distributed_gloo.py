"""World-size-2 test of the frame / chunk sharding on CPU (gloo), with the oracle as the injected compute:
the N>1 path must give exactly the single-process result."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from iterseg_amd import chunking
from iterseg_amd.distributed import frames_for_rank, predict_volume_sharded, segment_stack


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _frame_labels(frame):
    """Stand-in compute for the test: oracle watershed of a feature map derived from the frame."""
    from oracle import watershed as ows
    from tests.synth import smooth_feature_map
    seed = int(frame[0, 0, 0] * 1000) % 997
    fm = smooth_feature_map(frame.shape, seed)
    lab = np.zeros(tuple(s + 2 for s in frame.shape), np.uint32)
    ows.segment_output_image(fm, out=lab.ravel())
    return lab[1:-1, 1:-1, 1:-1].astype(np.int32)


def _component_sharded_labels(rank, world):
    """(full labels, labels of the components ``rank`` owns) with the product's ownership rule restated on the
    host: components in root order, owner = floor(voxel offset * world / total voxels)."""
    from scipy import ndimage as ndi
    from oracle import watershed as ows
    from tests.synth import blob_feature_map
    fm = blob_feature_map((16, 40, 40), 3, 0.25, 2.0)
    pshape = tuple(s + 2 for s in fm.shape[1:])
    full = np.zeros(pshape, np.uint32)
    _, seeds, mask = ows.segment_output_image(fm, out=full.ravel())
    aff = np.zeros((3,) + pshape, np.float32)
    aff[:, 1:-1, 1:-1, 1:-1] = fm[:3] / fm[:3].reshape(3, -1).max(1)[:, None, None, None]
    comp, n = ndi.label(mask)                       # label ids follow the smallest raveled index = root order
    sizes = ndi.sum(mask, comp, index=np.arange(1, n + 1)).astype(np.int64)
    seeded = [c for c in range(1, n + 1) if (comp[tuple((seeds + 1).T)] == c).any()]
    off = np.concatenate([[0], np.cumsum([sizes[c - 1] for c in seeded])])
    owned = np.zeros(pshape, np.uint32)
    for k, (zz, yy, xx) in enumerate(seeds + 1):
        owned[zz, yy, xx] = k + 1                   # seed voxels are labelled on every rank
    for j, c in enumerate(seeded):
        if int(off[j] * world // off[-1]) == rank:
            owned[comp == c] = full[comp == c]
    return full.astype(np.int32), owned.astype(np.int32)


def _stack():
    rng = np.random.default_rng(5)
    return rng.random((5, 8, 24, 24), dtype=np.float32)          # 5 frames: uneven split over 2 ranks


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        data = _stack()
        out = segment_stack(data, _frame_labels)
        # fewer frames than ranks: the rank without a frame still takes part in the gather
        one = segment_stack(data[:1], _frame_labels)
        assert tuple(one.shape) == (1,) + data.shape[1:] and torch.equal(one[0], out[0])
        # chunk-sharded predict: every chunk's crop is filled with (chunk id + 1)
        shape, chunk, margin = (12, 48, 40), (4, 32, 32), (1, 8, 8)
        tab = chunking.chunk_table(shape, chunk, margin)

        def predict_chunks(first, count, fm):
            for i in range(first, first + count):
                r = tab[i]
                fm[:, r[0] + r[3]:r[0] + r[4], r[1] + r[5]:r[1] + r[6], r[2] + r[7]:r[2] + r[8]] = float(i + 1)
        fm = torch.empty((2,) + shape)
        predict_volume_sharded(predict_chunks, tab, fm)
        # data-parallel gradient averaging on a flat buffer the size of the U-Net's parameters
        from iterseg_amd.distributed import average_gradients
        g = torch.full((9972673,), float(rank + 1))
        g[rank::7] += 0.5
        average_gradients(g)
        expect = torch.full((9972673,), 1.5)
        for r in range(world):
            expect[r::7] += 0.25
        assert torch.equal(g, expect)
        # component-sharded flood: each rank floods the components its range owns (oracle, one component at a
        # time), the max all-reduce must rebuild the single-process labels
        from iterseg_amd.distributed import combine_labels
        full, owned = _component_sharded_labels(rank, world)
        lab = combine_labels(torch.from_numpy(owned))
        assert torch.equal(lab, torch.from_numpy(full))
        q.put((rank, out.numpy(), fm.numpy()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_frames_for_rank():
    assert frames_for_rank(16, 3, 8) == [3, 11]
    assert sorted(sum((frames_for_rank(5, r, 2) for r in range(2)), [])) == list(range(5))


@pytest.mark.timeout(300)
def test_world_size_2_equals_single_process():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict()
    for _ in range(world):
        r, labels, fm = q.get(timeout=240)
        results[r] = (labels, fm)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    data = _stack()
    ref = np.stack([_frame_labels(f) for f in data])
    assert ref.max() > 0
    shape, chunk, margin = (12, 48, 40), (4, 32, 32), (1, 8, 8)
    tab = chunking.chunk_table(shape, chunk, margin)
    fm_ref = np.zeros((2,) + shape, np.float32)
    for i, r in enumerate(tab):
        fm_ref[:, r[0] + r[3]:r[0] + r[4], r[1] + r[5]:r[1] + r[6], r[2] + r[7]:r[2] + r[8]] = i + 1
    assert (fm_ref > 0).all()                       # crops tile the volume
    for r in range(world):
        assert np.array_equal(results[r][0], ref), f"rank {r}: gathered labels differ"
        assert np.array_equal(results[r][1], fm_ref), f"rank {r}: sharded feature map differs"
