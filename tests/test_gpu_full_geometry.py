"""Parity at the reference's own geometry -- chunk (10,256,256), margin (1,64,64) (segmentation.py:31-32) -- on the
BASELINE.json configurations, not on toy chunks:

* config 2 (64x512x512, 72 chunks): HIP feature map against the torch-CPU oracle on a 3-chunk sub-slab (<= 1e-4 on the
  sigmoid maps, the stated fp32 tolerance), and -- staged parity, as smoke() does at toy size -- the HIP labels of the
  WHOLE volume against the C oracle's watershed of the HIP feature map, bit for bit (one 8.3 M-voxel component floods
  through the speculative-window kernel here);
* config 1 (32x256x256) end to end through the plugin entry;
* 512^3 (the north-star volume) and 1024^3 (config 5): size-independent properties -- labels identical across the flood
  paths (window / serial heap), border zero, label ids are exactly the seed ids, every labelled voxel lies in the mask.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CHUNK, MARGIN = (10, 256, 256), (1, 64, 64)


@pytest.fixture(scope="module")
def unet():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from iterseg_amd import weights
    from iterseg_amd.unet import UNet
    u = UNet(1, 5, batch=32)
    u.load_state_dict(weights.generate_state_dict(0))
    return u


def test_config2_feature_map_subslab_and_whole_volume_labels(unet):
    from iterseg_amd import segmentation as seg, weights
    from oracle import unet_torch as ou, watershed as ows
    from tests.synth import positive_volume
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    shape = (64, 512, 512)
    vol = positive_volume(shape, 0)
    dvol = torch.from_numpy(vol).cuda()
    fm = unet.predict_volume(dvol, CHUNK, MARGIN)
    # (a) a 3-chunk sub-slab of the same volume through the oracle: chunks are independent (per-chunk BatchNorm
    # statistics, recomputed overlap), so a (10, 256, 512) corner is cut into the chunks x = 0, 128, 256 exactly like
    # the corresponding region of the full chunk table
    sub = np.ascontiguousarray(vol[:10, :256, :])
    ref_sub = ou.predict_volume(weights.generate_state_dict(0), sub, CHUNK, MARGIN)
    got_sub = unet.predict_volume(torch.from_numpy(sub).cuda(), CHUNK, MARGIN).cpu().numpy()
    assert np.abs(got_sub - ref_sub).max() <= 1e-4
    # the same chunks inside the full volume: crops z 0..9 of chunk z0 = 0 are written for z < 9 only (margin), y < 192
    # (launch sets of different sizes may pick different tilings, i.e. another summation order of the BatchNorm statistics)
    assert float((fm[:, :9, :192, :] - torch.from_numpy(got_sub[:, :9, :192, :]).cuda()).abs().max()) <= 5e-6
    # (b) whole-volume labels: HIP flood == C oracle on the HIP feature map
    lab = seg.segment_volume_device(unet, dvol, CHUNK, MARGIN)
    ref = np.zeros(tuple(s + 2 for s in shape), np.uint32)
    ows.segment_output_image(fm.cpu().numpy(), out=ref.ravel())
    got = lab.cpu().numpy().astype(np.uint32)
    assert np.array_equal(got, ref)
    assert (got > 0).sum() > 8_000_000                       # the giant component is there and flooded


def test_config1_plugin_entry(unet):
    from iterseg_amd import segmentation as seg
    from oracle import watershed as ows
    from tests.synth import positive_volume
    vol = positive_volume((32, 256, 256), 1)
    cur = np.zeros(tuple(s + 2 for s in vol.shape), np.uint32)
    seg.affinity_watershed_for_chunks(vol.copy(), cur, CHUNK, MARGIN, unet=unet,
                                      output_volume=np.zeros((5,) + vol.shape, np.float32))
    fm = unet.predict_volume(torch.from_numpy(vol).cuda(), CHUNK, MARGIN).cpu().numpy()
    ref = np.zeros_like(cur)
    ows.segment_output_image(fm, out=ref.ravel())
    assert np.array_equal(cur, ref) and cur.max() > 1000


def _properties(lab, mask, seeds):
    """size-independent checks on a padded label volume (torch, device)"""
    assert not bool(lab[0].any()) and not bool(lab[-1].any()) and not bool(lab[:, 0].any()) and not bool(lab[:, -1].any())
    assert not bool(lab[:, :, 0].any()) and not bool(lab[:, :, -1].any())
    assert not bool((lab[~mask] != 0).any())                             # floods never leave the mask
    sz, sy, sx = (torch.as_tensor(seeds, device=lab.device) + 1).T
    ids = lab[sz, sy, sx].to(torch.int64)
    assert torch.equal(ids, torch.arange(1, len(seeds) + 1, device=lab.device))      # label = seed position + 1
    present = torch.unique(lab)
    assert int(present.max()) <= len(seeds) and present.numel() - 1 == len(seeds)      # ids contiguous 1..n


@pytest.mark.parametrize("n", [512, 1024])
def test_large_volume_properties_and_flood_path_agreement(unet, n, flood_knobs):
    """512^3 / 1024^3 with the seeded random-weight U-Net (BASELINE configs 'north star' and 5): the watershed of the HIP
    feature map is deterministic across flood implementations, and has the structural properties of the reference."""
    from iterseg_amd import watershed as ws
    from tests.synth import positive_volume
    shape = (n, n, n)
    rng = np.random.default_rng(0)
    # the volume is generated in z slabs (a 1024^3 float32 volume is 4 GiB; the uniform generator is streamed)
    dvol = torch.empty(shape, dtype=torch.float32, device="cuda")
    for z0 in range(0, n, 64):
        dvol[z0:z0 + 64] = torch.from_numpy(1.0 - rng.random((64, n, n), dtype=np.float32)).cuda()
    dvol /= dvol.max()
    unet.set_batch(64)
    fm = unet.predict_volume(dvol, CHUNK, MARGIN)
    del dvol
    pshape = tuple(s + 2 for s in shape)
    lab_a = torch.zeros(pshape, dtype=torch.int32, device="cuda")
    seg, seeds, mask = ws.segment_output_image(fm, (0, 1, 2), 4, 3, out=lab_a)
    assert len(seeds) > 0
    _properties(lab_a, torch.as_tensor(mask, device="cuda"), seeds)
    flood_knobs({"ISX_FLOOD_COOP_MIN": "100000000"})                    # every component on the one-lane heap
    lab_b = torch.zeros(pshape, dtype=torch.int32, device="cuda")
    ws.segment_output_image(fm, (0, 1, 2), 4, 3, out=lab_b)
    assert torch.equal(lab_a, lab_b)
    unet.set_batch(32)


@pytest.mark.parametrize("cold_min", [None, "200000"])
def test_structured_512_labels_against_oracle(cold_min, flood_knobs):
    """The structured 512^3 volume of bench.py's `watershed_structured` leg (eight independently seeded blob tiles, ~10 M
    mask voxels in ~5000 components): whole-volume labels against the C oracle, bit for bit.  This input is the
    reproducer of a lost-residual bug: with three commit bursts per sub-round a component WITHOUT a cold arena can
    overflow its LDS run in the middle of an enqueue and hand over to the serial heap there -- every residual gathered
    for that burst has to reach the heap (two voxels were mislabelled when the rest of the batch was dropped).  The
    second case raises the arena threshold so that many more components take that hand-over."""
    from iterseg_amd import watershed as ws
    from oracle import watershed as ows
    from tests.synth import blob_feature_map_tiled
    if cold_min:
        flood_knobs({"ISX_FLOOD_COLD_MIN": cold_min})
    n = 512
    fm = blob_feature_map_tiled(n, 0, 0.12, 3.0)
    pshape = (n + 2,) * 3
    ref = np.zeros(pshape, np.uint32)
    ows.segment_output_image(fm, out=ref.ravel(), threads=16)
    lab = torch.zeros(pshape, dtype=torch.int32, device="cuda")
    ws.segment_output_image(torch.from_numpy(fm).cuda(), (0, 1, 2), 4, 3, out=lab)
    got = lab.cpu().numpy().astype(np.uint32)
    assert int((ref > 0).sum()) > 5_000_000
    assert int((got != ref).sum()) == 0


@pytest.mark.parametrize("seed", [1, 3])
def test_structured_512_giant_components_against_oracle(seed):
    """The structured 512^3 construction with the seeds that contain a giant component (seed 1: ~5 M voxels of tie plateaus,
    seed 3: 345 k), i.e. the inputs on which the work-group flood kernel runs by default -- scratch/time_structured.py's
    volumes, on which a development build of that kernel faulted in round 3 ("The round-3 memory fault").
    Whole-volume labels against the C oracle, bit for bit."""
    from iterseg_amd import watershed as ws
    from oracle import watershed as ows
    from tests.synth import blob_feature_map_tiled
    n = 512
    fm = blob_feature_map_tiled(n, seed, 0.12, 3.0)
    pshape = (n + 2,) * 3
    ref = np.zeros(pshape, np.uint32)
    ows.segment_output_image(fm, out=ref.ravel(), threads=16)
    lab = torch.zeros(pshape, dtype=torch.int32, device="cuda")
    _, _, mask = ws.segment_output_image(torch.from_numpy(fm).cuda(), (0, 1, 2), 4, 3, out=lab)
    _, sizes = ws.connected_components(mask)
    assert int(torch.as_tensor(sizes).max()) >= 262144            # the work-group kernel's threshold
    assert int((lab.cpu().numpy().astype(np.uint32) != ref).sum()) == 0


def test_mirrored_512_one_giant_component_against_oracle():
    """One 256^3 demo volume mirrored to 512^3 (tests/synth.blob_feature_map_mirrored): the components that touch a mirror
    plane merge with their images into ONE ~7 M-voxel blob component next to ~10 k small ones.  This is the input that
    tripped the flood's sub-round watchdog in round 2 (a window key sorting above the run head after a mid-insert run
    eviction -- the tier rule is now re-evaluated after every eviction and checked at every eviction);
    whole-volume labels against the C oracle, bit for bit, and no invariant report."""
    from iterseg_amd import watershed as ws
    from oracle import watershed as ows
    from tests.synth import blob_feature_map_mirrored
    n = 512
    fm = blob_feature_map_mirrored(n, 0, 0.12, 3.0)
    pshape = (n + 2,) * 3
    ref = np.zeros(pshape, np.uint32)
    ows.segment_output_image(fm, out=ref.ravel(), threads=16)
    lab = torch.zeros(pshape, dtype=torch.int32, device="cuda")
    _, _, mask = ws.segment_output_image(torch.from_numpy(fm).cuda(), (0, 1, 2), 4, 3, out=lab)
    got = lab.cpu().numpy().astype(np.uint32)
    _, sizes = ws.connected_components(mask)
    assert int(torch.as_tensor(sizes).max()) > 3_000_000          # the giant component is there
    assert int((got != ref).sum()) == 0


def test_config3_two_frames_at_stated_size(unet):
    """BASELINE configs[2] at its stated frame size: two frames of 32x512x512 (36 chunks each) through the device-resident
    call that bench.py's stack_step makes, through the pipelined host loop (segmentation_loop -> isx_pipe_*) and through
    the plugin's processing function frame by frame: identical labels, structural properties -- and, staged parity as for
    config 2, the labels of each frame against the C oracle's watershed of the HIP feature map, bit for bit."""
    from iterseg_amd import segmentation as seg
    from oracle import watershed as ows
    from tests.synth import positive_volume
    shape = (32, 512, 512)
    data = np.stack([positive_volume(shape, t) for t in (0, 1)])
    store = np.zeros(data.shape, np.int32)
    config = {"unet": unet, "output_volume": seg.a_w_output_volume(data, 5)}
    done = list(seg.segmentation_loop(None, data, CHUNK, MARGIN, store, seg.affinity_watershed_for_chunks, config))
    assert done == [0, 1]
    for t in (0, 1):
        dev = seg.segment_volume_device(unet, torch.from_numpy(data[t]).cuda(), CHUNK, MARGIN)
        assert np.array_equal(dev[1:-1, 1:-1, 1:-1].cpu().numpy(), store[t]), t
        cur = np.zeros(tuple(s + 2 for s in shape), np.uint32)
        seg.affinity_watershed_for_chunks(data[t].copy(), cur, CHUNK, MARGIN, **config)
        assert np.array_equal(cur[1:-1, 1:-1, 1:-1].view(np.int32), store[t]), t
        assert store[t].max() > 0 and not cur[0].any() and not cur[:, 0].any() and not cur[:, :, 0].any()
        fm = unet.predict_volume(torch.from_numpy(data[t] / data[t].max()).cuda(), CHUNK, MARGIN).cpu().numpy()
        ref = np.zeros(tuple(s + 2 for s in shape), np.uint32)
        ows.segment_output_image(fm, out=ref.ravel(), threads=16)
        assert np.array_equal(cur, ref), t
    assert not np.array_equal(store[0], store[1])                  # the frames differ


def test_config5_1024_cubed_f16(unet):
    """BASELINE configs[4]: 1024^3 with the fp16 U-Net (precision="f16": plain f16 operands, fp32 accumulate) + the
    int32 watershed -- all 6272 chunks in the reduced mode (no layer falls back to fp32 with these weights), structural
    properties of the labels, and the f16 feature map within its stated tolerance (4e-3) of the split-f16 one on the
    first launch set."""
    from iterseg_amd import watershed as ws
    n = 1024
    shape = (n, n, n)
    rng = np.random.default_rng(0)
    dvol = torch.empty(shape, dtype=torch.float32, device="cuda")
    for z0 in range(0, n, 64):
        dvol[z0:z0 + 64] = torch.from_numpy(1.0 - rng.random((64, n, n), dtype=np.float32)).cuda()
    dvol /= dvol.max()
    unet.set_batch(64)
    main = unet.precision
    try:
        ref_head = unet.predict_volume(dvol[:10].contiguous(), CHUNK, MARGIN).clone()
        unet.set_precision("f16")
        fm = unet.predict_volume(dvol, CHUNK, MARGIN)
        rep = unet.precision_report()
        assert "fp32-range-fallback" not in rep.values()
        assert sum(v == "reduced" for v in rep.values()) == 16      # the 15 wide layers + the folded tail layer
        head = unet.predict_volume(dvol[:10].contiguous(), CHUNK, MARGIN)
        assert float((head - ref_head).abs().max()) <= 4e-3
    finally:
        unet.set_precision(main)
        unet.set_batch(32)
    del dvol, ref_head, head
    pshape = tuple(s + 2 for s in shape)
    lab = torch.zeros(pshape, dtype=torch.int32, device="cuda")
    seg, seeds, mask = ws.segment_output_image(fm, (0, 1, 2), 4, 3, out=lab)
    assert len(seeds) > 0
    _properties(lab, torch.as_tensor(mask, device="cuda"), seeds)
