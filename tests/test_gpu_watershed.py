"""GPU parity of the HIP watershed stage (through the C ABI) against the golden vectors of the
reference and against the CPU oracle on seeded inputs.  Bit-exact: labels, masks, seeds,
float32 filter outputs."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

FM_CASES = ["ws_fm_10x48x48", "ws_fm_9x40x56", "ws_stages_0", "ws_stages_2", "ws_stages_3"]
KERNEL_CASES = ["ws_k_ties", "ws_k_giant", "ws_k_noseed_comp", "ws_k_scale_axis", "ws_k_scale_neg",
                "ws_k_negvals"]


@pytest.fixture(scope="module")
def ws():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from iterseg_amd import watershed
    return watershed


def _fm(golden, name):
    d = golden(name)
    if "feature_map" in d.files:
        return d, d["feature_map"]
    return d, golden("predict_" + name.split("ws_fm_")[1])["feature_map"]


def _seeds_equal_mod_ties(got, ref, g):
    if got.shape != ref.shape:
        return False
    if np.array_equal(got, ref):
        return True
    return (np.array_equal(g[tuple(got.T)], g[tuple(ref.T)])
            and set(map(tuple, got)) == set(map(tuple, ref)))


@pytest.mark.parametrize("name", FM_CASES + ["ws_stages_1"])
def test_filters_bit_exact(ws, golden, name):
    d, fm = _fm(golden, name)
    sm = ws.gaussian(fm[3], 2)
    assert np.array_equal(sm, d["mask_smooth"])
    assert ws.threshold_otsu(sm) == d["thresh"]
    assert np.array_equal(ws._get_mask(fm[3]), d["mask_raw"])
    assert np.array_equal(ws.gaussian(fm[4], (0, 1, 1)), d["cent_smooth"])
    pk = ws._get_centroids(fm[4])
    assert _seeds_equal_mod_ties(pk, d["seeds_raw"], d["cent_smooth"])
    mp = np.pad(d["mask_raw"], 1, constant_values=0)
    m2, c2 = ws._remove_unwanted_objects(mp, d["seeds_raw"] + 1, min_area=10, max_area=10000000)
    assert np.array_equal(m2, d["mask_filtered_padded"])
    assert np.array_equal(c2, d["seeds_filtered_padded"])


@pytest.mark.parametrize("name", FM_CASES)
def test_segment_output_image_plugin_path(ws, golden, name):
    d, fm = _fm(golden, name)
    pshape = tuple(s + 2 for s in fm.shape[1:])
    cur = np.zeros(pshape, np.uint32)
    seg, seeds, mask = ws.segment_output_image(fm, affinities_channels=(0, 1, 2), thresholding_channel=3,
                                               centroids_channel=4, out=cur.ravel())
    assert np.array_equal(mask, d["mask_filtered_padded"])
    # every fixture case has distinct seed intensities, so the seed order (= the label ids) is pinned: no
    # "same support" downgrade here -- instance labels are compared bit for bit
    assert np.array_equal(seeds + 1, d["seeds_filtered_padded"])
    assert np.array_equal(cur, d["labels_padded"])
    assert np.array_equal(seg, cur[1:-1, 1:-1, 1:-1])


def test_zero_seeds_raises_like_reference(ws, golden):
    d = golden("ws_stages_1")
    with pytest.raises(ValueError):
        ws.segment_output_image(d["feature_map"], (0, 1, 2), 4, 3)
    with pytest.raises(ValueError):
        ws.affinity_watershed(np.zeros((3, 4, 4, 4), np.float32), np.zeros((0, 3), np.int64),
                              np.zeros((4, 4, 4), bool))


@pytest.mark.parametrize("name", KERNEL_CASES)
def test_affinity_watershed_kernel_boundary(ws, golden, name):
    d = golden(name)
    ref = d["labels_padded"]
    scale = d["scale"] if "scale" in d.files else None
    out = np.zeros(ref.size, ref.dtype)
    lab = ws.affinity_watershed(d["aff_padded"], d["seeds_padded"], d["mask_padded"], scale=scale, out=out)
    assert lab.dtype == ref.dtype and np.array_equal(lab, ref)


def test_affinity_watershed_default_mask(ws, golden):
    d = golden("ws_k_giant_nomask")
    lab = ws.affinity_watershed(d["aff_padded"], d["seeds_padded"], None)
    assert lab.dtype == np.int32 and np.array_equal(lab, d["labels_padded"])


SEEDS_IN_UPSTREAM_ORDER = {"ws_blobs40": True, "ws_blobs32": True}


@pytest.mark.parametrize("name", ["ws_blobs40", "ws_blobs32"])
def test_reference_synthetic_blobs(ws, golden, name):
    d = golden(name)
    vol = d["volume"]
    assert np.array_equal(ws.gaussian(vol[3], (0, 1, 1)), d["cent_smooth"])
    assert _seeds_equal_mod_ties(ws._get_centroids(vol[3]), d["seeds_raw"], d["cent_smooth"])
    lab = ws.affinity_watershed(d["aff_padded"], d["seeds_padded"], d["mask_padded"])
    assert np.array_equal(lab[1:-1, 1:-1, 1:-1], d["labels"])
    # full pipeline with the absolute threshold the reference's self-check uses (watershed.py:266)
    seg, seeds, mask = ws.segment_output_image(vol, (0, 1, 2), 3, 4, absolute_thresh=0.5)
    assert np.array_equal(mask, d["mask_padded"])
    # distance-transform plateaus give equal-intensity seeds, which upstream orders by an unstable argsort: the
    # label ids are then a permutation of ours.  Which branch runs is pinned per fixture (SEEDS_IN_UPSTREAM_ORDER).
    exact = bool(np.array_equal(seeds + 1, d["seeds_padded"]))
    assert exact == SEEDS_IN_UPSTREAM_ORDER[name]
    if exact:
        assert np.array_equal(seg, d["labels"])
    else:
        assert np.array_equal(seg > 0, d["labels"] > 0)
        assert set(map(tuple, seeds + 1)) == set(map(tuple, d["seeds_padded"]))


def test_size_filter_and_plateaus(ws, golden):
    d = golden("ws_sizefilter")
    m2, c2 = ws._remove_unwanted_objects(d["mask_padded"], d["seeds_padded"], int(d["min_area"]),
                                         int(d["max_area"]))
    assert np.array_equal(m2, d["mask_out"]) and np.array_equal(c2, d["seeds_out"])
    p = golden("ws_plateau")
    pk = ws._get_centroids(p["image"], gaussian=False)
    assert set(map(tuple, pk)) == set(map(tuple, p["peaks"])) and len(pk) == len(p["peaks"])
    assert len(ws._get_centroids(p["const"], gaussian=False)) == 0
    assert ws.threshold_otsu(p["const"]) == p["otsu_const"]


@pytest.mark.parametrize("shape,seed", [((24, 96, 80), 0), ((7, 33, 129), 1), ((40, 64, 64), 2)])
def test_against_oracle_on_seeded_fields(ws, shape, seed):
    """Larger seeded inputs than the fixtures: HIP path == CPU oracle, bit for bit."""
    from oracle import watershed as ows
    from tests.synth import smooth_feature_map
    fm = smooth_feature_map(shape, seed)
    pshape = tuple(s + 2 for s in shape)
    ref = np.zeros(pshape, np.uint32)
    rseg, rseeds, rmask = ows.segment_output_image(fm, out=ref.ravel())
    cur = np.zeros(pshape, np.uint32)
    seg, seeds, mask = ws.segment_output_image(fm, (0, 1, 2), 4, 3, out=cur.ravel())
    assert np.array_equal(mask, rmask)
    assert np.array_equal(seeds, rseeds)
    assert np.array_equal(cur, ref)
    assert (cur > 0).sum() > 100


FLOOD_ENVS = {
    # every component through the speculative-window flood (LDS tiers only) / through the one-lane heap
    "coop-all": {"ISX_FLOOD_SMALL_MAX": "0", "ISX_FLOOD_COOP_MIN": "1", "ISX_FLOOD_COLD_MIN": "100000000"},
    "serial-all": {"ISX_FLOOD_COOP_MIN": "100000000"},
    # window flood with a cold arena and a deliberately tiny hot tier: bucket append / select-drain / eviction /
    # compaction carry most of the queue even on small inputs
    "cold-stress": {"ISX_FLOOD_SMALL_MAX": "0", "ISX_FLOOD_COOP_MIN": "1", "ISX_FLOOD_COLD_MIN": "1", "ISX_FLOOD_RUN_LIMIT": "96", "ISX_FLOOD_DRAIN": "48"},
    "cold-tiny": {"ISX_FLOOD_SMALL_MAX": "0", "ISX_FLOOD_COOP_MIN": "1", "ISX_FLOOD_COLD_MIN": "1", "ISX_FLOOD_RUN_LIMIT": "8", "ISX_FLOOD_DRAIN": "4"},
    # window corner cases: a window of one episode (pure sequential episodes), a few episodes with the shortest
    # possible discovery list (every started episode stalls until it is the front and commits partially), a full
    # window with short lists, and short lists over a tiny hot tier (window evictions + mid-window insertions)
    "win-one": {"ISX_FLOOD_SMALL_MAX": "0", "ISX_FLOOD_COOP_MIN": "1", "ISX_FLOOD_COLD_MIN": "1", "ISX_FLOOD_WIN": "1"},
    "win-tiny": {"ISX_FLOOD_SMALL_MAX": "0", "ISX_FLOOD_COOP_MIN": "1", "ISX_FLOOD_COLD_MIN": "1", "ISX_FLOOD_WIN": "3", "ISX_FLOOD_LIST": "6",
                 "ISX_FLOOD_RUN_LIMIT": "8", "ISX_FLOOD_DRAIN": "4"},
    "win-shortlist": {"ISX_FLOOD_SMALL_MAX": "0", "ISX_FLOOD_COOP_MIN": "1", "ISX_FLOOD_COLD_MIN": "100000000", "ISX_FLOOD_LIST": "8"},
    "win-mid": {"ISX_FLOOD_SMALL_MAX": "0", "ISX_FLOOD_COOP_MIN": "1", "ISX_FLOOD_COLD_MIN": "1", "ISX_FLOOD_WIN": "17", "ISX_FLOOD_LIST": "12",
                "ISX_FLOOD_RUN_LIMIT": "40", "ISX_FLOOD_DRAIN": "16"},
    # the work-group kernel (ws_flood_group.h: one episode per lane, 512-wide window, chunked lists) for EVERY component:
    # defaults, LDS tiers only (overflow -> hand-over to the serial heap), a tiny hot tier, a window of one / three episodes
    # with the shortest lists (forced commits of the front), short lists, a mid-size window, one commit burst per sub-round
    "group-all": {"ISX_FLOOD_COOP_MIN": "1", "ISX_FLOOD_COLD_MIN": "1", "ISX_FLOOD_GROUP_MIN": "1"},
    "group-nocold": {"ISX_FLOOD_COOP_MIN": "1", "ISX_FLOOD_COLD_MIN": "100000000", "ISX_FLOOD_GROUP_MIN": "1"},
    "group-cold-tiny": {"ISX_FLOOD_COOP_MIN": "1", "ISX_FLOOD_COLD_MIN": "1", "ISX_FLOOD_GROUP_MIN": "1", "ISX_FLOOD_RUN_LIMIT": "8",
                        "ISX_FLOOD_DRAIN": "4"},
    "group-win-one": {"ISX_FLOOD_COOP_MIN": "1", "ISX_FLOOD_COLD_MIN": "1", "ISX_FLOOD_GROUP_MIN": "1", "ISX_FLOOD_GWIN": "1"},
    "group-win-tiny": {"ISX_FLOOD_COOP_MIN": "1", "ISX_FLOOD_COLD_MIN": "1", "ISX_FLOOD_GROUP_MIN": "1", "ISX_FLOOD_GWIN": "3",
                       "ISX_FLOOD_GLIST": "6", "ISX_FLOOD_RUN_LIMIT": "8", "ISX_FLOOD_DRAIN": "4"},
    "group-shortlist": {"ISX_FLOOD_COOP_MIN": "1", "ISX_FLOOD_COLD_MIN": "100000000", "ISX_FLOOD_GROUP_MIN": "1", "ISX_FLOOD_GLIST": "8"},
    "group-mid": {"ISX_FLOOD_COOP_MIN": "1", "ISX_FLOOD_COLD_MIN": "1", "ISX_FLOOD_GROUP_MIN": "1", "ISX_FLOOD_GWIN": "17",
                  "ISX_FLOOD_GLIST": "12", "ISX_FLOOD_RUN_LIMIT": "40", "ISX_FLOOD_DRAIN": "16", "ISX_FLOOD_BURSTS": "1"},
    # the small instantiation of the single-wave kernel (W = 48, T = 24, run 512: ~30 KB of LDS, five components per CU;
    # round 3) for EVERY component: LDS tiers only (its 512-entry run overflows early: hand-over to the serial heap), with an
    # arena, a tiny hot tier, one episode, three episodes with the shortest lists, short lists
    "small-all": {"ISX_FLOOD_COOP_MIN": "1", "ISX_FLOOD_COLD_MIN": "100000000", "ISX_FLOOD_SMALL_MAX": "100000000"},
    "small-cold": {"ISX_FLOOD_COOP_MIN": "1", "ISX_FLOOD_COLD_MIN": "1", "ISX_FLOOD_SMALL_MAX": "100000000"},
    "small-cold-tiny": {"ISX_FLOOD_COOP_MIN": "1", "ISX_FLOOD_COLD_MIN": "1", "ISX_FLOOD_SMALL_MAX": "100000000", "ISX_FLOOD_RUN_LIMIT": "8",
                        "ISX_FLOOD_DRAIN": "4"},
    "small-win-one": {"ISX_FLOOD_COOP_MIN": "1", "ISX_FLOOD_COLD_MIN": "1", "ISX_FLOOD_SMALL_MAX": "100000000", "ISX_FLOOD_WIN": "1"},
    "small-win-tiny": {"ISX_FLOOD_COOP_MIN": "1", "ISX_FLOOD_COLD_MIN": "1", "ISX_FLOOD_SMALL_MAX": "100000000", "ISX_FLOOD_WIN": "3",
                       "ISX_FLOOD_LIST": "6", "ISX_FLOOD_RUN_LIMIT": "8", "ISX_FLOOD_DRAIN": "4"},
    "small-shortlist": {"ISX_FLOOD_COOP_MIN": "1", "ISX_FLOOD_COLD_MIN": "100000000", "ISX_FLOOD_SMALL_MAX": "100000000", "ISX_FLOOD_LIST": "8"},
}


@pytest.fixture(params=list(FLOOD_ENVS))
def flood_path(request, flood_knobs):
    flood_knobs(FLOOD_ENVS[request.param])
    return request.param


@pytest.mark.parametrize("name", KERNEL_CASES + ["ws_blobs40", "ws_blobs32"])
def test_flood_paths_bit_exact_on_fixtures(ws, golden, name, flood_path):
    d = golden(name)
    ref = d["labels_padded"] if "labels_padded" in d.files else np.pad(d["labels"], 1)
    scale = d["scale"] if "scale" in d.files else None
    out = np.zeros(ref.size, ref.dtype)
    lab = ws.affinity_watershed(d["aff_padded"], d["seeds_padded"], d["mask_padded"], scale=scale, out=out)
    assert np.array_equal(lab, ref)


@pytest.mark.parametrize("kind,shape,seed", [("smooth", (24, 96, 80), 0), ("smooth3", (32, 128, 128), 1),
                                              ("blobs", (40, 128, 128), 2)])
def test_flood_paths_against_oracle(ws, kind, shape, seed, flood_path):
    """Chain-heavy (smooth) and tie-heavy (blobs) inputs with components of 10^4..10^5 voxels: both
    flood paths must reproduce the oracle bit for bit (the LDS run overflows on the big ones, which
    exercises the hand-over to the serial heap too)."""
    from oracle import watershed as ows
    from tests.synth import blob_feature_map, smooth_feature_map
    fm = blob_feature_map(shape, seed) if kind == "blobs" else smooth_feature_map(shape, seed, 3.0 if kind == "smooth3" else 1.5)
    pshape = tuple(s + 2 for s in shape)
    ref = np.zeros(pshape, np.uint32)
    ows.segment_output_image(fm, out=ref.ravel())
    cur = np.zeros(pshape, np.uint32)
    ws.segment_output_image(fm, (0, 1, 2), 4, 3, out=cur.ravel())
    assert np.array_equal(cur, ref)
    assert (cur > 0).sum() > 1000


@pytest.mark.parametrize("world", [2, 3, 8])
def test_component_sharded_flood_union_is_exact(ws, world):
    """Multi-GPU flood (isx_ws_set_shard): every rank floods its own range of mask components; the element-wise
    max of the rank outputs is the unsharded result, supports are disjoint apart from the seed voxels."""
    import torch
    from tests.synth import blob_feature_map
    fm = torch.from_numpy(blob_feature_map((24, 64, 64), 5, 0.2, 2.5)).cuda()
    ctx = ws.get_context()
    pshape = tuple(s + 2 for s in fm.shape[1:])
    full = torch.zeros(pshape, dtype=torch.int32, device="cuda")
    _, seeds, _ = ws.segment_output_image(fm, (0, 1, 2), 4, 3, out=full)
    union = torch.zeros_like(full)
    nonseed_hits = torch.zeros(pshape, dtype=torch.int32, device="cuda")
    sz, sy, sx = (seeds + 1).T
    try:
        for r in range(world):
            ctx.set_flood_shard(r, world)
            part = torch.zeros_like(full)
            ws.segment_output_image(fm, (0, 1, 2), 4, 3, out=part, ctx=ctx)
            assert torch.equal(part[sz, sy, sx], full[sz, sy, sx])        # seeds labelled on every rank
            hit = (part > 0).to(torch.int32)
            hit[sz, sy, sx] = 0
            nonseed_hits += hit
            union = torch.maximum(union, part)
    finally:
        ctx.set_flood_shard(0, 1)
    assert torch.equal(union, full)
    assert int(nonseed_hits.max()) == 1 and len(seeds) > world            # disjoint supports
    with pytest.raises(ValueError):
        ctx.set_flood_shard(2, 2)


def test_connected_components_helper(ws):
    """ndi.label semantics: roots are the smallest raveled index, sizes the voxel counts."""
    from scipy import ndimage as ndi
    rng = np.random.default_rng(9)
    mask = ndi.gaussian_filter(rng.random((12, 40, 44)), 1.5) > 0.5
    roots, sizes = ws.connected_components(mask)
    roots, sizes = np.asarray(roots.cpu() if hasattr(roots, "cpu") else roots), np.asarray(sizes.cpu() if hasattr(sizes, "cpu") else sizes)
    lab, n = ndi.label(mask)
    assert len(np.unique(roots[mask])) == n
    for c in range(1, min(n, 20) + 1):
        idx = np.flatnonzero(lab.ravel() == c)
        assert (roots.ravel()[idx] == idx.min()).all() and sizes.ravel()[idx.min()] == idx.size


@pytest.mark.parametrize("shape", [(12, 40, 44), (20, 37, 70), (9, 18, 15)])
def test_component_pass_variants_agree(ws, flood_knobs, shape):
    """The labelling's passes in their other forms: flatten + sizes as two kernels (ISX_CCL_FUSED=0; default: one walk finds, stores and
    counts), and the seam pass's 16-voxel form against its voxel-per-thread form (taken for a mask that is not 16-byte aligned or has
    rows shorter than 16 voxels) -- roots and sizes identical, and ndi.label's."""
    import torch
    from scipy import ndimage as ndi
    rng = np.random.default_rng(11)
    mask = ndi.gaussian_filter(rng.random(shape), 1.2) > 0.5
    n = int(np.prod(shape))
    buf = torch.zeros(n + 1, dtype=torch.uint8, device="cuda")
    buf[1:] = torch.from_numpy(mask.ravel().view(np.uint8)).cuda()
    unaligned = buf[1:].view(shape)
    assert unaligned.data_ptr() % 16 != 0
    res = []
    for env in ({"ISX_CCL_FUSED": "1"}, {"ISX_CCL_FUSED": "0"}):
        flood_knobs(env)
        for m in (mask, unaligned):
            r, s = ws.connected_components(m)
            res.append((np.asarray(r.cpu() if hasattr(r, "cpu") else r).view(np.uint32), np.asarray(s.cpu() if hasattr(s, "cpu") else s).view(np.uint32)))
    for r, s in res[1:]:
        assert np.array_equal(r, res[0][0]) and np.array_equal(s, res[0][1])
    lab, ncomp = ndi.label(mask)
    roots, sizes = res[0]
    assert len(np.unique(roots[mask])) == ncomp and (roots[~mask] == 0xFFFFFFFF).all()
    for c in range(1, min(ncomp, 30) + 1):
        idx = np.flatnonzero(lab.ravel() == c)
        assert (roots.ravel()[idx] == idx.min()).all() and sizes.ravel()[idx.min()] == idx.size


@pytest.mark.parametrize("scale,abs_thresh", [((2.0, -1.0, 0.5), None), (0.25, 0.5)])
def test_segment_output_image_scale_and_absolute_thresh(ws, scale, abs_thresh):
    """segment_output_image's general form (watershed.py:165-223 with `scale=` / `absolute_thresh=`): one library call
    (isx_segment_output_image_scaled), checked against the oracle's stages composed the way the reference composes them."""
    from oracle import watershed as ows
    from tests.synth import smooth_feature_map
    fm = smooth_feature_map((12, 40, 36), 21)
    seg, seeds, mask = ws.segment_output_image(fm, (0, 1, 2), 4, 3, scale=scale, absolute_thresh=abs_thresh)
    aff = fm[:3] / fm[:3].max(axis=(1, 2, 3)).reshape(-1, 1, 1, 1)
    aff = np.pad(aff, ((0, 0), (1, 1), (1, 1), (1, 1)))
    cents = ows._get_centroids(fm[4]) + 1
    m = ows._get_mask(fm[3]) if abs_thresh is None else fm[3] > np.float32(abs_thresh)
    m, cents = ows._remove_unwanted_objects(np.pad(m, 1), cents, min_area=10, max_area=10000000)
    sc = np.asarray(scale, np.float32).reshape(-1, 1) if np.ndim(scale) else scale
    ref = ows.affinity_watershed(np.ascontiguousarray(aff, np.float32), cents, m, scale=sc)
    assert np.array_equal(mask, m)
    assert np.array_equal(np.asarray(seg).astype(np.int64), ref.reshape(m.shape)[1:-1, 1:-1, 1:-1].astype(np.int64))
    assert np.asarray(seg).max() > 0


@pytest.mark.parametrize("shift", [0.0, -0.3, -2.0])
def test_derived_affinity_range_is_the_measured_one(ws, flood_knobs, shift):
    """The flood's bucket map takes the value range of the padded, normalised affinities from the channel extrema
    (`aff_range_kernel`: x -> x / max is monotone) instead of reading all of them again.  ISX_WS_CHECK=1 makes the library measure the
    range as well and fail on any difference; with a channel whose maximum is not positive (shift -2) the derivation declares itself
    invalid and the measured range is used.  Labels as without the check."""
    from tests.synth import smooth_feature_map
    fm = smooth_feature_map((12, 40, 36), 23)
    fm[:3] += np.float32(shift)
    a = ws.segment_output_image(fm, (0, 1, 2), 4, 3)
    flood_knobs({"ISX_WS_CHECK": "1"})
    b = ws.segment_output_image(fm, (0, 1, 2), 4, 3)
    assert np.array_equal(np.asarray(a[0]), np.asarray(b[0]))


def test_gaussian_refuses_to_run_in_place(ws):
    """A Gaussian pass reads neighbours that other work-groups of the same pass write: isx_ws_gaussian with img == out is an
    argument error (include/itersegx.h), not a silent race."""
    import ctypes as C
    import torch
    from iterseg_amd import _lib
    from iterseg_amd.watershed import get_context
    ctx = get_context(0)
    t = torch.rand((4, 32, 64), device="cuda")
    sig = (C.c_int32 * 3)(0, 1, 1)
    rc = ctx.lib.isx_ws_gaussian(ctx.handle, C.c_void_p(t.data_ptr()), C.c_void_p(t.data_ptr()), _lib._dims(t.shape), sig)
    assert rc == _lib.ISX_ERR_ARG


@pytest.mark.parametrize("sigma", [2, (0, 1, 1), (0, 2, 2), 1])
def test_fused_gaussian_is_the_two_pass_one_bit_for_bit(ws, flood_knobs, sigma):
    """ISX_GAUSS_FUSED (default 1): the z pass as a march over a register ring of planes (`gauss_z_kernel`, several z segments
    on the tall shapes) and the y and the x pass of a plane in one kernel with a float32 intermediate in LDS
    (`gauss_yx_kernel`) against one kernel per axis -- scipy's axis order and per-pass float32 storage either way; shapes with
    partial tiles, single rows / columns and planes narrower than the filter (everything clamps, mode 'nearest'), and
    scipy itself on a small one."""
    from scipy import ndimage as ndi
    rng = np.random.default_rng(5)
    sig = (sigma,) * 3 if np.isscalar(sigma) else sigma
    for shape in [(9, 70, 130), (3, 16, 64), (5, 17, 65), (4, 5, 3), (2, 1, 200), (6, 33, 1), (150, 9, 300), (65, 3, 20)]:
        x = rng.random(shape, dtype=np.float32)
        flood_knobs({"ISX_GAUSS_FUSED": "1"})
        a = ws.gaussian(x, sig)
        flood_knobs({"ISX_GAUSS_FUSED": "0"})
        b = ws.gaussian(x, sig)
        assert np.array_equal(a, b), (shape, np.abs(a - b).max())
        ref = ndi.gaussian_filter(x, sig, mode="nearest")
        assert np.array_equal(a, ref), (shape, np.abs(a - ref).max())
