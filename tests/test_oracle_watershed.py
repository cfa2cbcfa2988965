"""Pins the CPU oracle (oracle/ws_oracle.c) against the golden vectors captured from the
reference's own watershed.py (tests/golden/gen_stage2_watershed.py).  CPU only."""
import numpy as np
import pytest

from oracle import watershed as ows

FM_CASES = ["ws_fm_10x48x48", "ws_fm_9x40x56", "ws_stages_0", "ws_stages_2", "ws_stages_3"]
KERNEL_CASES = ["ws_k_ties", "ws_k_giant", "ws_k_noseed_comp", "ws_k_scale_axis", "ws_k_scale_neg",
                "ws_k_negvals"]


def _fm(golden, name):
    d = golden(name)
    if "feature_map" in d.files:
        return d, d["feature_map"]
    tag = name.split("ws_fm_")[1]
    return d, golden("predict_" + tag)["feature_map"]


def seeds_equal_mod_ties(got, ref, intensity_of):
    """Ordered equality, except that runs of equal intensity may be permuted (upstream argsort
    is unstable)."""
    if got.shape != ref.shape:
        return False
    if np.array_equal(got, ref):
        return True
    iv = intensity_of(ref)
    if not np.array_equal(intensity_of(got), iv):
        return False
    return set(map(tuple, got)) == set(map(tuple, ref))


@pytest.mark.parametrize("name", FM_CASES + ["ws_stages_1"])
def test_gaussian_and_otsu(golden, name):
    d, fm = _fm(golden, name)
    sm = ows.gaussian(fm[3], 2)
    assert np.array_equal(sm, d["mask_smooth"])                    # bit-exact float32
    assert ows.threshold_otsu(sm) == d["thresh"]
    assert np.array_equal(ows._get_mask(fm[3]), d["mask_raw"])
    assert np.array_equal(ows.gaussian(fm[4], (0, 1, 1)), d["cent_smooth"])


@pytest.mark.parametrize("name", FM_CASES + ["ws_stages_1"])
def test_centroids_and_filter(golden, name):
    d, fm = _fm(golden, name)
    g = d["cent_smooth"]
    pk = ows._get_centroids(fm[4])
    assert seeds_equal_mod_ties(pk, d["seeds_raw"], lambda c: g[tuple(c.T)])
    mp = np.pad(d["mask_raw"], 1, constant_values=0)
    m2, c2 = ows._remove_unwanted_objects(mp, d["seeds_raw"] + 1, min_area=10, max_area=10000000)
    assert np.array_equal(m2, d["mask_filtered_padded"])
    assert np.array_equal(c2, d["seeds_filtered_padded"])


@pytest.mark.parametrize("name", FM_CASES)
def test_segment_output_image_plugin_path(golden, name):
    d, fm = _fm(golden, name)
    pshape = tuple(s + 2 for s in fm.shape[1:])
    cur = np.zeros(pshape, np.uint32)
    seg, seeds, mask = ows.segment_output_image(fm, (0, 1, 2), thresholding_channel=3, centroids_channel=4,
                                                out=cur.ravel())
    ref = d["labels_padded"]
    # label ids follow seed order; identical unless equal-intensity seeds were permuted upstream
    if np.array_equal(seeds + 1, d["seeds_filtered_padded"]):
        assert np.array_equal(cur, ref)
    else:
        assert np.array_equal(cur > 0, ref > 0)
    assert np.array_equal(mask, d["mask_filtered_padded"])


def test_zero_seeds_raises_like_reference(golden):
    d = golden("ws_stages_1")
    assert str(d["error"]) == "ValueError" and len(d["seeds_filtered_padded"]) == 0
    with pytest.raises(ValueError):
        ows.segment_output_image(d["feature_map"])


@pytest.mark.parametrize("name", KERNEL_CASES)
def test_affinity_watershed_kernel_boundary(golden, name):
    d = golden(name)
    ref = d["labels_padded"]
    scale = d["scale"] if "scale" in d.files else None
    out = np.zeros(ref.size, ref.dtype)
    lab = ows.affinity_watershed(d["aff_padded"], d["seeds_padded"], d["mask_padded"], scale=scale, out=out)
    assert lab.dtype == ref.dtype
    assert np.array_equal(lab, ref)


def test_affinity_watershed_default_mask_and_dtype(golden):
    d = golden("ws_k_giant_nomask")
    lab = ows.affinity_watershed(d["aff_padded"], d["seeds_padded"], None)
    assert lab.dtype == np.int32 and np.array_equal(lab, d["labels_padded"])


@pytest.mark.parametrize("name", ["ws_blobs40", "ws_blobs32"])
def test_reference_synthetic_blobs(golden, name):
    """The reference's own self-check construction (watershed.py:256-262), absolute threshold."""
    d = golden(name)
    vol = d["volume"]
    assert np.array_equal(ows.gaussian(vol[3], (0, 1, 1)), d["cent_smooth"])
    g = d["cent_smooth"]
    pk = ows._get_centroids(vol[3])
    assert seeds_equal_mod_ties(pk, d["seeds_raw"], lambda c: g[tuple(c.T)])
    lab = ows.affinity_watershed(d["aff_padded"], d["seeds_padded"], d["mask_padded"])
    assert np.array_equal(lab[1:-1, 1:-1, 1:-1], d["labels"])


def test_size_filter_upper_bound(golden):
    d = golden("ws_sizefilter")
    m2, c2 = ows._remove_unwanted_objects(d["mask_padded"], d["seeds_padded"], int(d["min_area"]),
                                          int(d["max_area"]))
    assert np.array_equal(m2, d["mask_out"]) and np.array_equal(c2, d["seeds_out"])


def test_plateaus_and_trivial_images(golden):
    d = golden("ws_plateau")
    pk = ows._get_centroids(d["image"], gaussian=False)
    assert set(map(tuple, pk)) == set(map(tuple, d["peaks"])) and len(pk) == len(d["peaks"])
    assert tuple(pk[0]) == tuple(d["peaks"][0])                       # the unique global maximum is first
    assert len(ows._get_centroids(d["const"], gaussian=False)) == 0
    assert ows.threshold_otsu(d["const"]) == d["otsu_const"]


def test_flood_decomposes_over_components(golden):
    """flooding each mask component separately and merging equals the joint run."""
    from scipy import ndimage as ndi
    d = golden("ws_k_ties")
    aff, seeds, mask, ref = d["aff_padded"], d["seeds_padded"], d["mask_padded"], d["labels_padded"]
    cc, n = ndi.label(mask)
    merged = np.zeros_like(ref)
    for c in range(1, n + 1):
        sel = cc[tuple(seeds.T)] == c
        if not sel.any():
            continue
        out = np.zeros(ref.size, np.uint32)
        part = ows.affinity_watershed(aff, seeds[sel], cc == c, out=out)
        ids = np.flatnonzero(sel) + 1                                    # local label k -> global id
        merged[part > 0] = ids[part[part > 0] - 1]
    assert np.array_equal(merged, ref)


@pytest.mark.parametrize("name", ["ws_k_ties", "ws_k_noseed_comp", "ws_k_negvals", "ws_blobs40"])
def test_component_parallel_flood_equals_single_heap(golden, name):
    """The cpu_baseline of bench.py floods component by component on several threads; same labels as
    the single heap (and as the reference's fixture)."""
    d = golden(name)
    ref = d["labels_padded"] if "labels_padded" in d.files else np.pad(d["labels"], 1)
    for threads in (1, 4):
        lab = ows.affinity_watershed(d["aff_padded"], d["seeds_padded"], d["mask_padded"], threads=threads)
        assert np.array_equal(lab, ref)


def test_component_parallel_segment_output_image():
    from tests.synth import blob_feature_map
    fm = blob_feature_map((24, 64, 64), 3, 0.2, 2.5)
    a = np.zeros((26, 66, 66), np.uint32)
    b = np.zeros_like(a)
    ows.segment_output_image(fm, out=a.ravel())
    ows.segment_output_image(fm, out=b.ravel(), threads=4)
    assert np.array_equal(a, b) and a.max() > 3


def test_threaded_flood_with_off_mask_seed_equals_single_heap():
    """A marker OFF the mask is still popped by the reference loop and labels its in-mask neighbours (watershed.py:139-153);
    such a list does not decompose over components, so the threaded oracle must take the single heap for it."""
    from oracle import watershed as ows
    rng = np.random.default_rng(5)
    shape = (8, 12, 12)
    aff = np.pad(rng.random((3,) + tuple(s - 2 for s in shape), dtype=np.float32), ((0, 0), (1, 1), (1, 1), (1, 1)))
    mask = np.zeros(shape, bool)
    mask[1:-1, 1:-1, 1:5] = True
    mask[1:-1, 1:-1, 6:-1] = True                     # two components separated by the plane x = 5
    seeds = np.array([[3, 3, 2], [4, 6, 5], [3, 8, 8]])          # the middle one sits on the separating plane (off-mask)
    a = ows.affinity_watershed(aff, seeds, mask, threads=0)
    b = ows.affinity_watershed(aff, seeds, mask, threads=4)
    assert np.array_equal(a, b) and (a == 2).sum() > 1          # the off-mask seed labelled neighbours on both sides
