"""Pins the U-Net/chunker/training oracle (oracle/unet_torch.py) and the host-side chunk
geometry + weight schema against golden vectors captured from the reference's own modules
(tests/golden/gen_stage1_unet.py).  CPU only."""
import json
import os

import numpy as np
import pytest
import torch

from iterseg_amd import chunking, weights
from oracle import unet_torch as ou

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
FWD_TOL = 2e-5      # same ATen kernels, different call path / thread partition


def test_schema_matches_reference_state_dict():
    with open(os.path.join(GOLDEN, "unet_schema.json")) as f:
        ref = json.load(f)
    mine = weights.unet_schema(1, 5)
    assert [k for k, _, _ in ref] == list(mine.keys())
    for k, shape, _ in ref:
        assert tuple(shape) == tuple(mine[k])
    sd = weights.generate_state_dict(0)
    assert weights.check_state_dict(sd)
    n_param = sum(v.size for k, v in sd.items() if "running" not in k and "num_batches" not in k)
    assert n_param == 9972673


def test_weight_generator_is_deterministic_and_order_free():
    a, b = weights.generate_state_dict(3), weights.generate_state_dict(3)
    assert all(np.array_equal(a[k], b[k]) for k in a)
    c = weights.generate_state_dict(4)
    assert not np.array_equal(a["c0.conv0.weight"], c["c0.conv0.weight"])


def _cases():
    with open(os.path.join(GOLDEN, "make_chunks.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("case", _cases(), ids=lambda c: "x".join(map(str, c["shape"])))
@pytest.mark.parametrize("impl", [chunking.make_chunks, ou.make_chunks], ids=["host", "oracle"])
def test_make_chunks(case, impl):
    st, cr = impl(tuple(case["shape"]), tuple(case["chunk"]), case["margin"] if isinstance(case["margin"], int)
                  else tuple(case["margin"]))
    st = np.asarray(st, np.int64)
    cr = np.asarray(cr, np.int64)
    assert len(st) == case["n"]
    assert int(st.sum()) == case["starts_sum"] and int(cr.sum()) == case["crops_sum"]
    if case.get("starts") is not None:
        assert st.tolist() == case["starts"] and cr.tolist() == case["crops"]
    if "per_axis_starts" in case:
        for d in range(3):
            assert sorted(set(int(s) for s in st[:, d])) == case["per_axis_starts"][d]


def test_chunks_tile_the_volume_exactly_once():
    """Size-independent property at the full BASELINE sizes: crops partition the volume."""
    for shape in [(512, 512, 512), (64, 512, 512), (11, 257, 300)]:
        tab = chunking.chunk_table(shape, (10, 256, 256), (1, 64, 64))
        for d in range(3):
            seen = np.zeros(shape[d], np.int32)
            for s, lo, hi in {(int(r[d]), int(r[3 + 2 * d]), int(r[4 + 2 * d])) for r in tab}:
                seen[s + lo:s + hi] += 1
            assert (seen == 1).all()
    assert len(chunking.chunk_table((512,) * 3, (10, 256, 256), (1, 64, 64))) == 576


def test_make_chunks_rejects_small_volume():
    with pytest.raises(ValueError):
        chunking.make_chunks((8, 256, 256), (10, 256, 256), (1, 64, 64))


def test_shard_range_partitions():
    for n, w in [(576, 8), (16, 8), (7, 3), (3, 8)]:
        spans = [chunking.shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))


@pytest.fixture(scope="module")
def sd():
    return weights.generate_state_dict(0)


def test_forward_taps(sd, golden):
    d = golden("unet_taps_4x32x32")
    taps = {}
    p = ou._as_params(sd)
    with torch.no_grad():
        y = ou.unet_forward(p, torch.from_numpy(d["x"])[None, None], taps=taps)
    for k in ["c0.conv0", "c0.batch0", "c0", "d0", "c1", "d3", "c4", "up0", "c5_0.in", "c5_0", "up3",
              "c8_0.in", "c8_0.conv1", "c8_0"]:
        ref = d[k]
        err = np.abs(taps[k].numpy() - ref).max() / max(1.0, np.abs(ref).max())
        assert err < FWD_TOL, (k, err)
    assert np.abs(y.numpy() - d["y"]).max() < FWD_TOL
    # train-mode BN side effect on running statistics
    for k in ("c0.batch0.running_mean", "c0.batch0.running_var", "c8_0.batch1.running_mean",
              "c8_0.batch1.running_var"):
        assert np.allclose(p[k].numpy(), d["post." + k], rtol=1e-5, atol=1e-6), k
    assert int(p["c0.batch0.num_batches_tracked"]) == int(d["post.c0.batch0.num_batches_tracked"]) == 1


@pytest.mark.parametrize("tag", ["10x48x48", "9x40x56"])
def test_chunked_predict(sd, golden, tag):
    d = golden("predict_" + tag)
    fm = ou.predict_volume(sd, d["volume"], tuple(d["chunk"]), tuple(d["margin"]))
    assert fm.shape == d["feature_map"].shape
    assert np.abs(fm - d["feature_map"]).max() < FWD_TOL


def test_train_step_single_and_double(sd, golden):
    d = golden("train_step_4x32x32")
    for double in (False, True):
        loss, y_hat, grads, new = ou.train_step(sd, d["x"], d["y"], lr=0.01, double_step=double)
        tag = "double" if double else "single"
        assert abs(loss - float(d["loss_bce"])) < 1e-6
        for k in [f[5:] for f in d.files if f.startswith("grad.")]:
            g = grads[k].reshape(grads[k].shape[0], -1)[:, :64]
            ref = d["grad." + k]
            assert np.abs(g - ref).max() <= 1e-5 + 1e-3 * np.abs(ref).max(), k
            w = new[k].reshape(new[k].shape[0], -1)[:, :64]
            # Adam's first step is lr*g/(|g|+1e-8).  A conv bias in front of train-mode BN has a
            # mathematically zero gradient (float noise ~1e-9), so its update is noise in
            # [-lr, lr] in the reference too: compare only where the gradient is meaningful.
            sig = np.abs(ref) > 1e-5
            assert np.abs(w - d[f"post_{tag}." + k])[sig].max(initial=0) < 1e-4, k
            assert np.abs(w - d[f"post_{tag}." + k]).max() <= (2 if double else 1) * 0.01 * 2 + 1e-6, k
    yh = torch.from_numpy(d["y_hat"])
    yy = torch.from_numpy(d["y"])[None]
    assert abs(float(ou.loss_fn("DiceLoss", yh, yy)) - float(d["loss_dice"])) < 1e-6
    assert abs(float(ou.loss_fn("WeightedBCE", yh, yy, (1., 2., 2., 1., .5))) - float(d["loss_wbce"])) < 1e-6
    assert abs(float(ou.loss_fn("MSELoss", yh, yy)) - float(d["loss_mse"])) < 1e-6


def test_dice_and_channelwise_losses_match_reference_classes(golden):
    """Value, autograd gradient and the per-channel log of DiceLoss / ChannelwiseLoss against the reference's
    own classes (tests/golden/gen_stage1b_losses.py)."""
    d, g = golden("train_step_4x32x32"), golden("losses_4x32x32")
    yy = torch.from_numpy(d["y"])[None]
    cases = {"dice": ("DiceLoss", None),
             "channelwise": ("Channelwise", (["BCELoss", "MSELoss"], [slice(0, 4), 4])),
             "channelwise3": ("Channelwise", (["MSELoss", "BCELoss", "BCELoss"], [slice(0, 3), 3, 4]))}
    for tag, (name, arg) in cases.items():
        yh = torch.from_numpy(d["y_hat"]).clone().requires_grad_(True)
        l = ou.loss_fn(name, yh, yy, arg)
        l.backward()
        assert abs(float(l) - float(g[f"{tag}.loss"])) < 1e-6, tag
        assert np.abs(yh.grad.numpy() - g[f"{tag}.grad"]).max() <= 1e-9 + 1e-5 * np.abs(g[f"{tag}.grad"]).max(), tag
    log = ou.channel_log("DiceLoss", torch.from_numpy(d["y_hat"]), yy)
    assert np.allclose(log, g["dice.per_channel_log"], atol=1e-6)


def test_forked_network_matches_reference(golden):
    """UNet(out_channels=(3, 2)) (unet.py:152-157,313-362): forward, loss, gradients (also of the shared encoder and
    transposed convolutions, which receive both decoders' contributions) and the Adam update against the fixture
    captured from the reference (tests/golden/gen_stage1c_forked.py)."""
    from iterseg_amd import weights
    d = golden("forked_step_4x32x32")
    forks = tuple(int(c) for c in d["forks"])
    sd = weights.generate_state_dict(3, 1, forks)
    assert list(sd.keys()) == [str(k) for k in d["keys"]]
    for double in (False, True):
        loss, y_hat, grads, new = ou.train_step(sd, d["x"], d["y"], lr=0.01, double_step=double)
        tag = "double" if double else "single"
        assert abs(loss - float(d["loss_bce"])) < 1e-6
        assert np.abs(y_hat - d["y_hat"]).max() < FWD_TOL
        for k in [f[5:] for f in d.files if f.startswith("grad.")]:
            g = grads[k].reshape(grads[k].shape[0], -1)[:, :64]
            ref = d["grad." + k]
            assert np.abs(g - ref).max() <= 1e-5 + 1e-3 * np.abs(ref).max(), k
            w = new[k].reshape(new[k].shape[0], -1)[:, :64]
            sig = np.abs(ref) > 1e-5
            assert np.abs(w - d[f"param_{tag}." + k])[sig].max(initial=0) < 1e-4, k
