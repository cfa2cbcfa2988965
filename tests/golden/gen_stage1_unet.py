#!/usr/bin/env python3
"""Golden-vector generator, stage 1: chunker, U-Net forward/predict, training step.

Run ONLY in the build container (needs /root/reference, torch CPU):

    /usr/local/bin/python3 tests/golden/gen_stage1_unet.py

It imports the reference's ``unet.py`` / ``predict.py`` / ``custom_loss.py`` *unmodified*
(read-only, from /root/reference) behind import stubs for GUI/IO packages that the
called functions never execute (recipe: SURVEY.md Appendix B), feeds them the
build-owned seeded weights (iterseg_amd.weights.generate_state_dict) and writes small
``.npz``/``.json`` fixtures next to this script.  The reference itself never travels;
only these data files do.
"""
import importlib
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference/src/iterseg"

from iterseg_amd.weights import generate_state_dict, to_torch_state_dict  # noqa: E402


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def import_reference():
    ident = lambda *a, **k: (a[0] if a and callable(a[0]) else (lambda f: f))
    _stub("napari")
    _stub("napari.qt", thread_worker=ident)
    _stub("toolz")
    _stub("skimage.exposure", rescale_intensity=None)
    _stub("skimage", filters=_stub("skimage.filters"), morphology=_stub("skimage.morphology"),
          feature=_stub("skimage.feature"))
    _stub("skimage.morphology._util", _offsets_to_raveled_neighbors=None,
          _validate_connectivity=None)

    def jit(*a, **k):
        def deco(f):
            f.py_func = f
            return f
        return deco
    _stub("numba", jit=jit)
    pkg = types.ModuleType("iterseg")
    pkg.__path__ = [REF]
    sys.modules["iterseg"] = pkg
    _stub("iterseg.helpers", get_regex_images=None, LINE="-" * 10, write_log=lambda *a: None)
    predict = importlib.import_module("iterseg.predict")
    unet_mod = importlib.import_module("iterseg.unet")
    closs = importlib.import_module("iterseg.custom_loss")
    return predict, unet_mod, closs


def positive_volume(shape, seed):
    """SURVEY.md 8(d) synthetic input: uniform (0,1], float32, then /max."""
    rng = np.random.default_rng(seed)
    v = 1.0 - rng.random(shape, dtype=np.float32)      # (0, 1]
    return (v / v.max()).astype(np.float32)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    predict, unet_mod, closs = import_reference()
    sd_np = generate_state_dict(seed=0)
    u = unet_mod.UNet(in_channels=1, out_channels=5)

    # ---- schema ---------------------------------------------------------------
    ref_sd = u.state_dict()
    schema = [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in ref_sd.items()]
    with open(os.path.join(HERE, "unet_schema.json"), "w") as f:
        json.dump(schema, f, indent=0)
    u.load_state_dict(to_torch_state_dict(sd_np), strict=True)

    # ---- chunker --------------------------------------------------------------
    cases = []
    shapes = [(32, 256, 256), (64, 512, 512), (32, 512, 512), (512,) * 3, (1024,) * 3,
              (12, 300, 300), (10, 256, 256), (11, 257, 300)]
    for shp in shapes:
        st, cr = predict.make_chunks(shp, (10, 256, 256), (1, 64, 64))
        cases.append({"shape": shp, "chunk": (10, 256, 256), "margin": (1, 64, 64),
                      "starts": np.asarray(st, np.int64).tolist() if len(st) <= 80 else None,
                      "crops": np.asarray(cr, np.int64).tolist() if len(st) <= 80 else None,
                      "n": len(st),
                      "starts_sum": int(np.asarray(st, np.int64).sum()),
                      "crops_sum": int(np.asarray(cr, np.int64).sum()),
                      "per_axis_starts": [sorted(set(int(s[d]) for s in st)) for d in range(3)],
                      "per_axis_crops": [sorted(set((int(s[d]), int(c[d][0]), int(c[d][1]))
                                                    for s, c in zip(st, cr))) for d in range(3)]})
    for shp, chk, mrg in [((10, 48, 48), (4, 32, 32), (1, 8, 8)), ((9, 40, 56), (4, 32, 32), (1, 8, 8)),
                          ((4, 32, 32), (4, 32, 32), (1, 8, 8)), ((20, 64, 64), (6, 32, 32), 2)]:
        st, cr = predict.make_chunks(shp, chk, mrg)
        cases.append({"shape": shp, "chunk": chk, "margin": mrg,
                      "starts": np.asarray(st, np.int64).tolist(),
                      "crops": np.asarray(cr, np.int64).tolist(), "n": len(st),
                      "starts_sum": int(np.asarray(st, np.int64).sum()),
                      "crops_sum": int(np.asarray(cr, np.int64).sum())})
    with open(os.path.join(HERE, "make_chunks.json"), "w") as f:
        json.dump(cases, f)

    # ---- per-layer taps on a tiny chunk (1,1,4,32,32) ---------------------------
    taps = {}

    def grab(name):
        def hook(mod, inp, out):
            taps[name] = out.detach().numpy().copy()
        return hook

    def grab_in(name):
        def hook(mod, inp):
            taps[name] = inp[0].detach().numpy().copy()
        return hook
    hs = [u.c0.conv0.register_forward_hook(grab("c0.conv0")),
          u.c0.batch0.register_forward_hook(grab("c0.batch0")),
          u.c0.register_forward_hook(grab("c0")),
          u.d0.register_forward_hook(grab("d0")),
          u.c1.register_forward_hook(grab("c1")),
          u.d3.register_forward_hook(grab("d3")),
          u.c4.register_forward_hook(grab("c4")),
          u.up0.register_forward_hook(grab("up0")),
          u.c5_0.register_forward_pre_hook(grab_in("c5_0.in")),
          u.c5_0.register_forward_hook(grab("c5_0")),
          u.up3.register_forward_hook(grab("up3")),
          u.c8_0.register_forward_pre_hook(grab_in("c8_0.in")),
          u.c8_0.conv1.register_forward_hook(grab("c8_0.conv1")),
          u.c8_0.register_forward_hook(grab("c8_0"))]
    x = positive_volume((4, 32, 32), seed=1)
    y = u(torch.from_numpy(x[None, None]))          # train-mode BN, like the reference (F2)
    for h in hs:
        h.remove()
    taps["x"] = x
    taps["y"] = y.detach().numpy().copy()
    # side effect of train-mode BN on the running statistics (U2)
    post = u.state_dict()
    for k in ("c0.batch0.running_mean", "c0.batch0.running_var", "c0.batch0.num_batches_tracked",
              "c8_0.batch1.running_mean", "c8_0.batch1.running_var"):
        taps["post." + k] = post[k].numpy().copy()
    np.savez_compressed(os.path.join(HERE, "unet_taps_4x32x32.npz"), **taps)
    print("taps:", {k: v.shape for k, v in taps.items()})

    # ---- chunked predict (P2-P4) on (10,48,48), chunk (4,32,32), margin (1,8,8) --
    u.load_state_dict(to_torch_state_dict(sd_np), strict=True)   # reset running stats
    vol = positive_volume((10, 48, 48), seed=2)
    out = np.zeros((5, 10, 48, 48), np.float32)
    predict.process_chunks(vol, (4, 32, 32), out, (1, 8, 8), predict.predict_chunk_feature_map,
                           config={"unet": u, "output_volume": out})
    np.savez_compressed(os.path.join(HERE, "predict_10x48x48.npz"), volume=vol, feature_map=out,
                        chunk=np.array([4, 32, 32]), margin=np.array([1, 8, 8]))
    print("predict fm", out.shape, out.min(), out.max())

    # second predict case: ragged geometry (last chunk pinned, uneven crops)
    vol2 = positive_volume((9, 40, 56), seed=3)
    out2 = np.zeros((5, 9, 40, 56), np.float32)
    predict.process_chunks(vol2, (4, 32, 32), out2, (1, 8, 8), predict.predict_chunk_feature_map,
                           config={"unet": u, "output_volume": out2})
    np.savez_compressed(os.path.join(HERE, "predict_9x40x56.npz"), volume=vol2, feature_map=out2,
                        chunk=np.array([4, 32, 32]), margin=np.array([1, 8, 8]))

    # ---- training step (T3/T2) on a (1,1,4,32,32) patch -------------------------
    train = {}
    rng = np.random.default_rng(7)
    tx = positive_volume((4, 32, 32), seed=4)
    ty = (rng.random((5, 4, 32, 32)) < 0.5).astype(np.float32)
    train["x"], train["y"] = tx, ty
    watch = ["c0.conv0.weight", "c0.conv0.bias", "c0.batch0.weight", "c3.conv1.weight",
             "c8_0.conv1.bias", "c8_0.conv1.weight", "c8_0.batch1.bias", "up0.weight", "up3.bias",
             "c5_0.conv0.weight"]
    for double_step in (False, True):
        u.load_state_dict(to_torch_state_dict(sd_np), strict=True)
        opt = torch.optim.Adam(u.parameters(), lr=0.01)
        loss = torch.nn.BCELoss()
        xx = torch.from_numpy(tx[None, None])
        yy = torch.from_numpy(ty[None])
        opt.zero_grad()                        # train.py:349-353
        y_hat = u(xx.float())
        l = loss(y_hat, yy)
        l.backward()
        opt.step()
        if double_step:                        # train.py:310 (second step on the same grads)
            opt.step()
        tag = "double" if double_step else "single"
        params = dict(u.named_parameters())
        if not double_step:
            train["loss_bce"] = np.float64(l.item())
            train["y_hat"] = y_hat.detach().numpy().copy()
            for k in watch:
                g = params[k].grad.detach().numpy()
                train["grad." + k] = g.reshape(g.shape[0], -1)[:, :64].copy()
            train["gradnorm"] = np.array([float(p.grad.double().pow(2).sum()) for p in u.parameters()])
            # other losses on the same (y_hat, y) (T4)
            yh = y_hat.detach()
            train["loss_dice"] = np.float64(closs.DiceLoss()(yh, yy).item())
            train["loss_wbce"] = np.float64(closs.WeightedBCELoss(
                chan_weights=(1., 2., 2., 1., 0.5), device="cpu")(yh, yy).item())
            train["loss_mse"] = np.float64(torch.nn.MSELoss()(yh, yy).item())
            chan = {}
            closs.channel_losses_to_dict(yh, yy, ("a", "b", "c", "d", "e"), loss,
                                         {k: chan.setdefault(k, []) for k in "abcde"})
            train["loss_bce_per_channel"] = np.array([chan[k][0] for k in "abcde"])
        for k in watch:
            w = params[k].detach().numpy()
            train[f"post_{tag}." + k] = w.reshape(w.shape[0], -1)[:, :64].copy()
    np.savez_compressed(os.path.join(HERE, "train_step_4x32x32.npz"), **train)
    print("train loss", train["loss_bce"], train["loss_dice"], train["loss_wbce"])


if __name__ == "__main__":
    main()
