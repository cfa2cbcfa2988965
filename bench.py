#!/usr/bin/env python3
"""Headline benchmark: voxels/s end-to-end (chunked affinity U-Net + watershed) on a synthetic
512^3 volume, one frame per GPU (frames of a stack are independent in the reference:
src/iterseg/segmentation.py:873-882), inputs resident in HBM when the timed region starts.

    python bench.py --gpus N --steps K --warmup W

N > 1 is launched by the driver with torch.distributed.run (one rank per GPU); every rank segments
its own frame (weak scaling, no data-path collective), the timed region is bracketed by a barrier +
synchronize on both sides and the maximum over ranks is reported.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_MFMA_F32 = 157.3      # TFLOP/s, fp32 matrix (MI355X_MICROARCH.md, chip-level parameters)
PEAK_MFMA_F16 = 2500.0     # TFLOP/s, dense f16 matrix
PEAK_HBM = 8000.0          # GB/s


def conv_flops_per_chunk(chunk, out_channels=5):
    """Algorithmic FLOPs of one chunk forward, split into the MFMA-conv layers and the direct ones."""
    D, H, W = chunk
    lv = [(D, H, W)]
    for k in range(1, 5):
        d, h, w = lv[-1]
        lv.append((d // 2 if k == 4 else d, h // 2 + 1, w // 2 + 1))
    mods = [(1, 32, 0), (32, 64, 1), (64, 128, 2), (128, 256, 3), (256, 256, 4), (512, 128, 3), (256, 64, 2),
            (128, 32, 1), (64, out_channels, 0)]
    mfma = direct = 0.0
    for cin, cout, l in mods:
        m = lv[l][0] * lv[l][1] * lv[l][2]
        for ci in (cin, cout):
            f = 2.0 * m * cout * ci * 27
            if ci >= 32 and (cout % 32 == 0 or cout <= 5):      # incl. the folded tail layer (64 -> 5)
                mfma += f
            else:
                direct += f
    return mfma, direct


def cpu_baseline(sd, chunk, margin):
    """The oracle (CPU restatement) timed on a bounded sample of the same workload."""
    from oracle import unet_torch as ou, watershed as ows
    from tests.synth import positive_volume
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    shape = (18, chunk[1], chunk[2])                     # 2 chunks in z, one in y/x
    vol = positive_volume(shape, seed=123)
    t0 = time.perf_counter()
    fm = ou.predict_volume(sd, vol, chunk, margin)
    t1 = time.perf_counter()
    lab = np.zeros(tuple(s + 2 for s in shape), np.uint32)
    try:
        ows.segment_output_image(fm, out=lab.ravel())
    except ValueError:
        pass
    t2 = time.perf_counter()
    nvox = float(np.prod(shape))
    return {"value": nvox / (t2 - t0), "unit": "voxels/s", "cores": cores, "kind": "port",
            "sample": f"{shape[0]}x{shape[1]}x{shape[2]} volume = 2 chunks {chunk} through the torch-CPU U-Net oracle "
                      f"({t1 - t0:.2f} s, {cores} threads) + C watershed oracle ({t2 - t1:.2f} s, 1 thread); "
                      f"chunk over-compute 1.11x here vs 2.81x at 512^3"}


DTYPE = {"fp32": "f32 (v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulate)",
         "split-f16": "f32 operands split into two f16 halves each (x = x_h + x_l / 2^11), three v_mfma_f32_32x32x16_f16 per K "
                      "block with separate fp32 accumulators: fp32-class results on the f16 matrix cores"}
# L2-miss bytes per chunk-launch of the wide-conv kernels: FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE from separate
# rocprofv3 --pmc passes over one 64-chunk launch set, 16 launches (profiles/r01g_pmc_*, r01i_pmc_*)
TRAFFIC_PER_CHUNK_LAUNCH = {"fp32": 0.1054e9, "split-f16": 0.0848e9}
TRAFFIC_ALGORITHMIC = 0.05725e9          # every activation read once + written once + weights, per chunk-launch


def conv_roofline(precision, mfma_flops, n_chunks, steps, ms_mfma, n_mfma, batch):
    peak = PEAK_MFMA_F32 if precision == "fp32" else PEAK_MFMA_F16
    ach = (mfma_flops * n_chunks * steps) / (ms_mfma * 1e-3) / 1e12 if ms_mfma > 0 else 0.0
    r = {"bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
         "traffic": TRAFFIC_PER_CHUNK_LAUNCH[precision] * batch, "traffic_algorithmic": TRAFFIC_ALGORITHMIC * batch,
         "kernel": ("conv_mfma_kernel (fp32 32x32x2 implicit-GEMM conv, 16 of 18 layers)" if precision == "fp32" else
                    "conv_h2_kernel (split-f16 32x32x16 implicit-GEMM conv, 16 of 18 layers)"),
         "avg_launch_ms": ms_mfma / max(n_mfma, 1), "launches": n_mfma, "algorithmic_gflop_per_chunk": mfma_flops / 1e9}
    if precision != "fp32":
        r["note"] = ("achieved = algorithmic (fp32-equivalent) FLOPs; the split executes 3 f16 MFMAs per product (x 28/27 tap "
                     "padding), so the f16 pipe runs at about 3.1x this fraction; against the fp32 matrix peak (157.3) the same "
                     "work rate is %.2f" % (ach / PEAK_MFMA_F32))
    return r


def other_precision_leg(u, other, vol, fm, labels, chunk, margin, args, n_chunks, mfma_flops):
    """The same workload in the other arithmetic mode, reported beside the headline (never as `value`), and how far
    the two feature maps are apart on the first 64 planes of the volume."""
    from iterseg_amd import segmentation
    dev = vol.device
    main = u.precision
    sub = vol[:64].contiguous()
    a = u.predict_volume(sub, chunk, margin).clone()
    u.set_precision(other)
    b = u.predict_volume(sub, chunk, margin)
    diff = float((a - b).abs().max().item())
    del a, b

    def step():
        labels.zero_()
        try:
            segmentation.segment_volume_device(u, vol, chunk, margin, feature_map=fm, labels=labels)
        except ValueError:
            pass
    step()
    u.ctx.prof_enable(False)
    u.ctx.prof_enable(True)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize(dev)
    el = time.perf_counter() - t0
    ms_mfma, n_mfma = u.ctx.prof_get("conv_mfma")
    n_inst = int(labels.max().item())
    u.set_precision(main)
    nvox = float(vol.numel())
    return {"dtype": DTYPE[other], "value": nvox * args.steps / el, "unit": "voxels/s", "ms_per_step": 1e3 * el / args.steps,
            "max_abs_diff_of_feature_maps": diff, "instances": n_inst,
            "roofline": conv_roofline(other, mfma_flops, n_chunks, args.steps, ms_mfma, n_mfma, args.batch)}


def train_step_leg(sd, patch=(10, 256, 256), steps=5):
    """The other entry point of the path (train_unet, train.py:25-183): one training step on one (10,256,256) patch --
    forward, BCE loss, backward, the reference's two Adam updates per sample -- in fp32 (training never uses the split
    arithmetic).  Reported beside the headline, never as `value`."""
    from iterseg_amd.train import Trainer
    from tests.synth import positive_volume
    t = Trainer(sd, 5, patch)
    x = torch.from_numpy(positive_volume(patch, 1)).cuda()
    y = torch.from_numpy((np.random.default_rng(1).random((5,) + patch) < 0.5).astype(np.float32)).cuda()
    for _ in range(2):
        t.train_step(x, y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss, _ = t.train_step(x, y)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    fwd, direct = conv_flops_per_chunk(patch)
    return {"patch": list(patch), "ms_per_step": 1e3 * dt, "voxels_per_s": float(np.prod(patch)) / dt, "loss": float(loss),
            "tflops_of_3x_forward": 3.0 * (fwd + direct) / dt / 1e12, "peak_tflops_fp32_matrix": PEAK_MFMA_F32,
            "note": "forward + BCE + backward (dgrad, wgrad, BatchNorm) + the reference's double Adam update, batch 1"}


def structured_watershed_leg(u, n=256, volume_fraction=0.12):
    """the watershed stage alone on a *structured* feature map -- blobs / distance-transform centroid
    channel / gradient affinities as in the reference's own demo (watershed.py:256-262; tests/synth.blob_feature_map) --
    where the mask falls into many components and the per-component parallelism of the flood is visible."""
    from iterseg_amd import watershed as ws
    from tests.synth import blob_feature_map
    shape = (n, n, n)
    fm = torch.from_numpy(blob_feature_map(shape, 0, volume_fraction, 3.0)).cuda()
    lab = torch.zeros(tuple(s + 2 for s in shape), dtype=torch.int32, device=fm.device)
    best = None
    for _ in range(2):
        lab.zero_()
        u.ctx.prof_enable(False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ws.segment_output_image(fm, (0, 1, 2), 4, 3, out=lab)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    nvox = float(np.prod(shape))
    return {"shape": list(shape), "volume_fraction": volume_fraction, "ms": 1e3 * best, "voxels_per_s": nvox / best,
            "instances": int(lab.max().item()), "labelled_voxels": int((lab > 0).sum().item()),
            "hbm_frac": (24.0 * nvox / best / 1e9) / PEAK_HBM}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=int, default=512, help="edge of the cubic volume (default: the 512^3 north star)")
    ap.add_argument("--shape", type=int, nargs=3, default=None, metavar=("Z", "Y", "X"),
                    help="non-cubic volume instead of --size, e.g. 64 512 512 (BASELINE.json configs[1])")
    ap.add_argument("--batch", type=int, default=64, help="chunks per launch set (1..64; ~0.6 GB of workspace each)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--precision", default=None, choices=["fp32", "split-f16"],
                    help="arithmetic of the wide convolutions (default: the library default, split-f16)")
    ap.add_argument("--no-split-f16", "--no-other-precision", dest="no_other", action="store_true",
                    help="skip the extra measurement in the other arithmetic mode (N = 1 only)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # Rehearsal knobs (never set by the driver): ISX_BENCH_BACKEND=gloo with ISX_BENCH_ONE_DEVICE=1 runs
    # the N > 1 control flow with every rank on cuda:0 of a one-GPU box (RCCL refuses two ranks per device).
    backend = os.environ.get("ISX_BENCH_BACKEND", "nccl")
    if os.environ.get("ISX_BENCH_ONE_DEVICE"):
        local_rank = 0
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (there is no CPU path)"
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    red_dev = dev if backend == "nccl" else torch.device("cpu")

    from iterseg_amd import segmentation, weights
    from iterseg_amd.unet import UNet
    from tests.synth import positive_volume

    chunk, margin = (10, 256, 256), (1, 64, 64)
    shape = tuple(args.shape) if args.shape else (args.size,) * 3
    shape_name = "x".join(str(v) for v in shape) if args.shape else f"{args.size}^3"
    sd = weights.generate_state_dict(0)
    u = UNet(1, 5, device=local_rank, batch=args.batch, precision=args.precision)
    precision = u.precision
    u.load_state_dict(sd)

    # one frame per rank, resident in HBM (seed = rank: frames of a stack differ)
    vol = torch.from_numpy(positive_volume(shape, seed=rank)).to(dev)
    fm = torch.empty((5,) + shape, dtype=torch.float32, device=dev)
    labels = torch.zeros(tuple(s + 2 for s in shape), dtype=torch.int32, device=dev)
    n_chunks = len(__import__("iterseg_amd.chunking", fromlist=["x"]).chunk_table(shape, chunk, margin))

    def step():
        labels.zero_()
        try:
            segmentation.segment_volume_device(u, vol, chunk, margin, feature_map=fm, labels=labels)
        except ValueError:
            pass            # zero surviving seeds on this frame: labels stay 0 (the reference raises)

    def sync_all():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    u.ctx.prof_enable(True)
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    elapsed = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    prof = {k: u.ctx.prof_get(k) for k in ("conv_mfma", "conv_direct", "elementwise", "ws_filters", "ws_ccl", "ws_flood")}
    n_labelled = int((labels > 0).sum().item())
    n_inst = int(labels.max().item())

    if rank == 0:
        nvox = float(np.prod(shape))
        mfma_flops, direct_flops = conv_flops_per_chunk(chunk)
        ms_mfma, n_mfma = prof["conv_mfma"]
        t_unet = sum(prof[k][0] for k in ("conv_mfma", "conv_direct", "elementwise")) / args.steps
        t_ws = sum(prof[k][0] for k in ("ws_filters", "ws_ccl", "ws_flood")) / args.steps
        out = {
            "metric": "voxels/sec end-to-end (U-Net + watershed)",
            "value": world * nvox * args.steps / elapsed,
            "unit": "voxels/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": DTYPE[precision],
            "data": "synthetic",
            "config": {"workload": f"{shape_name} float32 volume per GPU (one frame per rank), chunk (10,256,256), "
                                   f"precision {precision}, "
                                   f"margin (1,64,64) = {n_chunks} chunks, seeded random U-Net weights (seed 0), "
                                   f"train-mode BatchNorm per chunk, watershed min_area 10 / max_area 1e7",
                       "chunks_per_launch_set": args.batch, "labelled_voxels": n_labelled, "instances": n_inst},
            "roofline": conv_roofline(precision, mfma_flops, n_chunks, args.steps, ms_mfma, n_mfma, args.batch),
            "stages_ms_per_step": {k: v[0] / args.steps for k, v in prof.items()},
            "watershed_roofline": {"bound": "hbm", "achieved": 24.0 * nvox / (t_ws * 1e-3) / 1e9 if t_ws > 0 else 0.0,
                                   "peak": PEAK_HBM, "unit": "GB/s",
                                   "frac": (24.0 * nvox / (t_ws * 1e-3) / 1e9) / PEAK_HBM if t_ws > 0 else 0.0,
                                   "bytes_per_voxel": 24},
            "t_unet_ms": t_unet, "t_ws_ms": t_ws,
        }
        # combined fraction t_ideal / t_measured, t_ideal = FLOPs / peak_MFMA + 24 B * ZYX / peak_HBM
        peak_tf = PEAK_MFMA_F32 if precision == "fp32" else PEAK_MFMA_F16
        t_ideal = ((mfma_flops + direct_flops) * n_chunks / (peak_tf * 1e12) + 24.0 * nvox / (PEAK_HBM * 1e9)) * 1e3
        out["combined_roofline_frac"] = t_ideal / (1e3 * elapsed / args.steps)
        out["t_ideal_ms"] = t_ideal
        if world == 1 and not args.no_other:
            other = "fp32" if precision != "fp32" else "split-f16"
            out["other_precision"] = {other: other_precision_leg(u, other, vol, fm, labels, chunk, margin, args, n_chunks,
                                                                 mfma_flops)}
        if world == 1 and not args.no_cpu_baseline:
            out["watershed_structured"] = structured_watershed_leg(u)
            out["train_step"] = train_step_leg(sd)
            out["cpu_baseline"] = cpu_baseline(sd, chunk, margin)
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
