#!/usr/bin/env python3
"""Headline benchmark: voxels/s end-to-end (chunked affinity U-Net + watershed) on a synthetic
512^3 volume, one frame per GPU (frames of a stack are independent in the reference:
src/iterseg/segmentation.py:873-882), inputs resident in HBM when the timed region starts.

    python bench.py --gpus N --steps K --warmup W

N > 1 is launched by the driver with torch.distributed.run (one rank per GPU); every rank segments
its own frame (weak scaling, no data-path collective), the timed region is bracketed by a barrier +
synchronize on both sides and the maximum over ranks is reported.  Rank 0 prints ONE JSON line.

Beside the headline the same line carries (N = 1): the exact-fp32 arithmetic (`other_precision`), BASELINE
configs[1] end to end (`config2`: 64x512x512, where one 8.3 M-voxel component is NOT dropped by max_area and the
flood carries the frame), the watershed stage on a structured 512^3 input (`watershed_structured`), one training
step (`train_step`) and the CPU oracle on a bounded sample (`cpu_baseline`).  For N > 1 it also carries the
strong-scaling job (`strong_scaling`: ONE 512^3 volume on all ranks, feature-map slab all-gather + component-sharded
flood + label max-reduce inside the timed region) and BASELINE configs[2] (`config3_stack`: 16 frames of 32x512x512,
frame t on rank t mod N, label all-gather inside the timed region); `--mode strong|stack` makes one of them the
headline instead.
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")        # before the first HIP call: the flood's three streams need a queue each (iterseg_amd/__init__.py)

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
    """The oracle (CPU restatement) timed on a bounded sample of the same workload: a 26x512x512 slab = 27 chunks at the
    headline's chunk over-compute (z starts 0, 8, 16; 3 x 3 in y, x: 2.60x against 2.81x at 512^3), U-Net on all host
    threads, watershed with the flood component by component on the same threads."""
    from oracle import unet_torch as ou, watershed as ows
    from tests.synth import positive_volume
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    shape = (26, 2 * chunk[1], 2 * chunk[2])
    vol = positive_volume(shape, seed=123)
    t0 = time.perf_counter()
    fm = ou.predict_volume(sd, vol, chunk, margin)
    t1 = time.perf_counter()
    lab = np.zeros(tuple(s + 2 for s in shape), np.uint32)
    try:
        ows.segment_output_image(fm, out=lab.ravel(), threads=cores)
    except ValueError:
        pass
    t2 = time.perf_counter()
    nvox = float(np.prod(shape))
    n_chunks = len(__import__("iterseg_amd.chunking", fromlist=["x"]).chunk_table(shape, chunk, margin))
    over = n_chunks * float(np.prod(chunk)) / nvox
    return {"value": nvox / (t2 - t0), "unit": "voxels/s", "cores": cores, "kind": "port",
            "sample": f"{shape[0]}x{shape[1]}x{shape[2]} slab = {n_chunks} chunks {chunk} (over-compute {over:.2f}x; 2.81x at 512^3) "
                      f"through the torch-CPU U-Net oracle ({t1 - t0:.2f} s, {cores} threads) + C watershed oracle with the "
                      f"flood component-parallel on {cores} threads ({t2 - t1:.2f} s; {int((lab > 0).sum())} voxels labelled)"}


DTYPE = {"fp32": "f32 (v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulate)",
         "split-f16": "f32 operands split into two f16 halves each (x = x_h + x_r, operands pre-scaled by powers of two so that the residual "
                      "is a normal half), three v_mfma_f32_16x16x32_f16 per K block (a_h b_h + a_h b_r + a_r b_h) into one fp32 "
                      "accumulator: fp32-class results on the f16 matrix cores",
         # reduced-precision characterisation runs only (--precision f16 / bf16): never the default, never the headline
         "f16": "f16 operands (round to nearest), one v_mfma_f32_16x16x32_f16 per K block, fp32 accumulate -- REDUCED precision "
                "(BASELINE config 5's 'fp16 U-Net'; stated tolerance 4e-3 on the sigmoid maps)",
         "bf16": "bf16 operands, one v_mfma_f32_16x16x32_bf16 per K block, fp32 accumulate -- REDUCED precision (stated tolerance 3e-2)"}
# roofline.traffic: HBM bytes per launch of the wide-conv kernel from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes
# (separate passes, FETCH x2 on gfx950, MI355X_MICROARCH.md) over one launch set of THIS build's kernels.  bench.py
# cannot collect PMC counters on itself, so it reads the summary that scratch/pmc_traffic_json.py wrote from those
# passes (committed under profiles/ with the CSVs); without one, or if it was made for another kernel build, traffic
# is null.
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r05_pmc_traffic.json")
TRAFFIC_ALGORITHMIC = 0.05725e9          # every activation read once + written once + weights, per chunk-launch


def measured_traffic(precision, batch):
    try:
        d = json.load(open(TRAFFIC_FILE))
        e = d[precision]
        return float(e["bytes_per_chunk_launch"]) * batch, e.get("source")
    except Exception:
        return None, None


def conv_roofline(precision, mfma_flops, n_chunks, steps, ms_mfma, n_mfma, batch):
    peak = PEAK_MFMA_F32 if precision == "fp32" else PEAK_MFMA_F16
    ach = (mfma_flops * n_chunks * steps) / (ms_mfma * 1e-3) / 1e12 if ms_mfma > 0 else 0.0
    traffic, traffic_src = measured_traffic(precision, batch)
    r = {"bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
         "traffic": traffic, "traffic_source": traffic_src, "traffic_algorithmic": TRAFFIC_ALGORITHMIC * batch,
         "kernel": ("conv_mfma_kernel (fp32 32x32x2 implicit-GEMM conv, 16 of 18 layers)" if precision == "fp32" else
                    "conv_h2c16nw8m_kernel / conv_h2c16nw8x2_kernel / conv_h2c16_kernel / conv_tail_kernel (split-f16 implicit-GEMM conv on v_mfma_f32_16x16x32_f16: matrix phase software-pipelined, weight slabs by LDS-DMA, 512-voxel work-groups with 64 x 64 wave tiles for Cout = 64 and c2, two N tiles per 256-voxel work-group for Cout % 128 == 0 on the linearised levels; the thin tail layer in one pass with up3 up-sampled on read; 16 of 18 layers)"),
         "avg_launch_ms": ms_mfma / max(n_mfma, 1), "launches": n_mfma, "algorithmic_gflop_per_chunk": mfma_flops / 1e9}
    if precision in ("f16", "bf16"):
        r["kernel"] = f"conv_h2c16nw8m_kernel / conv_h2c16nw8x2_kernel / conv_h2c16_kernel / conv_tail_kernel ({precision} operands, one MFMA per product, 16 of 18 layers)"
        r["note"] = "reduced-precision mode: algorithmic FLOPs = executed MFMA FLOPs x 27/28"
    elif precision != "fp32":
        r["note"] = ("achieved = algorithmic (fp32-equivalent) FLOPs; the split executes 3 f16 MFMAs per product (x 28/27 tap "
                     "padding), so the f16 pipe runs at about 3.1x this fraction; against the fp32 matrix peak (157.3) the same "
                     "work rate is %.2f; measured with rocm-smi (profiles/r02_clock_power_during_bench.txt): the kernel runs at the "
                     "package power cap (~1300 W, shader clock 2.05-2.1 GHz instead of 2.4)" % (ach / PEAK_MFMA_F32))
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
        segmentation.segment_volume_device(u, vol, chunk, margin, feature_map=fm, labels=labels)
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


def train_step_leg(sd, patch=(10, 256, 256), steps=5, precision="fp32"):
    """The other entry point of the path (train_unet, train.py:25-183): one training step on one (10,256,256) patch --
    forward, BCE loss, backward, the reference's two Adam updates per sample -- in fp32 (training never uses the split
    arithmetic).  Reported beside the headline, never as `value`."""
    from iterseg_amd.train import Trainer
    from tests.synth import positive_volume
    t = Trainer(sd, 5, patch, precision=precision)
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
            "tflops_of_3x_forward": 3.0 * (fwd + direct) / dt / 1e12,
            # the peak of the pipe the step's convolutions run on: bf16 matrix cores (dense) / fp32 matrix cores
            "peak_tflops": PEAK_MFMA_F16 if precision == "bf16" else PEAK_MFMA_F32,
            "frac_of_peak": 3.0 * (fwd + direct) / dt / 1e12 / (PEAK_MFMA_F16 if precision == "bf16" else PEAK_MFMA_F32),
            "precision": precision,
            "note": "forward + BCE + backward (dgrad, wgrad, BatchNorm) + the reference's double Adam update, batch 1"
                    + ("; bf16 = forward, data-gradient and weight-gradient convolutions of the wide layers on the bf16 matrix "
                       "cores, fp32 master weights / BatchNorm / Adam (BASELINE config 4's arithmetic); ~160 launches per step: "
                       "launch- and tail-bound at batch 1" if precision == "bf16" else "")}


def structured_watershed_leg(u, n=512, volume_fraction=0.12, t_unet_ms=None):
    """the watershed stage alone on a *structured* feature map -- blobs / distance-transform centroid
    channel / gradient affinities as in the reference's own demo (watershed.py:256-262; tests/synth.blob_feature_map) --
    where the mask falls into many components of every size, which is what a trained network produces.  The n^3 input is
    eight independently seeded (n/2)^3 demo volumes side by side (tests/synth.blob_feature_map_tiled).
    `end_to_end_voxels_per_s` prices a whole frame with this watershed behind the measured U-Net stage."""
    from iterseg_amd import watershed as ws
    from tests.synth import blob_feature_map_tiled
    shape = (n, n, n)
    fm = torch.from_numpy(blob_feature_map_tiled(n, 0, volume_fraction, 3.0)).cuda()
    lab = torch.zeros(tuple(s + 2 for s in shape), dtype=torch.int32, device=fm.device)
    best, stages = None, None
    for _ in range(2):
        lab.zero_()
        u.ctx.prof_enable(False)
        wctx = ws.get_context(fm.device.index)
        wctx.prof_enable(False)
        wctx.prof_enable(True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _, _, mask = ws.segment_output_image(fm, (0, 1, 2), 4, 3, out=lab)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if best is None or dt < best:
            best, stages = dt, {k: wctx.prof_get(k)[0] for k in ("ws_filters", "ws_ccl", "ws_flood")}
    nvox = float(np.prod(shape))
    roots, sizes = ws.connected_components(mask)
    sz = torch.as_tensor(sizes).flatten()
    sz = torch.sort(sz[sz > 0], descending=True).values
    n_lab = int((lab > 0).sum().item())
    out = {"shape": list(shape), "volume_fraction": volume_fraction, "construction": "8 independently seeded (n/2)^3 demo volumes", "ms": 1e3 * best, "voxels_per_s": nvox / best,
           "stages_ms": stages, "instances": int(lab.max().item()), "labelled_voxels": n_lab,
           "components": int(sz.numel()), "largest_components": [int(v) for v in sz[:4].tolist()],
           "flood_pops_per_s": n_lab / max(stages["ws_flood"], 1e-9) * 1e3,
           "hbm_frac": (24.0 * nvox / best / 1e9) / PEAK_HBM}
    if t_unet_ms is not None:
        out["end_to_end_voxels_per_s"] = nvox / (t_unet_ms * 1e-3 + best)
        out["unet_voxels_per_s"] = nvox / (t_unet_ms * 1e-3)
    return out


def config2_leg(u, sd_seed_rank, chunk, margin, batch):
    """BASELINE.json configs[1]: one 64x512x512 volume end to end (72 chunks + watershed).  With the seeded random-weight
    network the foreground is ONE 8.3 M-voxel component below max_area, so unlike at 512^3 it is flooded, not dropped:
    the frame is the flood of a single chain-like component (the worst case for the speculative-window kernel)."""
    from iterseg_amd import segmentation
    from tests.synth import positive_volume
    shape = (64, 512, 512)
    dev = torch.device("cuda", u.device)
    vol = torch.from_numpy(positive_volume(shape, seed=sd_seed_rank)).to(dev)
    fm = torch.empty((5,) + shape, dtype=torch.float32, device=dev)
    labels = torch.zeros(tuple(s + 2 for s in shape), dtype=torch.int32, device=dev)
    best, prof = None, None
    for _ in range(2):
        labels.zero_()
        u.ctx.prof_enable(False)
        u.ctx.prof_enable(True)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        segmentation.segment_volume_device(u, vol, chunk, margin, feature_map=fm, labels=labels)
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        if best is None or dt < best:
            best = dt
            prof = {k: u.ctx.prof_get(k)[0] for k in ("conv_mfma", "conv_direct", "elementwise", "ws_filters", "ws_ccl", "ws_flood")}
    u.ctx.prof_enable(False)
    nvox = float(np.prod(shape))
    n_lab = int((labels > 0).sum().item())
    assert n_lab > 0
    return {"workload": "64x512x512 float32 volume, chunk (10,256,256), margin (1,64,64) = 72 chunks, same weights",
            "value": nvox / best, "unit": "voxels/s", "ms_per_step": 1e3 * best, "stages_ms": prof,
            "labelled_voxels": n_lab, "instances": int(labels.max().item()),
            "flood_pops_per_s": n_lab / max(prof["ws_flood"], 1e-9) * 1e3}


def plugin_path_leg(u, chunk, margin, device_ms_per_frame, shape=(512, 512, 512), frames=4):
    """The call the reference actually makes (segmentation.py:833-900): HOST float32 frames in, HOST labels out.
    `single_frame`: segment_single_volume(frame, ..., affinity_watershed_for_chunks) on one 512^3 host array -- the
    plugin's processing function behind the reference's own per-frame host code (min / max / divide / padded zeros),
    with the share of the time spent outside the device work.  `stack`: segmentation_loop over a (4,512,512,512) host
    stack into an int32 label store, frames pipelined through isx_pipe_* (upload of t+1 and download of t-1 under the
    compute of t); `vs_device_resident` compares its per-frame time with the headline's."""
    from iterseg_amd import segmentation
    from iterseg_amd._lib import NoSeedsError
    from tests.synth import positive_volume
    nvox = float(np.prod(shape))
    stack = np.stack([positive_volume(shape, seed=t) for t in range(frames)])
    config = {"unet": u, "output_volume": segmentation.a_w_output_volume(stack, 5)}
    out = {}
    # one frame through the plugin pair, as segment_single_volume calls it
    best = None
    for _ in range(2):
        frame = stack[0].copy()                               # segmentation_loop's astype copy (segmentation.py:877)
        u.ctx.prof_enable(False)
        u.ctx.prof_enable(True)
        t0 = time.perf_counter()
        try:
            labels = segmentation.segment_single_volume(frame, chunk, config, margin, segmentation.affinity_watershed_for_chunks)
        except NoSeedsError:
            labels = None
        dt = time.perf_counter() - t0
        dev_ms = sum(u.ctx.prof_get(k)[0] for k in ("conv_mfma", "conv_direct", "elementwise", "ws_filters", "ws_ccl", "ws_flood"))
        if best is None or dt < best[0]:
            best = (dt, dev_ms)
    u.ctx.prof_enable(False)
    out["single_frame"] = {"value": nvox / best[0], "unit": "voxels/s", "ms": 1e3 * best[0], "device_kernels_ms": best[1],
                           "host_and_pcie_share": 1.0 - best[1] * 1e-3 / best[0],
                           "labelled_voxels": int((labels > 0).sum()) if labels is not None else 0}
    # the stack through segmentation_loop (pipelined)
    store = np.zeros((frames,) + shape, np.int32)
    # warm-up like the headline's: one 2-frame pass allocates the pipe's slots (kept by the U-Net) and touches the store
    for _ in segmentation.segmentation_loop(None, stack[:2], chunk, margin, store[:2], segmentation.affinity_watershed_for_chunks, config):
        pass
    store[:] = 0
    t0 = time.perf_counter()
    done = [t for t in segmentation.segmentation_loop(None, stack, chunk, margin, store, segmentation.affinity_watershed_for_chunks, config)]
    dt = time.perf_counter() - t0
    assert done == list(range(frames))
    assert np.array_equal(store[0], labels.view(np.int32)) if labels is not None else True
    out["stack"] = {"frames": frames, "value": frames * nvox / dt, "unit": "voxels/s", "ms_per_frame": 1e3 * dt / frames,
                    "vs_device_resident": (1e3 * dt / frames) / device_ms_per_frame if device_ms_per_frame else None,
                    "labelled_voxels": int((store > 0).sum())}
    os.environ["ISX_FRAME_PIPE"] = "0"
    store[:] = 0
    t0 = time.perf_counter()
    for _ in segmentation.segmentation_loop(None, stack, chunk, margin, store, segmentation.affinity_watershed_for_chunks, config):
        pass
    dt0 = time.perf_counter() - t0
    os.environ["ISX_FRAME_PIPE"] = "1"
    out["stack_unpipelined_ms_per_frame"] = 1e3 * dt0 / frames
    out["workload"] = (f"host float32 frames {shape} -> host labels through the plugin pair; device-resident headline "
                       f"{device_ms_per_frame:.1f} ms/frame")
    return out


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
    ap.add_argument("--precision", default=None, choices=["fp32", "split-f16", "f16", "bf16"],
                    help="arithmetic of the wide convolutions (default: the library default, split-f16; f16 / bf16 are "
                         "reduced-precision characterisation runs, e.g. BASELINE config 5, and say so in `dtype`)")
    ap.add_argument("--no-split-f16", "--no-other-precision", dest="no_other", action="store_true",
                    help="skip the extra measurement in the other arithmetic mode (N = 1 only)")
    ap.add_argument("--mode", default="weak", choices=["weak", "strong", "stack"],
                    help="weak (default): one frame per rank; strong: ONE volume sharded over all ranks (chunk-sharded "
                         "U-Net, feature-map slab all-gather, component-sharded flood, label max-reduce); stack: BASELINE "
                         "configs[2], 16 frames of 32x512x512, frame t on rank t mod N, label all-gather in the timed region")
    ap.add_argument("--no-extra-legs", action="store_true", help="headline only (profiling runs)")
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

    from iterseg_amd._lib import NoSeedsError
    from iterseg_amd import distributed as isx_dist
    group = None

    def strong_step(volume, fmap, labs):
        """one volume on all ranks; `volume` is identical on every rank (seed 0)"""
        labs.zero_()
        try:
            isx_dist.segment_volume_sharded(u, volume, chunk, margin, group=group, feature_map=fmap, labels=labs)
        except NoSeedsError:
            pass

    stack_shape, stack_T = (32, 512, 512), 16

    def stack_step(frames, fmap, labs, gathered):
        """configs[2]: this rank's frames (t mod N), then ONE all-gather of the int32 label tiles"""
        mine = isx_dist.frames_for_rank(stack_T, rank, world)
        for i, t in enumerate(mine):
            labs.zero_()
            try:
                segmentation.segment_volume_device(u, frames[i], chunk, margin, feature_map=fmap, labels=labs)
            except NoSeedsError:
                pass
            gathered[rank, i] = labs[1:-1, 1:-1, 1:-1]
        if world > 1:
            import torch.distributed as dist
            dist.all_gather(list(gathered_views), gathered[rank].clone())

    def step():
        if args.mode == "strong":
            return strong_step(vol, fm, labels)
        if args.mode == "stack":
            return stack_step(stack_frames, stack_fm, stack_labels, stack_gathered)
        labels.zero_()
        try:
            segmentation.segment_volume_device(u, vol, chunk, margin, feature_map=fm, labels=labels)
        except NoSeedsError:
            pass            # zero surviving seeds on this frame: labels stay 0 (the reference raises); anything else is a bug

    def make_stack():
        mine = isx_dist.frames_for_rank(stack_T, rank, world)
        frames = [torch.from_numpy(positive_volume(stack_shape, seed=t)).to(dev) for t in mine]
        per = (stack_T + world - 1) // world
        gathered = torch.zeros((world, per) + stack_shape, dtype=torch.int32, device=dev)
        return (frames, torch.empty((5,) + stack_shape, dtype=torch.float32, device=dev),
                torch.zeros(tuple(s + 2 for s in stack_shape), dtype=torch.int32, device=dev), gathered)

    stack_frames = stack_fm = stack_labels = stack_gathered = gathered_views = None
    if args.mode == "stack":
        stack_frames, stack_fm, stack_labels, stack_gathered = make_stack()
        gathered_views = stack_gathered.unbind(0)
    if args.mode == "strong" and rank != 0:
        vol.copy_(torch.from_numpy(positive_volume(shape, seed=0)).to(dev))       # the SAME volume everywhere

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
    if args.mode == "stack":
        n_labelled = int((stack_gathered > 0).sum().item())
        n_inst = int(stack_gathered.max().item())
    else:
        n_labelled = int((labels > 0).sum().item())
        n_inst = int(labels.max().item())
    assert n_labelled > 0 and n_inst > 0, "the timed steps produced no labels: not a valid measurement"

    if rank == 0:
        nvox = float(np.prod(shape))
        jobs = world                                              # weak: one frame per rank
        scaling = "weak"
        work = (f"{shape_name} float32 volume per GPU (one frame per rank), chunk (10,256,256), precision {precision}, "
                f"margin (1,64,64) = {n_chunks} chunks")
        if args.mode == "strong":
            jobs, scaling = 1, "strong"
            work = (f"ONE {shape_name} float32 volume on {world} rank(s): chunk table cut by z-rows, feature-map slab all-gather, "
                    f"W0-W3 replicated, component-sharded flood, label max-reduce; chunk (10,256,256), margin (1,64,64) = "
                    f"{n_chunks} chunks, precision {precision}")
        elif args.mode == "stack":
            nvox, jobs, scaling = float(np.prod(stack_shape)), stack_T, "strong"
            n_chunks = len(__import__("iterseg_amd.chunking", fromlist=["x"]).chunk_table(stack_shape, chunk, margin))
            work = (f"BASELINE configs[2]: t = {stack_T} frames of 32x512x512, frame t on rank t mod {world}, "
                    f"{n_chunks} chunks per frame, label all-gather (int32 tiles) inside the timed region, precision {precision}")
            n_chunks *= len(isx_dist.frames_for_rank(stack_T, 0, world))
        mfma_flops, direct_flops = conv_flops_per_chunk(chunk)
        ms_mfma, n_mfma = prof["conv_mfma"]
        t_unet = sum(prof[k][0] for k in ("conv_mfma", "conv_direct", "elementwise")) / args.steps
        t_ws = sum(prof[k][0] for k in ("ws_filters", "ws_ccl", "ws_flood")) / args.steps
        out = {
            "metric": "voxels/sec end-to-end (U-Net + watershed)",
            "value": jobs * nvox * args.steps / elapsed,
            "unit": "voxels/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": DTYPE[precision],
            "data": "synthetic",
            "config": {"workload": work + ", seeded random U-Net weights (seed 0), train-mode BatchNorm per chunk, "
                                          "watershed min_area 10 / max_area 1e7", "mode": args.mode,
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
        extra = world == 1 and args.mode == "weak" and not args.no_extra_legs
        if extra and not args.no_other:
            other = "fp32" if precision != "fp32" else "split-f16"
            out["other_precision"] = {other: other_precision_leg(u, other, vol, fm, labels, chunk, margin, args, n_chunks,
                                                                 mfma_flops)}
        if extra:
            del vol, fm, labels
            out["plugin_path"] = plugin_path_leg(u, chunk, margin, 1e3 * elapsed / args.steps)
            out["config2"] = config2_leg(u, 0, chunk, margin, args.batch)
            out["watershed_structured"] = structured_watershed_leg(u, t_unet_ms=t_unet if shape == (512, 512, 512) else None)
            out["train_step"] = train_step_leg(sd)
            out["train_step_bf16"] = train_step_leg(sd, precision="bf16")
        if extra and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(sd, chunk, margin)
    # N > 1, default mode: the two jobs with a collective in the timed region ride along (one step each, all ranks)
    if world > 1 and args.mode == "weak" and not args.no_extra_legs:
        import torch.distributed as dist
        legs = {}
        v0 = torch.from_numpy(positive_volume((512, 512, 512), seed=0)).to(dev)
        f0 = torch.empty((5, 512, 512, 512), dtype=torch.float32, device=dev)
        l0 = torch.zeros((514, 514, 514), dtype=torch.int32, device=dev)
        for name in ("strong_scaling", "config3_stack"):
            if name == "config3_stack":
                del v0, f0, l0
                stack_frames, stack_fm, stack_labels, stack_gathered = make_stack()
                gathered_views = stack_gathered.unbind(0)
                run = lambda: stack_step(stack_frames, stack_fm, stack_labels, stack_gathered)
                total_vox = float(np.prod(stack_shape)) * stack_T
            else:
                run = lambda: strong_step(v0, f0, l0)
                total_vox = 512.0 ** 3
            run()                                                  # warm-up (workspace shapes, RCCL channels)
            sync_all()
            t0 = time.perf_counter()
            run()
            sync_all()
            tt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=red_dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            legs[name] = {"value": total_vox / float(tt.item()), "unit": "voxels/s", "ms": 1e3 * float(tt.item()), "n_gpus": world,
                          "scaling": "strong"}
        if rank == 0:
            legs["strong_scaling"]["workload"] = "ONE 512^3 volume on all ranks (segment_volume_sharded): slab all-gather of the feature map, component-sharded flood, int32 label max-reduce"
            legs["config3_stack"]["workload"] = "BASELINE configs[2]: 16 frames of 32x512x512, frame t on rank t mod N, int32 label all-gather in the timed region"
            out.update(legs)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
