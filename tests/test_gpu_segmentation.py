"""GPU tests of the operator / plugin boundary (iterseg_amd/segmentation.py, mirror of the reference's
src/iterseg/segmentation.py:24-195,700-916): same call sequence and error behaviour as the reference, labels written
in place, and staged parity -- the labels equal the CPU oracle's watershed of the feature map the HIP U-Net produced
(bit-exact), the feature map itself being covered by test_gpu_unet.py."""
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CHUNK, MARGIN = (4, 32, 32), (1, 8, 8)


@pytest.fixture(scope="module")
def unet():
    from iterseg_amd import weights
    from iterseg_amd.unet import UNet
    u = UNet(1, 5, batch=8)
    u.load_state_dict(weights.generate_state_dict(0))
    return u


def _oracle_labels(u, vol):
    """oracle watershed of the feature map the HIP U-Net gives for the (already normalised) volume"""
    from oracle import watershed as ows
    fm = u.predict_volume(torch.from_numpy(vol).cuda(), CHUNK, MARGIN).cpu().numpy()
    lab = np.zeros(tuple(s + 2 for s in vol.shape), np.uint32)
    try:
        ows.segment_output_image(fm, out=lab.ravel())
    except ValueError:
        return None
    return lab


def test_processing_function_writes_labels_in_place(unet):
    from iterseg_amd import segmentation as seg
    from tests.synth import positive_volume
    vol = positive_volume((10, 48, 40), 3)
    cur = np.zeros(tuple(s + 2 for s in vol.shape), np.uint32)
    fm = np.ones((5,) + vol.shape, np.float32)
    ret = seg.affinity_watershed_for_chunks(vol, cur, CHUNK, MARGIN, unet=unet, output_volume=fm)
    assert ret is None and not fm.any()                         # labels by mutation; output_volume zeroed (:195)
    ref = _oracle_labels(unet, vol)
    assert ref is not None and np.array_equal(cur, ref) and cur.max() > 0
    assert not cur[0].any() and not cur[:, 0].any() and not cur[:, :, -1].any()      # 1-voxel border stays 0


def test_processing_function_errors_like_reference(unet):
    from iterseg_amd import segmentation as seg
    vol = np.ones((8, 32, 32), np.float32)
    cur = np.zeros((10, 34, 34), np.uint32)
    with pytest.raises(ValueError, match="output_volume must not be None"):
        seg.affinity_watershed_for_chunks(vol, cur, CHUNK, MARGIN, unet=unet, output_volume=None)
    with pytest.raises(ValueError, match="unet must not be none"):
        seg.affinity_watershed_for_chunks(vol, cur, CHUNK, MARGIN, unet=None, output_volume=np.zeros((5, 8, 32, 32), np.float32))
    with pytest.raises(ValueError):                             # a volume smaller than the chunk: make_chunks breaks upstream too
        seg.affinity_watershed_for_chunks(vol[:2], np.zeros((4, 34, 34), np.uint32), CHUNK, MARGIN, unet=unet,
                                          output_volume=np.zeros((5, 2, 32, 32), np.float32))


def test_segment_single_volume_normalises_and_drops_empty_slices(unet):
    from iterseg_amd import segmentation as seg
    from tests.synth import positive_volume
    vol = positive_volume((9, 40, 40), 5) * 7.0
    vol[4] = 0.0                                                # an all-zero plane: min() == 0 -> remove_sum_zero_slices
    work = vol.copy()
    config = {"unet": unet, "output_volume": np.zeros((5, 8, 40, 40), np.float32)}
    out = seg.segment_single_volume(work, CHUNK, config, MARGIN, seg.affinity_watershed_for_chunks)
    assert out.shape == (8, 40, 40) and out.dtype == np.uint32  # the empty plane is gone (shape change, as upstream)
    kept = np.delete(vol, 4, axis=0)
    ref = _oracle_labels(unet, (kept / kept.max()).astype(np.float32))
    assert ref is not None and np.array_equal(out, ref[1:-1, 1:-1, 1:-1])


def test_operator_entry_on_a_stack(unet, tmp_path):
    """affinity_unet_watershed(viewer, layer, ...) with a stand-in layer object and a .pt file: per-frame labels equal
    the per-frame calls; frames are independent; a second call on a partly filled store skips finished frames."""
    from iterseg_amd import segmentation as seg, weights
    from tests.synth import positive_volume
    path = str(tmp_path / "unet.pt")
    torch.save(weights.to_torch_state_dict(weights.generate_state_dict(0)), path)
    data = np.stack([positive_volume((8, 40, 48), s) for s in (1, 2, 3)])
    layer = types.SimpleNamespace(data=data, scale=(4, 1, 1), translate=(0, 0, 0), metadata={})
    labels = seg.segmenters['affinity-unet-watershed'](None, layer, save_dir=None, name='t', unet_or_config_file=path,
                                                       layer_reference=None, chunk_size=CHUNK, margin=MARGIN, debug=True)
    labels = np.asarray(labels)
    assert labels.shape == data.shape and labels.dtype == np.int32
    for t in range(3):
        ref = _oracle_labels(unet, (data[t] / data[t].max()).astype(np.float32))
        assert ref is not None and np.array_equal(labels[t].astype(np.uint32), ref[1:-1, 1:-1, 1:-1]), t
    with pytest.raises(AssertionError):
        seg.affinity_watershed_prep_config(layer, str(tmp_path / "missing.pt"), None)


def test_json_config_and_single_process_sharded_path(unet, tmp_path):
    """The .json form of the network argument (segmentation.py:98-107, read as documented) and the world-size-1 path of
    segment_volume_sharded (== segment_volume_device)."""
    import json
    from iterseg_amd import segmentation as seg, weights, predict
    from iterseg_amd.distributed import segment_volume_sharded
    from tests.synth import positive_volume
    pt = str(tmp_path / "u.pt")
    torch.save(weights.to_torch_state_dict(weights.generate_state_dict(0)), pt)
    cfg = str(tmp_path / "cfg.json")
    json.dump({"unet": pt}, open(cfg, "w"))
    assert seg.read_config_json(cfg) == {"unet": pt}
    layer = types.SimpleNamespace(data=np.zeros((8, 32, 32), np.float32))
    config = seg.affinity_watershed_prep_config(layer, cfg, None)
    assert config["output_volume"].shape == (5, 8, 32, 32) and config["unet"].out_channels == 5
    assert predict.get_device() in ("cuda", "cuda:0") or "cuda" in str(predict.get_device())
    vol = torch.from_numpy(positive_volume((8, 48, 48), 4)).cuda()
    a = segment_volume_sharded(unet, vol, CHUNK, MARGIN)
    b = seg.segment_volume_device(unet, vol, CHUNK, MARGIN)
    assert torch.equal(a, b) and int(a.max()) > 0


def test_two_rank_sharded_volume():
    """One volume on two ranks (gloo, both on cuda:0): chunk-sharded U-Net + summed feature map, replicated W0-W3,
    component-sharded flood + max of the labels == the single-GPU labels (tests/_sharded_worker.py)."""
    import os, socket, subprocess, sys
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "tests", "_sharded_worker.py")]
    r = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]


def test_plugin_call_from_worker_threads_two_contexts():
    """SURVEY.md 8(b) Threading: the reference drives the processing function from a napari thread_worker QThread
    (segmentation.py:808-825).  Two contexts (two UNet objects) are driven concurrently from two non-main threads, one
    frame sequence each; every result equals the same call made alone on the main thread."""
    import threading
    from iterseg_amd import segmentation as seg, weights
    from iterseg_amd.unet import UNet
    from tests.synth import positive_volume
    sd = weights.generate_state_dict(0)
    nets = [UNet(1, 5, batch=4) for _ in range(2)]
    for u in nets:
        u.load_state_dict(sd)
    vols = [[positive_volume((8, 40, 48), 10 * k + t) for t in range(3)] for k in range(2)]

    def run(u, vol):
        cur = np.zeros(tuple(s + 2 for s in vol.shape), np.uint32)
        seg.affinity_watershed_for_chunks(vol.copy(), cur, CHUNK, MARGIN, unet=u,
                                          output_volume=np.zeros((5,) + vol.shape, np.float32))
        return cur

    expected = [[run(nets[k], v) for v in vols[k]] for k in range(2)]
    got = [[None] * 3 for _ in range(2)]
    errors = []

    def worker(k):
        try:
            assert threading.current_thread() is not threading.main_thread()
            for rep in range(2):
                for t, v in enumerate(vols[k]):
                    got[k][t] = run(nets[k], v)
        except BaseException as e:          # surfaced on the main thread like napari's handle_errors
            errors.append(e)

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=300)
    assert not errors, errors
    for k in range(2):
        for t in range(3):
            assert got[k][t] is not None and np.array_equal(got[k][t], expected[k][t]) and expected[k][t].max() > 0


def test_segment_stack_with_the_hip_path(unet):
    """SURVEY.md 8(e), frames of a (t,z,y,x) stack: distributed.segment_stack drives the HIP compute (not the injected
    oracle of the gloo CPU test) -- single process here, two gloo ranks in tests/_stack_worker.py; per-frame labels equal
    the per-frame plugin calls."""
    from iterseg_amd import segmentation as seg
    from iterseg_amd.distributed import segment_stack
    from tests.synth import positive_volume
    data = np.stack([positive_volume((8, 40, 48), s) for s in (21, 22, 23)])

    def frame(vol):
        vol = vol / vol.max()
        cur = np.zeros(tuple(s + 2 for s in vol.shape), np.uint32)
        seg.affinity_watershed_for_chunks(vol.astype(np.float32), cur, CHUNK, MARGIN, unet=unet,
                                          output_volume=np.zeros((5,) + vol.shape, np.float32))
        return cur[1:-1, 1:-1, 1:-1].astype(np.int32)

    out = segment_stack(data, frame)
    assert tuple(out.shape) == data.shape and out.dtype == torch.int32
    for t in range(3):
        assert np.array_equal(out[t].cpu().numpy(), frame(data[t])) and int(out[t].max()) > 0
    # two ranks (gloo, both on cuda:0): frame t on rank t mod 2, all-gather of the label tiles
    import os, socket, subprocess, sys
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "tests", "_stack_worker.py")]
    r = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=300)
    if r.returncode != 0 and os.path.isdir(os.path.join(root, "gpurun_out")):
        open(os.path.join(root, "gpurun_out", "stack_worker_stderr.log"), "w").write(r.stdout + "\n----\n" + r.stderr)
    assert r.returncode == 0, r.stderr[-2000:]


def test_frame_pipeline_equals_frame_by_frame_loop(unet, monkeypatch):
    """segmentation_loop over a host stack: the pipelined route (isx_pipe_*: upload of t+1 / download of t-1 under the
    compute of t, normalisation on the device) gives the labels of the frame-by-frame route bit for bit -- including a
    frame that loses an all-zero plane (min() == 0 -> remove_sum_zero_slices changes its shape upstream; here such a frame
    cannot be stored in a fixed-shape stack, so the loop must fail the same way on both routes), a frame with zeros that
    keeps its shape, a frame scaled by 9 (normalisation), and the warm restart."""
    from iterseg_amd import segmentation as seg
    from tests.synth import positive_volume
    data = np.stack([positive_volume((8, 40, 48), s) for s in (1, 2, 3, 4, 5)])
    data[1] *= 9.0
    data[2][3, 5, 7] = 0.0                                       # min() == 0 but no empty plane: shape kept
    layer = types.SimpleNamespace(data=data)
    config = seg.affinity_watershed_prep_config(layer, unet, None)
    assert not config['output_volume'].flags.writeable           # pristine zeros: never written, never re-zeroed
    outs = {}
    for pipe in ("1", "0"):
        monkeypatch.setenv("ISX_FRAME_PIPE", pipe)
        store = np.zeros(data.shape, np.int32)
        store[3, 0, 0, 0] = 7                                    # warm restart: frame 3 counts as done
        order = list(seg.segmentation_loop(None, data, CHUNK, MARGIN, store, seg.affinity_watershed_for_chunks, config))
        assert order == [0, 1, 2, 4]
        outs[pipe] = store
    assert np.array_equal(outs["1"], outs["0"]) and outs["1"][0].max() > 0
    assert not outs["1"][3].ravel()[1:].any()
    assert not config['output_volume'].any()
    for t in (1, 2):
        ref = _oracle_labels(unet, (data[t] / data[t].max()).astype(np.float32))
        assert np.array_equal(outs["1"][t].astype(np.uint32), ref[1:-1, 1:-1, 1:-1]), t


def test_frame_pipe_direct_use_and_errors(unet):
    """isx_pipe_*: more submits than `depth` are refused, wait without a frame is refused, padded and cropped outputs
    agree with the one-frame host entry, a non-zero current_output is kept by isx_segment_volume_host."""
    from iterseg_amd import segmentation as seg
    from tests.synth import positive_volume
    vols = [positive_volume((8, 40, 48), s) for s in (11, 12, 13)]
    pipe = seg.FramePipe(unet, vols[0].shape, CHUNK, MARGIN, depth=2)
    with pytest.raises(ValueError):
        pipe.wait()
    pipe.submit(vols[0])
    pipe.submit(vols[1])
    with pytest.raises(ValueError):
        pipe.submit(vols[2])
    a = pipe.wait()
    pipe.submit(vols[2])
    b, c = pipe.wait(), pipe.wait()
    pipe.close()
    for v, got in zip(vols, (a, b, c)):
        cur = np.zeros(tuple(s + 2 for s in v.shape), np.uint32)
        seg.affinity_watershed_for_chunks(v, cur, CHUNK, MARGIN, unet=unet, output_volume=np.zeros((5,) + v.shape, np.float32))
        assert np.array_equal(got, cur[1:-1, 1:-1, 1:-1]) and got.max() > 0
    # labels a caller left in current_output survive (the flood keeps pre-labelled voxels: isx_affinity_watershed)
    cur2 = np.zeros_like(cur)
    cur2[0, 0, 0] = 12345                                        # on the border: outside the mask, nothing floods it
    seg.affinity_watershed_for_chunks(vols[2], cur2, CHUNK, MARGIN, unet=unet, output_volume=np.zeros((5,) + v.shape, np.float32))
    assert cur2[0, 0, 0] == 12345
    cur2[0, 0, 0] = 0
    assert np.array_equal(cur2, cur)


def test_unet_refuses_direct_calls_while_its_frame_pipe_is_busy(unet):
    """The context is the pipe's while frames are in flight (segmentation_loop yields between frames): a forward from the
    consumer's side raises instead of racing; ``discard`` finishes a frame whatever its status; the pipe holds the U-Net
    weakly (no cycle keeping 9 GB of slots alive)."""
    import gc
    from iterseg_amd import segmentation as seg
    from tests.synth import positive_volume
    v = positive_volume((8, 40, 48), 21)
    pipe = seg._pipe_for(unet, v.shape, CHUNK, MARGIN)
    assert seg._pipe_for(unet, v.shape, CHUNK, MARGIN) is pipe and pipe.unet is unet
    pipe.submit(v)
    x = v[None, None, :, :32, :32]
    with pytest.raises(RuntimeError, match="frames in flight"):
        unet(x)
    pipe.discard()
    assert pipe.in_flight == 0
    assert unet(x).shape == (1, 5, 8, 32, 32)
    assert not any(r is pipe for r in gc.get_referrers(unet))
    unet._drop_pipe()
