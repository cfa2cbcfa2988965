"""Worker of tests/test_gpu_unet.py::test_results_do_not_depend_on_gpu_time_slicing: the chunked predict and the one-frame
host path on toy frames, repeated while a second copy of this script shares the GPU; every repetition must reproduce the
first one bit for bit (feature maps and labels).  Exit code 0 = no mismatch."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from iterseg_amd import segmentation as seg, weights
    from iterseg_amd.unet import UNet
    from tests.synth import positive_volume
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    chunk, margin = (4, 32, 32), (1, 8, 8)
    u = UNet(1, 5, device=0, batch=4)
    u.load_state_dict(weights.generate_state_dict(0))
    vols = []
    for t in range(4):
        v = positive_volume((8, 40, 48), 30 + t)
        vols.append((v / v.max()).astype(np.float32))

    def frame(v):
        cur = np.zeros(tuple(s + 2 for s in v.shape), np.uint32)
        seg.affinity_watershed_for_chunks(v.copy(), cur, chunk, margin, unet=u, output_volume=np.zeros((5,) + v.shape, np.float32))
        return cur

    dev = [torch.from_numpy(v).cuda() for v in vols]
    ref_fm = [u.predict_volume(d, chunk, margin).clone() for d in dev]
    ref_lab = [frame(v) for v in vols]
    bad = 0
    for it in range(reps):
        for t in range(len(vols)):
            if not torch.equal(u.predict_volume(dev[t], chunk, margin), ref_fm[t]):
                bad += 1
                print(f"repetition {it}, frame {t}: feature map differs from the first run", flush=True)
            if not np.array_equal(frame(vols[t]), ref_lab[t]):
                bad += 1
                print(f"repetition {it}, frame {t}: labels differ from the first run", flush=True)
    sys.exit(0 if bad == 0 else 1)


if __name__ == "__main__":
    main()
