"""Development aid (GPU box): the path tracer's exact arithmetic against option "fp_contract" = 1 on several scenes -- frame time of each,
how many pixels differ, per-channel RMS of the difference, largest difference.
usage: python tools/contract_check.py [steps]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import fray_amd  # noqa: E402

CASES = [("cornell_box.fray", 1920, 1080, dict(gi=1, numPaths=64)),
         ("smallpt.fray", 1920, 1080, dict(gi=1, numPaths=64)),
         ("boxed.fray", 960, 540, dict(gi=1, numPaths=16)),
         ("../tests/scenes/csg_nested.fray", 960, 720, dict(gi=1, numPaths=16)),
         ("hw12/sphtri.fray", 960, 540, dict(gi=1, numPaths=32))]


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    fray_amd.lib.frayhip_init(0)
    for name, W, H, over in CASES:
        s = fray_amd.Scene.parseScene(os.path.join(ROOT, "scenes", name))
        s.settings.frameWidth, s.settings.frameHeight = W, H
        for k, v in over.items():
            setattr(s.settings, k, v)
        s.beginRender()
        frames, ms = [], []
        for mode in (0, 1):
            s.set_option("fp_contract", mode)
            f = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
            s.render_device(f.data_ptr(), seed=42)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                s.render_device(f.data_ptr(), seed=42)
            torch.cuda.synchronize()
            ms.append((time.perf_counter() - t0) * 1e3 / steps)
            frames.append(f.cpu().numpy())
        d = frames[1].astype(np.float64) - frames[0]
        differ = int((frames[0] != frames[1]).any(axis=2).sum())
        print("%-34s %4dx%-4d exact %8.2f ms  contracted %8.2f ms (%.3fx)  pixels differing %d of %d  rms %s  max %.3g" % (
            name, W, H, ms[0], ms[1], ms[0] / ms[1], differ, W * H, np.sqrt((d ** 2).mean(axis=(0, 1))), np.abs(d).max()), flush=True)
        s.close()


if __name__ == "__main__":
    main()
