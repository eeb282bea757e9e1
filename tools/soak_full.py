"""Development aid (GPU box): full-size frames rendered again and again on four batch lanes -- every frame must be the first frame, bit for bit (the hazard of
profiles/r05_experiments/README.md H showed only now and then, and only at full frame size)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import fray_amd
fray_amd.lib.frayhip_init(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 25
cases = [("cornell_box.fray", 1920, 1080, dict(gi=1, numPaths=64)), ("smallpt.fray", 1920, 1080, dict(gi=1, numPaths=64)),
         ("boxed.fray", 960, 540, dict(gi=1, numPaths=16)), ("../tests/scenes/csg_nested.fray", 960, 720, dict(gi=1, numPaths=16)),
         ("forest.fray", 1920, 1080, dict(wantAA=0, dof=1, numDOFSamples=16, interactive=0)), ("hw9/dragon.fray", 1920, 1080, dict(wantAA=0))]
bad = 0
for name, W, H, over in cases:
    s = fray_amd.Scene.parseScene(os.path.join(ROOT, "scenes", name))
    s.settings.frameWidth, s.settings.frameHeight = W, H
    for k, v in over.items():
        setattr(s.settings if hasattr(s.settings, k) else s.camera, k, v)
    s.beginRender()
    for arith in (0, 1):
        if arith and not s.settings.gi:
            continue
        s.set_option("fp_contract", arith)
        first, t0, differing = None, time.time(), 0
        for rep in range(N):
            img, _ = s.render(seed=42)
            if first is None:
                first = img
            elif not np.array_equal(first, img):
                differing += 1
        bad += differing
        print("%-34s %dx%d fp_contract=%d: %d frames, %d differ from the first, %.1f ms per frame (host buffers)" % (name, W, H, arith, N, differing, (time.time() - t0) * 1e3 / N), flush=True)
    s.close()
print("soak_full:", "ok" if bad == 0 else "%d frames differed" % bad)
sys.exit(1 if bad else 0)
