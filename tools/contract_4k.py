"""Development aid (GPU box): BASELINE configs[4] (smallpt 4096 x 4096 x 1024 spp) in both arithmetics: how many pixels differ, by how much."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fray_amd
fray_amd.lib.frayhip_init(0)
s = fray_amd.Scene.parseScene(os.path.join(ROOT, "scenes", "smallpt.fray"))
s.settings.frameWidth, s.settings.frameHeight, s.settings.gi, s.settings.numPaths = 4096, 4096, 1, int(sys.argv[1]) if len(sys.argv) > 1 else 1024
s.beginRender()
t = time.time(); a, _ = s.render(seed=42); ta = time.time() - t
s.set_option("fp_contract", 1)
t = time.time(); b, _ = s.render(seed=42); tb = time.time() - t
d = a.astype(np.float64) - b
px = int((a != b).any(axis=2).sum())
print("smallpt 4096 x 4096 x %d spp: exact %.2f s, fp_contract %.2f s (host buffers); pixels that differ %d of %d; rms per channel %s; largest difference %.3g (relative to the pixel: %.3g)"
      % (s.settings.numPaths, ta, tb, px, a.shape[0] * a.shape[1], np.sqrt((d ** 2).mean(axis=(0, 1))), np.abs(d).max(), (np.abs(d) / np.maximum(np.abs(a), 1e-6)).max()))
s.close()
