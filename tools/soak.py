"""Repeats small frames of every kind and watches device memory: the workspace, event pools and
lane streams must stop growing after the first frames."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import fray_amd

fray_amd.lib.frayhip_init(0)
free0 = torch.cuda.mem_get_info()[0]
cases = [("cornell_box.fray", dict(gi=1, numPaths=9)), ("boxed.fray", dict(wantAA=0)), ("hw10/bokeh.fray", dict(wantAA=0, numDOFSamples=3)),
         ("smallpt.fray", dict(gi=0, wantAA=0)), ("forest.fray", dict(wantAA=0, interactive=0, stereoSeparation=0.2))]
first = {}
for rep in range(40):
    for name, over in cases:
        s = fray_amd.Scene.parseScene(os.path.join(ROOT, "scenes", name))
        s.settings.frameWidth, s.settings.frameHeight = 96, 64
        for k, v in over.items():
            setattr(s.settings if hasattr(s.settings, k) else s.camera, k, v)
        s.beginRender()
        for _ in range(3):
            img, _ = s.render(seed=42)
        key = name
        if key in first:
            assert np.array_equal(first[key], img), name
        else:
            first[key] = img
        s.close()
    if rep in (1, 39):
        torch.cuda.synchronize()
        print("rep %d: device memory in use by this process' allocations: %.1f MB" % (rep, (free0 - torch.cuda.mem_get_info()[0]) / 1e6), flush=True)
print("soak ok: 40 x 5 scenes x 3 frames, every frame identical to its first rendering")
