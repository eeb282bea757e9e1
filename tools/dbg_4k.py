"""Development aid (GPU box): smallpt.fray at 4096 x 4096 with a few samples per pixel (the batches of BASELINE configs[4], fewer of them), ms per frame."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import fray_amd
fray_amd.lib.frayhip_init(0)
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 32
s = fray_amd.Scene.parseScene(os.path.join(ROOT, "scenes", "smallpt.fray"))
s.settings.frameWidth, s.settings.frameHeight, s.settings.gi, s.settings.numPaths = 4096, 4096, 1, spp
s.beginRender()
f = torch.zeros((4096, 4096, 3), dtype=torch.float32, device="cuda")
s.render_device(f.data_ptr(), seed=42)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(2):
    st = s.render_device(f.data_ptr(), seed=42)
torch.cuda.synchronize()
print("%s: smallpt 4096x4096 x %d spp: %.1f ms per frame, %d bounce launches" % (os.path.basename(ROOT), spp, (time.perf_counter() - t0) * 500, st["trace_launches"]))
