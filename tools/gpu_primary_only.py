"""Bring-up helper: primary-ray pass of every config scene at full size (for rocprofv3 runs)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fray_amd
fray_amd.lib.frayhip_init(0)
for name, W, H in [('cornell_box.fray', 1920, 1080), ('boxed.fray', 1920, 1080), ('hw9/dragon.fray', 1920, 1080), ('smallpt.fray', 1920, 1080)]:
    s = fray_amd.Scene.parseScene(os.path.join(ROOT, 'scenes', name))
    s.settings.frameWidth, s.settings.frameHeight, s.settings.wantAA = W, H, 0
    s.beginRender()
    for i in range(2):
        ids, dist, st = s.primary_hits()
    print(name, 'k_primary %.3f ms  %.1f Mrays/s' % (st['ms_trace'], W * H / st['ms_trace'] / 1e3), flush=True)
    s.close()
