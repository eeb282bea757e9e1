"""Development aid (GPU box): a scene rendered with and without speculative glossy fans (option speculate_fans): identical pictures? kernel times."""
import sys, time, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import fray_amd
def run(path, W, H, spec, reps=1):
    s = fray_amd.Scene.parseScene(path)
    s.settings.frameWidth, s.settings.frameHeight = W, H
    s.beginRender(0)
    s.set_option("speculate_fans", spec)
    img, st = s.render()
    ms = []
    for _ in range(reps):
        img, st = s.render(); ms.append(st["ms_kernels"])
    return img, min(ms)
for path, W, H in (("tests/scenes/glossy_fans/scene.fray", 160, 120), ("tests/scenes/glossy_fans/scene.fray", 640, 480), ("scenes/hw9/dragon.fray", 1920, 1080)):
    a, ta = run(path, W, H, 0, 3)
    b, tb = run(path, W, H, 1, 3)
    print(path, W, H, "plain %.3f ms, speculative %.3f ms, identical pixels %.6f, max abs diff %g, finite %s, mean %.4f" % (ta, tb, (a == b).all(axis=2).mean(), np.abs(a - b).max(), np.isfinite(b).all(), b.mean()))
