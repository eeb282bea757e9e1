import os, sys, time
sys.path.insert(0, '/root/repo' if os.path.isdir('/root/repo') else '.')
import torch, fray_amd
fray_amd.lib.frayhip_init(0)
s = fray_amd.Scene.parseScene(os.path.join(os.path.dirname(fray_amd.__file__), '..', 'scenes', 'hw10', 'bokeh.fray'))
s.settings.frameWidth, s.settings.frameHeight = 1920, 1080
s.beginRender()
for i in range(3):
    t = time.time(); img, st = s.render(seed=42); print('bokeh 1080p as shipped: %.1f ms (kernels %.1f)' % ((time.time() - t) * 1e3, st['ms_kernels']))
