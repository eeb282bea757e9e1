"""Prints the work counters of one headline frame (cornell_box 1080p x 64 spp)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fray_amd
fray_amd.lib.frayhip_init(0)
s = fray_amd.Scene.parseScene(os.path.join(ROOT, 'scenes', 'cornell_box.fray'))
s.settings.frameWidth, s.settings.frameHeight, s.settings.numPaths = 1920, 1080, 64
s.beginRender()
img, st = s.render(stats=True)
for k, v in st.items():
    print('%-18s %s' % (k, v))
r = st['closest_rays'] + st['shadow_rays']
print('per ray: node %.2f tri %.2f prim %.2f' % (st['node_tests'] / r, st['tri_tests'] / r, st['prim_tests'] / r))
print('alg bytes per closest ray %.0f, per shadow ray %.0f' % (st['alg_bytes_trace'] / st['closest_rays'], st['alg_bytes_shadow'] / st['shadow_rays']))
