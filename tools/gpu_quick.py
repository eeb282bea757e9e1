"""Ad-hoc GPU bring-up script: HIP path vs oracle on a few scenes."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fray_amd
from fray_amd import abi
from oracle.oracle import Oracle

orc = Oracle(abi)
GOLD = {('boxed.fray', 640, 480): ('93e63cdbf6f43858', 'b0db0f931c62cd8a'),
        ('zaphod.fray', 1920, 1080): ('d1866e2db47c178d', '47540151ef8c0145'),
        ('cornell_box.fray', 1920, 1080): ('fffcebb6df337a39', 'ca8053a3e5180c7d'),
        ('forest.fray', 1920, 1080): ('400f546dba537592', 'f114db2d6acc504d'),
        ('smallpt.fray', 4096, 4096): ('cd109f1726a6d7e5', 'a071814766cfbb6d'),
        ('hw9/dragon.fray', 1920, 1080): ('0baea1c1c3008f48', 'ef5c8b2a088eefee')}

def scene(name, W, H, **over):
    s = fray_amd.Scene.parseScene(os.path.join(ROOT, 'scenes', name))
    s.settings.frameWidth, s.settings.frameHeight = W, H
    for k, v in over.items():
        if hasattr(s.settings, k): setattr(s.settings, k, v)
        else: setattr(s.camera, k, v)
    return s

fray_amd.lib.frayhip_init(0)
ok = True
for (name, W, H), (hid, hdist) in GOLD.items():
    s = scene(name, W, H, wantAA=0)
    s.beginRender()
    t0 = time.time(); ids, dist, st = s.primary_hits(); t1 = time.time()
    a, b = orc.fnv(ids), orc.fnv(dist)
    good = (a, b) == (hid, hdist)
    ok &= good
    print('PRIMARY', name, W, H, 'OK' if good else 'MISMATCH %s %s' % (a, b), 'kernel %.2f ms' % st['ms_trace'], 'call %.0f ms' % ((t1 - t0) * 1e3), flush=True)
    if not good:
        oi, od, _ = orc.render(s.desc, abi.MODE_PRIMARY_ID)
        bad = np.argwhere((oi != ids) | (od != dist))
        print('   differing pixels:', len(bad), bad[:5].tolist())
        for y, x in bad[:5]: print('    ', (x, y), 'gpu', ids[y, x], repr(dist[y, x]), 'oracle', oi[y, x], repr(od[y, x]))
    s.close()

def cmp_render(name, W, H, tag, **over):
    global ok
    s = scene(name, W, H, **over)
    s.beginRender()
    t0 = time.time(); g, st = s.render(stats=False); t1 = time.time()
    o, ost = orc.render(s.desc, abi.MODE_RENDER)
    t2 = time.time()
    rms = np.sqrt(((g.astype(np.float64) - o) ** 2).mean(axis=(0, 1)))
    mx = np.abs(g - o).max()
    good = bool((rms <= 1e-4).all())
    ok &= good
    print(tag, name, W, H, 'spp', s.samples_per_pixel(), 'rms', rms, 'max', mx, 'mean', o.mean(), 'OK' if good else 'FAIL',
          'gpu %.1f ms (kernels %.1f)' % ((t1 - t0) * 1e3, st['ms_kernels']), 'oracle %.1f s' % (t2 - t1), flush=True)
    g2, st2 = s.render(stats=True)
    same = np.array_equal(g, g2)
    keys = ['closest_rays', 'shadow_rays', 'node_tests', 'kd_inner_visits', 'leaf_refs', 'tri_tests', 'prim_tests', 'smooth_hits', 'samples', 'texture_fetches']
    diff = {k: (st2[k], ost[k]) for k in keys if st2[k] != ost[k]}
    print('    stats-run identical image:', same, 'counter mismatches:', diff, flush=True)
    s.close()

cmp_render('zaphod.fray', 320, 180, 'WHITTED', wantAA=0, dof=0)
cmp_render('boxed.fray', 160, 120, 'WHITTED', wantAA=0)
cmp_render('boxed.fray', 96, 72, 'WHITTED-AA', wantAA=1)
cmp_render('forest.fray', 160, 120, 'WHITTED-DOF', wantAA=0, dof=1, numDOFSamples=8, interactive=0)
cmp_render('cornell_box.fray', 96, 96, 'PT', numPaths=16)
cmp_render('smallpt.fray', 96, 72, 'PT', numPaths=16)
cmp_render('hw12/sphtri.fray', 96, 72, 'PT', numPaths=8)
print('ALL OK' if ok else 'SOME FAILED')
