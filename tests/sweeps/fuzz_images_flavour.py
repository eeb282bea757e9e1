"""One-off sweep (run on the GPU box): the same for the scene flavours without Cube / CSG geometry (1: KD meshes, 2: only small meshes).
usage: python tests/sweeps/fuzz_images_flavour.py LO HI FLAVOUR"""
import sys, os, numpy as np, pathlib, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # tests/sweeps/ -> the repo
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import fray_amd
from fray_amd import abi
import test_fuzz_parity as T
from oracle.oracle import Oracle
orc = Oracle(abi)
fray_amd.lib.frayhip_init(0)
lo, hi, fl = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
bad = 0
for seed in range(lo, hi):
    rng = np.random.default_rng(1000 + seed)
    tmp = pathlib.Path(tempfile.mkdtemp())
    gi = (seed // 2) % 2
    s = fray_amd.Scene.parseScene(T.random_scene(rng, tmp, gi, flavour=fl))
    s.beginRender()
    msg = []
    for stats in (True, False):
        ids, dist, _ = s.primary_hits(stats=stats)
        oi, od, _ = orc.render(s.desc, abi.MODE_PRIMARY_ID)
        if not (np.array_equal(ids, oi) and np.array_equal(dist, od)): msg.append('hits stats=%s' % stats)
    for g in (gi, 1 - gi):
        s.settings.gi = g
        s.beginRender()
        img, _ = s.render(seed=seed)
        img2, _ = s.render(seed=seed, stats=True)
        ref, _ = orc.render(s.desc, abi.MODE_RENDER, seed=seed)
        if not np.array_equal(img, img2): msg.append('gi %d timed != instrumented' % g)
        if not np.all(np.abs(img.astype(np.float64) - ref) <= 1e-5 * np.maximum(1.0, np.abs(ref))): msg.append('gi %d timed vs ref: %d px' % (g, int((np.abs(img - ref).max(axis=2) > 1e-5).sum())))
    if msg:
        bad += 1
        print('seed', seed, msg, flush=True)
    s.close()
print('flavour', fl, 'seeds', lo, hi, 'bad', bad)
