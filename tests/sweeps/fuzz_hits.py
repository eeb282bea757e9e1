"""One-off sweep (run on the GPU box): primary hit records of random scenes against the CPU checker, bit for bit (this is what found the sincos() camera difference).
usage: python tests/sweeps/fuzz_hits.py LO HI [FLAVOUR]"""
import sys, os, numpy as np, pathlib, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # tests/sweeps/ -> the repo
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import fray_amd
from fray_amd import abi
import test_fuzz_parity as T
from oracle.oracle import Oracle
orc = Oracle(abi)
fray_amd.lib.frayhip_init(0)
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng = np.random.default_rng(1000 + seed)
    tmp = pathlib.Path(tempfile.mkdtemp())
    s = fray_amd.Scene.parseScene(T.random_scene(rng, tmp, seed % 2, flavour=int(sys.argv[3]) if len(sys.argv) > 3 else 2))
    s.beginRender()
    ids, dist, st = s.primary_hits(stats=False)
    oi, od, ost = orc.render(s.desc, abi.MODE_PRIMARY_ID)
    bad = (ids != oi) | (dist != od)
    if bad.any():
        c = s.camera
        print('seed', seed, 'W H', s.settings.frameWidth, s.settings.frameHeight, 'bad', int(bad.sum()), 'yaw pitch roll fov aspect', c.yaw, c.pitch, c.roll, c.fov, c.aspectRatio, 'pos', list(c.pos), flush=True)
    s.close()
print('done')
