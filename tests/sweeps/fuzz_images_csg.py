"""One-off sweep (run on the GPU box): tests/test_fuzz_parity.py's scene generator over seeds LO..HI (flavour 0: Cube / CSG + KD meshes), hit records,
both integrators, timed and instrumented kernel variants against the CPU checker; prints the seeds whose pictures differ by more than last places.
usage: python tests/sweeps/fuzz_images_csg.py LO HI"""
import sys, os, numpy as np, pathlib, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # tests/sweeps/ -> the repo
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import fray_amd
from fray_amd import abi
import test_fuzz_parity as T
from oracle.oracle import Oracle
orc = Oracle(abi)
fray_amd.lib.frayhip_init(0)
lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for seed in range(lo, hi):
    rng = np.random.default_rng(1000 + seed)
    tmp = pathlib.Path(tempfile.mkdtemp())
    path = T.random_scene(rng, tmp, seed % 2)
    s = fray_amd.Scene.parseScene(path)
    s.beginRender()
    ids, dist, _ = s.primary_hits(stats=False)
    oi, od, _ = orc.render(s.desc, abi.MODE_PRIMARY_ID)
    img, _ = s.render(seed=seed)
    img2, _ = s.render(seed=seed, stats=True)
    ref, _ = orc.render(s.desc, abi.MODE_RENDER, seed=seed)
    ok = np.array_equal(ids, oi) and np.array_equal(dist, od) and np.array_equal(img, ref) and np.array_equal(img2, ref)
    # the other integrator on the same scene
    s.settings.gi = 1 - (seed % 2)
    s.beginRender()
    img3, _ = s.render(seed=seed)
    ref3, _ = orc.render(s.desc, abi.MODE_RENDER, seed=seed)
    d3 = int((np.abs(img3 - ref3).max(axis=2) > 0).sum())
    big = np.sqrt(((img - ref) ** 2).mean()) > 1e-6 or np.sqrt(((img2 - ref) ** 2).mean()) > 1e-6 or np.sqrt(((img3 - ref3) ** 2).mean()) > 1e-6 or not np.array_equal(ids, oi) or not np.array_equal(dist, od)
    if big:
        bad += 1
        print('seed', seed, 'gi', seed % 2, 'ids', np.array_equal(ids, oi), 'dist', np.array_equal(dist, od), 'timed differing', int((np.abs(img - ref).max(axis=2) > 0).sum()),
              'instrumented differing', int((np.abs(img2 - ref).max(axis=2) > 0).sum()), 'other integrator differing', d3, flush=True)
    s.close()
    if seed % 10 == 9:
        print('... seed', seed, 'done, bad so far', bad, flush=True)
print('seeds', lo, hi, 'bad', bad)
