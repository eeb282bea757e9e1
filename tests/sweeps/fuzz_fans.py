"""One-off sweep (run on the GPU box): tests/test_fuzz_parity.py's generator with fans=True (point lights, glossy fans of 8-13 samples) over seeds LO..HI, all three
kernel families: the speculative picture against the picture without speculation, the counting variant's and the CPU checker's.
usage: python tests/sweeps/fuzz_fans.py LO HI"""
import sys, os, numpy as np, pathlib, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import fray_amd
from fray_amd import abi
import test_fuzz_parity as T
from oracle.oracle import Oracle
orc = Oracle(abi)
fray_amd.lib.frayhip_init(0)
lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad = 0
tot = np.zeros(4, np.int64)
for seed in range(lo, hi):
    rng = np.random.default_rng(1000 + seed)
    tmp = pathlib.Path(tempfile.mkdtemp())
    s = fray_amd.Scene.parseScene(T.random_scene(rng, tmp, 0, flavour=seed % 3, fans=True))
    s.camera.stereoSeparation = 0.0
    s.beginRender()
    img, _ = s.render(seed=seed)
    fig = [s.get_option(k) for k in ("fans_filed", "fan_children", "fan_children_looked_up", "fans_given_up")]
    tot += fig
    img2, _ = s.render(seed=seed, stats=True)
    s.set_option("speculate_fans", 0)
    img3, _ = s.render(seed=seed)
    ref, _ = orc.render(s.desc, abi.MODE_RENDER, seed=seed)
    ok = np.array_equal(img, img2) and np.array_equal(img, img3) and np.all(np.abs(img.astype(np.float64) - ref) <= 1e-5 * np.maximum(1.0, np.abs(ref)))
    if not ok or fig[0] == 0:
        bad += 1
        print('seed', seed, 'flavour', seed % 3, 'figures', fig, 'vs counting', int((img != img2).any(axis=2).sum()), 'vs no speculation', int((img != img3).any(axis=2).sum()),
              'vs oracle beyond last places', int((np.abs(img.astype(np.float64) - ref) > 1e-5 * np.maximum(1.0, np.abs(ref))).any(axis=2).sum()), flush=True)
    s.close()
    if seed % 20 == 19:
        print('... seed', seed, 'done, bad so far', bad, 'totals filed / children / looked up / given up', tot.tolist(), flush=True)
print('seeds', lo, hi, 'bad', bad, 'totals', tot.tolist())
