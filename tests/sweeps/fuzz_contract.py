"""One-off sweep (run on the GPU box): option "fp_contract" over generated scenes (tests/test_fuzz_parity.py's generator, path traced, the flavours whose kernel
variants have a contracted copy: 1 = KD meshes, 2 = only small meshes): the contracted frame against the exact frame and against the CPU checker.  Prints every seed
whose contracted frame differs from the exact one at all, and the largest RMS seen (the guarantee is 1e-4 per channel).
usage: python tests/sweeps/fuzz_contract.py LO HI FLAVOUR"""
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
worst, differing, routed = 0.0, 0, 0
for seed in range(lo, hi):
    rng = np.random.default_rng(1000 + seed)
    tmp = pathlib.Path(tempfile.mkdtemp())
    s = fray_amd.Scene.parseScene(T.random_scene(rng, tmp, 1, flavour=fl))
    s.beginRender()
    a, _ = s.render(seed=seed)
    s.set_option("fp_contract", 1)
    b, _ = s.render(seed=seed)
    n = s.get_option("contracted_launches")
    routed += 1 if n > 0 else 0
    ref, _ = orc.render(s.desc, abi.MODE_RENDER, seed=seed)
    rms_ab = float(np.sqrt(((a.astype(np.float64) - b) ** 2).mean(axis=(0, 1)).max()))
    rms_ref = float(np.sqrt(((b.astype(np.float64) - ref) ** 2).mean(axis=(0, 1)).max()))
    px = int((a != b).any(axis=2).sum())
    worst = max(worst, rms_ab, rms_ref)
    if px:
        differing += 1
        print('seed', seed, 'contracted launches', n, 'pixels that differ from the exact frame', px, 'of', a.shape[0] * a.shape[1], 'rms vs exact %.3g vs checker %.3g' % (rms_ab, rms_ref), flush=True)
    s.close()
print('flavour', fl, 'seeds', lo, hi, 'frames routed through contracted kernels', routed, 'frames with any differing pixel', differing, 'largest rms %.3g' % worst, 'PASS' if worst <= 1e-4 else 'ABOVE 1e-4')
