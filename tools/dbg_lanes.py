"""Development aid (GPU box): the headline frame by the timed and by the counting kernels against the oracle's buckets (every 40-th bucket from 17)."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import fray_amd
from fray_amd import abi
from conftest import bucket_xy
from oracle.oracle import Oracle
orc = Oracle(abi)
fray_amd.lib.frayhip_init(0)
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
W, H = 1920, 1080
s = fray_amd.Scene.parseScene(os.path.join(ROOT, "scenes", "cornell_box.fray"))
s.settings.frameWidth, s.settings.frameHeight, s.settings.gi, s.settings.numPaths = W, H, 1, spp
s.beginRender()
ref, _ = orc.render(s.desc, abi.MODE_RENDER, seed=42, bucket_first=17, bucket_stride=40, threads=16)
mask = np.zeros((H, W), bool)
for b in range(17, 40 * 23, 40):
    bx, by = bucket_xy(W, b)
    mask[by * 48:by * 48 + 48, bx * 48:bx * 48 + 48] = True
for lanes, stats in ((4, False), (4, True), (1, False), (1, True), (4, False)):
    s.set_option("pt_lanes", lanes)
    img, st = s.render(seed=42, stats=stats)
    d = (img[mask] != ref[mask]).any(axis=1)
    print("lanes", lanes, "stats", stats, "pixels differing from the oracle:", int(d.sum()), "of", int(mask.sum()), " rays", st["closest_rays"] + st["shadow_rays"], flush=True)
