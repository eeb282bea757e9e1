"""Scenes nobody hand-picked, through the reference's object code: tests/test_fuzz_parity.py's random scenes (every geometry kind under random transforms,
KD and tree-less meshes, CsgOps three levels deep, all shaders, bitmap / bump / Fresnel textures, thin lens, stereo) are written to a scratch folder,
rendered and probed by oracle/_ref -- the reference's OWN parser, OBJ / BMP loaders, KD builder, geometry, lights, camera, shaders -- and by the oracle, which
is fed by the PRODUCT's parser and KD builder.  Probe records and pictures must be equal bit for bit, for both integrators on every scene.  (The fixed
fixtures of tests/test_oracle_vs_ref.py are scenes somebody chose; eight of these generated ones are committed as ref_fuzz*.npz for the GPU suite.)
Skipped where oracle/_ref is not built (it needs the reference tree).  FRAY_REF_FUZZ_SEEDS=50 runs the long form."""
import os
import pathlib
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT
from test_fuzz_parity import random_scene

REF_SO = os.path.join(ROOT, "oracle", "_ref", "libfray_ref.so")
N = int(os.environ.get("FRAY_REF_FUZZ_SEEDS", "12"))

pytestmark = pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref is only built where the reference tree is mounted")


def ref_fixture(scene_path, name, W, H, over, outdir):
    env = dict(os.environ, GOLDEN_OUT=str(outdir))
    env.pop("FRAY_REF_FACES_DIR", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "make_golden.py"), name, scene_path, str(W), str(H), over, "3", "120"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    return np.load(os.path.join(outdir, "ref_%s.npz" % name))


def compare(fray, abi, oracle, z, scene_path, gi):
    s = fray.Scene.parseScene(scene_path)
    s.settings.gi = gi
    assert (s.settings.frameWidth, s.settings.frameHeight) == (int(z["W"]), int(z["H"]))
    S, D = z["ray_start"], z["ray_dir"]
    for i in range(len(S)):
        out = np.zeros(9)
        a, b = np.ascontiguousarray(S[i]), np.ascontiguousarray(D[i])
        hid = oracle.lib.fray_oracle_probe(s.desc, a.ctypes.data, b.ctypes.data, out.ctypes.data)
        assert hid == z["hit_id"][i], (i, hid, z["hit_id"][i])
        want = z["hit_rec"][i]
        if hid == -1:
            assert out[0] == want[0] == 1e99
        elif hid <= -2:
            assert np.array_equal(out[:7], want[:7])
        else:
            assert np.array_equal(out, want), (i, out, want)
    img, _ = oracle.render(s.desc, abi.MODE_RENDER, seed=42, threads=4)
    ref = z["image"]
    assert ref.shape == img.shape and np.isfinite(ref).all()
    assert np.array_equal(img, ref), "gi %d: max abs diff %g over %d pixels" % (gi, np.abs(img - ref).max(), int((img != ref).any(axis=2).sum()))
    s.close()


@pytest.mark.parametrize("seed", range(3000, 3000 + N))
def test_generated_scene_through_reference_object_code_and_oracle(fray, abi, oracle, tmp_path, seed):
    rng = np.random.default_rng(seed)
    gi = seed % 2
    scene = random_scene(rng, pathlib.Path(tmp_path), gi, flavour=seed % 3, bump_on=("blob",))      # bump maps on meshes only: elsewhere the reference reads uninitialised memory
    s = fray.Scene.parseScene(scene)
    W, H = s.settings.frameWidth, s.settings.frameHeight
    s.close()
    for g in (gi, 1 - gi):
        z = ref_fixture(scene, "fz%d_%d" % (seed, g), W, H, "gi=%d" % g, tmp_path)
        compare(fray, abi, oracle, z, scene, g)


@pytest.mark.parametrize("seed", (3200, 3201, 3202))
def test_generated_glossy_fan_scene_through_reference_object_code_and_oracle(fray, abi, oracle, tmp_path, seed):
    """The scenes the device's speculative glossy fans apply to (point lights only, fans of 8-13 samples: tests/test_fuzz_parity.py): here the oracle they
    are compared with is itself compared with the reference's object code."""
    rng = np.random.default_rng(seed)
    scene = random_scene(rng, pathlib.Path(tmp_path), 0, flavour=seed % 3, bump_on=("blob",), fans=True)
    s = fray.Scene.parseScene(scene)
    W, H = s.settings.frameWidth, s.settings.frameHeight
    s.close()
    z = ref_fixture(scene, "fans%d" % seed, W, H, "gi=0", tmp_path)
    compare(fray, abi, oracle, z, scene, 0)
