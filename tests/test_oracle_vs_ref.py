"""Pins the oracle's geometry AND colour path to fixtures produced by the partial reference build
(oracle/_ref: the reference's own object code for parser / geometry / KD / lights / camera /
shaders / textures, with the integrator loop restated -- see oracle/ref_glue.cpp).  Same host
toolchain on both sides (glibc libm, libstdc++ <random>), so equality is bit for bit."""
import glob
import os

import numpy as np
import pytest

from conftest import ROOT, open_scene

FIXTURES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "ref_*.npz")))


def load_case(fray, path):
    z = np.load(path)
    over = {}
    for kv in str(z["overrides"]).split(";"):
        k, v = kv.split("=")
        over[k] = float(v) if "." in v else int(v)
    s = open_scene(fray, str(z["scene"]), int(z["W"]), int(z["H"]), **over)
    if s.desc.environment.present and not ("env_loaded" in z.files and int(z["env_loaded"])):
        s.desc.environment.loaded = 0        # _ref has no EXR reader: unless the fixture was made with the decoded faces handed over
                                             # (oracle/make_golden.py ENV_LOADED) its cubemap stays unloaded and misses are black
    return z, s


@pytest.mark.parametrize("path", FIXTURES, ids=lambda p: os.path.basename(p)[4:-4])
def test_oracle_hit_records_equal_reference_object_code(fray, oracle, path):
    z, s = load_case(fray, path)
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
            assert np.array_equal(out[:7], want[:7])                 # lights: dist, ip, norm
        else:
            assert np.array_equal(out, want), (i, out, want)         # dist, ip, norm, u, v -- bit for bit
    s.close()


@pytest.mark.parametrize("path", FIXTURES, ids=lambda p: os.path.basename(p)[4:-4])
def test_oracle_colour_equals_reference_shaders(fray, abi, oracle, path):
    z, s = load_case(fray, path)
    img, _ = oracle.render(s.desc, abi.MODE_RENDER, seed=int(z["seed"]), threads=4)
    ref = z["image"]
    assert ref.shape == img.shape and ref.mean() > 1e-3
    assert np.array_equal(img, ref), "max abs diff %g over %d pixels" % (np.abs(img - ref).max(), int((img != ref).any(axis=2).sum()))
    s.close()
