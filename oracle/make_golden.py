#!/usr/bin/env python3
"""Generates tests/golden/ref_*.npz from oracle/_ref (the partial reference build; see
oracle/ref_glue.cpp for exactly which code in it is reference object code and which is restated).

    python oracle/make_golden.py            # needs /root/reference mounted (make ref)

One subprocess per scene: the reference's `scene` is a process-wide singleton.  Each fixture holds
  * probes: a strided set of camera rays plus seeded random rays -> id, dist, ip, norm, u, v
    computed by the reference's own Node::intersect / Light::intersect object code;
  * image: a small frame under the RNG contract (reference shaders, lights, camera; restated
    integrator loop), float32.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.environ.get("GOLDEN_OUT") or os.path.join(ROOT, "tests", "golden")      # GOLDEN_OUT: tests/test_oracle_vs_ref_fuzz.py makes throw-away fixtures

CASES = [
    # name, scene, W, H, overrides, probe stride, n random rays
    ("boxed_whitted", "boxed.fray", 64, 48, "wantAA=0", 3, 400),
    ("boxed_aa", "boxed.fray", 32, 24, "wantAA=1", 5, 0),
    ("zaphod_dof", "zaphod.fray", 48, 32, "wantAA=0;dof=1;numDOFSamples=6", 3, 200),
    ("cornell_pt", "cornell_box.fray", 48, 48, "gi=1;numPaths=8", 3, 400),
    ("smallpt_pt", "smallpt.fray", 48, 36, "gi=1;numPaths=8", 3, 400),
    ("smallpt_whitted", "smallpt.fray", 48, 36, "gi=0;wantAA=0", 5, 0),
    ("sphtri_pt", "hw12/sphtri.fray", 48, 36, "gi=1;numPaths=4", 5, 100),
    ("dragon_whitted", "hw9/dragon.fray", 48, 32, "wantAA=0", 3, 300),
    ("bokeh_dof", "hw10/bokeh.fray", 48, 36, "wantAA=0;numDOFSamples=6", 3, 300),      # Cube - Cube CSG, Layered(Refl over textured Lambert), Phong, DOF
    ("axe_whitted", "hw9/axe_test.fray", 48, 36, "wantAA=0", 3, 300),
    ("nonconvex_aa", "hw9/nonconvex.fray", 48, 36, "wantAA=1", 3, 200),
    ("csg_nested", "../tests/scenes/csg_nested.fray", 60, 45, "wantAA=0", 2, 400),       # this repository's scene: CsgOp trees three levels deep
    ("csg_deep", "../tests/scenes/csg_deep.fray", 48, 36, "wantAA=0", 2, 400),           # this repository's scene: six CsgOp levels, an operand with up to 22 intersections (introsort ties)
    ("fuzz1009_pt", "../tests/scenes/fuzz1009/scene.fray", 76, 32, "gi=1;numPaths=4", 2, 400),   # a generated scene that caught a path-tracing mismatch
    # forest with its cubemap LOADED: the reference's CubemapEnvironment::loadMaps / getEnvironment object code (environment.cpp:31-98)
    # on the faces this project's EXR decoder produced (handed to oracle/ref_glue.cpp's Bitmap::loadEXR as plain texel files)
    ("forest_env_whitted", "forest.fray", 64, 40, "wantAA=0", 3, 300),
    ("forest_env_dof", "forest.fray", 48, 30, "wantAA=0;dof=1;numDOFSamples=6", 5, 0),
]
ENV_LOADED = {"forest_env_whitted", "forest_env_dof"}
# Scenes nobody hand-picked: tests/test_fuzz_parity.py's generator (bump maps on meshes only: on a Sphere / Cube the reference reads uninitialised
# memory), written once to tests/scenes/fuzz<seed>/ and rendered with the integrator seed % 2 selects.  tests/test_oracle_vs_ref_fuzz.py runs
# 50 more of them through oracle/_ref and the oracle without keeping them.
FUZZ_SEEDS = [3001, 3002, 3003, 3004, 3006, 3009, 3010, 3014]


def fuzz_cases():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, ROOT)
    os.environ.setdefault("FRAYHIP_NO_TORCH", "1")
    import pathlib
    from test_fuzz_parity import random_scene
    import fray_amd
    out = []
    for seed in FUZZ_SEEDS:
        folder = pathlib.Path(ROOT) / "tests" / "scenes" / ("fuzz%d" % seed)
        folder.mkdir(parents=True, exist_ok=True)
        gi = seed % 2
        random_scene(np.random.default_rng(seed), folder, gi, flavour=seed % 3, bump_on=("blob",))
        s = fray_amd.Scene.parseScene(str(folder / "scene.fray"))
        W, H = s.settings.frameWidth, s.settings.frameHeight
        s.close()
        out.append(("fuzz%d_%s" % (seed, "pt" if gi else "whitted"), "../tests/scenes/fuzz%d/scene.fray" % seed, W, H, "gi=%d" % gi, 2, 300))
    return out


def dump_faces(scene, folder):
    """The cubemap faces of `scene` as this project's host layer decoded them -> <folder>/<face>.exr.f32 (int32 w, h; float32 RGB rows)."""
    sys.path.insert(0, ROOT)
    os.environ.setdefault("FRAYHIP_NO_TORCH", "1")
    import fray_amd
    s = fray_amd.Scene.parseScene(os.path.join(ROOT, "scenes", scene))
    e = s.desc.environment
    assert e.present and e.loaded, "the scene has no decodable cubemap"
    tex = np.ctypeslib.as_array(s.desc.texels, shape=(s.desc.n_texels,))
    for f, name in enumerate(("negx", "negy", "negz", "posx", "posy", "posz")):
        w, h = e.width[f], e.height[f]
        with open(os.path.join(folder, name + ".exr.f32"), "wb") as out:
            out.write(np.array([w, h], np.int32).tobytes())
            out.write(np.ascontiguousarray(tex[e.texel_offset[f]:e.texel_offset[f] + w * h * 3], np.float32).tobytes())
    s.close()



def worker(name, scene, W, H, over, stride, nrand):
    lib = C.CDLL(os.path.join(HERE, "_ref", "libfray_ref.so"))
    lib.ref_load.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_char_p]
    lib.ref_probe.argtypes = [C.c_void_p] * 3
    lib.ref_camera_ray.argtypes = [C.c_double, C.c_double, C.c_void_p, C.c_void_p]
    lib.ref_render.argtypes = [C.c_void_p, C.c_uint]
    rc = lib.ref_load(os.path.join(ROOT, "scenes", scene).encode(), W, H, over.encode())
    assert rc == 0, rc
    rays = []
    for y in range(0, H, stride):
        for x in range(0, W, stride):
            s, d = np.zeros(3), np.zeros(3)
            lib.ref_camera_ray(float(x), float(y), s.ctypes.data, d.ctypes.data)
            rays.append((s, d))
    # secondary-like rays: start at primary hit points, go in seeded random directions
    rng = np.random.default_rng(12345)
    base = list(rays)
    for k in range(nrand):
        s, d = base[int(rng.integers(len(base)))]
        out = np.zeros(9)
        hid = lib.ref_probe(s.ctypes.data, d.ctypes.data, out.ctypes.data)
        if hid < 0:
            continue
        v = rng.normal(size=3)
        v /= np.linalg.norm(v)
        if np.dot(v, out[4:7]) < 0:
            v = -v
        rays.append((out[1:4] + out[4:7] * 1e-6, v))
    S = np.array([r[0] for r in rays])
    D = np.array([r[1] for r in rays])
    ids = np.zeros(len(rays), np.int32)
    rec = np.zeros((len(rays), 9))
    for i in range(len(rays)):
        s, d = np.ascontiguousarray(S[i]), np.ascontiguousarray(D[i])
        ids[i] = lib.ref_probe(s.ctypes.data, d.ctypes.data, rec[i].ctypes.data)
    img = np.zeros((H, W, 3), np.float32)
    lib.ref_render(img.ctypes.data, 42)
    np.savez_compressed(os.path.join(OUT, "ref_%s.npz" % name), scene=scene, W=W, H=H, overrides=over, seed=42,
                        ray_start=S, ray_dir=D, hit_id=ids, hit_rec=rec, image=img, env_loaded=int(name in ENV_LOADED))
    print(name, "rays", len(rays), "hits", int((ids != -1).sum()), "image mean %.4f" % img.mean(), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        a = sys.argv[1:]
        worker(a[0], a[1], int(a[2]), int(a[3]), a[4], int(a[5]), int(a[6]))
    else:
        os.makedirs(OUT, exist_ok=True)
        only = os.environ.get("GOLDEN_ONLY", "").split(",") if os.environ.get("GOLDEN_ONLY") else None
        for c in CASES + fuzz_cases():
            if only and c[0] not in only:
                continue
            env = dict(os.environ)
            env.pop("FRAY_REF_FACES_DIR", None)
            if c[0] in ENV_LOADED:
                import tempfile
                faces = tempfile.mkdtemp(prefix="fray_faces_")
                dump_faces(c[1], faces)
                env["FRAY_REF_FACES_DIR"] = faces
            subprocess.run([sys.executable, os.path.abspath(__file__)] + [str(x) for x in c], check=True, env=env)
