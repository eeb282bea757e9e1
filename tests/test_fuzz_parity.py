"""Randomly generated scenes: every geometry kind under random transforms, meshes big enough to get a
KD-tree, random materials and lights.  The HIP path must give bit-identical primary hit records and
colours within tolerance versus the oracle on all of them."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def random_mesh_obj(rng, n_tris, with_normals):
    """A blob of triangles around the origin (shared vertices, so boxes overlap and ties occur)."""
    nv = max(6, n_tris // 2)
    verts = rng.normal(size=(nv, 3)) * 1.5
    lines = ["v %.6f %.6f %.6f" % tuple(v) for v in verts]
    if with_normals:
        for v in verts:
            n = v / (np.linalg.norm(v) + 1e-9)
            lines.append("vn %.6f %.6f %.6f" % tuple(n))
        for v in verts:
            lines.append("vt %.4f %.4f" % (v[0] * 0.3 + 0.5, v[1] * 0.3 + 0.5))
    for _ in range(n_tris):
        a = int(rng.integers(nv))
        near = np.argsort(np.linalg.norm(verts - verts[a], axis=1))[1:6]
        b, c = rng.choice(near, 2, replace=False)
        if with_normals:
            lines.append("f %d/%d/%d %d/%d/%d %d/%d/%d" % (a + 1, a + 1, a + 1, b + 1, b + 1, b + 1, c + 1, c + 1, c + 1))
        else:
            lines.append("f %d %d %d" % (a + 1, b + 1, c + 1))
    return "\n".join(lines) + "\n"


def random_scene(rng, tmp, gi):
    W, H = int(rng.integers(40, 90)), int(rng.integers(30, 70))
    s = ["GlobalSettings {\n\tframeWidth %d\n\tframeHeight %d\n\tambientLight (0.15, 0.15, 0.2)\n\tmaxTraceDepth %d\n\twantAA off\n\tgi %d\n\tpathsPerPixel 3\n}" % (W, H, int(rng.integers(2, 5)), gi)]
    s.append("Camera camera {\n\tposition (%.3f, %.3f, -14)\n\tyaw %.2f\n\tpitch %.2f\n\troll %.2f\n\tfov %.1f\n\taspectRatio %.3f\n}" %
             (rng.normal() * 2, 3 + rng.normal(), rng.normal() * 8, -10 + rng.normal() * 5, rng.normal() * 5, 55 + rng.random() * 30, W / H))
    s.append("RectLight l1 {\n\ttranslate (%.2f, 12, %.2f)\n\tscale (5, 5, 5)\n\trotate (%.1f, 0, %.1f)\n\tpower 40\n\txSubd 2\n\tySubd 2\n}" % (rng.normal() * 2, rng.normal() * 2, rng.normal() * 10, rng.normal() * 10))
    if rng.random() < 0.5:
        s.append("PointLight l2 {\n\tpos (%.2f, 9, -6)\n\tpower 60\n\tcolor (0.9, 0.8, 0.7)\n}" % (rng.normal() * 4))
    s.append("Plane floor {\n\ty -2\n\tlimit 40\n}")
    s.append("Sphere ball {\n\tR 1.4\n\tO (0.2, 0.1, -0.1)\n}")
    s.append("Cube box {\n\thalfSide 1.2\n}")
    s.append("Cube box2 {\n\thalfSide 1.0\n\tO (0.7, 0.6, 0.5)\n}")
    s.append("CsgMinus carved {\n\tleft box\n\tright ball\n}")
    s.append("CsgAnd lens {\n\tleft ball\n\tright box2\n}")
    (tmp / "m1.obj").write_text(random_mesh_obj(rng, int(rng.integers(30, 120)), True))
    (tmp / "m2.obj").write_text(random_mesh_obj(rng, int(rng.integers(4, 18)), False))
    s.append('Mesh blob {\n\tfile "m1.obj"\n\tbackfaceCulling %s\n}' % ("false" if rng.random() < 0.5 else "true"))
    s.append('Mesh shard {\n\tfile "m2.obj"\n\tbackfaceCulling false\n}')
    s.append("CheckerTexture chk {\n\tcolor1 (0.8, 0.7, 0.2)\n\tcolor2 (0.1, 0.2, 0.6)\n\tscaling %.2f\n}" % (0.5 + rng.random() * 4))
    s.append("Lambert lam {\n\ttexture chk\n}")
    s.append("Lambert grey {\n\tcolor (0.6, 0.6, 0.6)\n}")
    s.append("Phong ph {\n\tcolor (0.8, 0.3, 0.2)\n\tspecularExponent %.0f\n}" % (10 + rng.random() * 80))
    s.append("Refl mir {\n\tmultiplier 0.85\n}")
    s.append("Refr glass {\n\tior 1.4\n\tmultiplier 0.9\n}")
    shaders = ["lam", "grey", "ph", "mir", "glass"] if not gi else ["lam", "grey", "mir", "glass"]
    geoms = ["ball", "box", "carved", "lens", "blob", "shard"]
    s.append("Node floorNode {\n\tgeometry floor\n\tshader lam\n}")
    for i, g in enumerate(geoms):
        sh = shaders[int(rng.integers(len(shaders)))]
        sc = 0.6 + rng.random() * 1.2
        s.append("Node n%d {\n\tgeometry %s\n\tshader %s\n\tscale (%.3f, %.3f, %.3f)\n\trotate (%.1f, %.1f, %.1f)\n\ttranslate (%.2f, %.2f, %.2f)\n}" %
                 (i, g, sh, sc, sc * (0.7 + rng.random() * 0.6), sc, rng.random() * 360, rng.normal() * 20, rng.normal() * 20,
                  (i - 2.5) * 2.6 + rng.normal() * 0.3, rng.random() * 2, rng.normal() * 1.5))
    (tmp / "scene.fray").write_text("\n".join(s) + "\n")
    return str(tmp / "scene.fray")


@pytest.mark.parametrize("seed", range(8))
def test_random_scene_parity(fray, abi, oracle, gpu, tmp_path, seed):
    rng = np.random.default_rng(1000 + seed)
    gi = seed % 2
    path = random_scene(rng, tmp_path, gi)
    s = fray.Scene.parseScene(path)
    s.beginRender()
    ids, dist, st = s.primary_hits(stats=True)
    oi, od, ost = oracle.render(s.desc, abi.MODE_PRIMARY_ID)
    assert len(np.unique(oi)) >= 4
    assert np.array_equal(ids, oi), np.argwhere(ids != oi)[:5]
    assert np.array_equal(dist, od), np.argwhere(dist != od)[:5]
    for k in ("node_tests", "tri_tests", "kd_inner_visits", "leaf_refs", "prim_tests"):
        assert st[k] == ost[k], k
    img, _ = s.render(seed=seed)
    ref, _ = oracle.render(s.desc, abi.MODE_RENDER, seed=seed)
    assert np.all(np.isfinite(img))
    r = np.sqrt(((img.astype(np.float64) - ref) ** 2).mean(axis=(0, 1)))
    assert np.all(r <= 1e-4), r
    s.close()
