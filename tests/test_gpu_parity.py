"""Parity of the HIP path with the oracle, through the C ABI, on a real MI355X.

Bars (BASELINE.json north_star): primary-ray hit records bit-exact; shaded colour within 1e-4
per-channel RMS at a fixed RNG-contract seed."""
import json
import os

import numpy as np
import pytest

from conftest import ROOT, bucket_xy, open_scene

pytestmark = pytest.mark.gpu
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "primary_hashes.json")))
RMS_TOL = 1e-4          # per channel, north_star
COUNTERS = ["closest_rays", "shadow_rays", "node_tests", "kd_inner_visits", "leaf_refs", "tri_tests", "prim_tests",
            "smooth_hits", "samples", "texture_fetches"]


def rms(a, b):
    return np.sqrt(((a.astype(np.float64) - b.astype(np.float64)) ** 2).mean(axis=(0, 1)))


@pytest.mark.parametrize("case", GOLD["cases"] + GOLD.get("cases_ref", []), ids=lambda c: "%s-%dx%d" % (c["scene"], c["w"], c["h"]))
def test_primary_hits_match_reference_hashes_at_full_size(fray, oracle, gpu, case):
    s = open_scene(fray, case["scene"], case["w"], case["h"], wantAA=0)
    s.beginRender()
    ids, dist, _ = s.primary_hits()
    assert int((ids != -1).sum()) == case["hits"]
    assert oracle.fnv(ids) == case["id"] and oracle.fnv(dist) == case["dist"]
    s.close()


@pytest.mark.parametrize("scene,W,H", [("boxed.fray", 100, 75), ("forest.fray", 97, 61), ("smallpt.fray", 1, 1),
                                       ("hw9/dragon.fray", 150, 100), ("cornell_box.fray", 47, 49), ("hw12/sphtri.fray", 64, 48),
                                       ("hw10/bokeh.fray", 640, 480), ("hw9/axe_test.fray", 640, 480), ("hw9/nonconvex.fray", 640, 480)])
def test_primary_hits_bit_exact_vs_oracle_ragged_sizes(fray, abi, oracle, gpu, scene, W, H):
    s = open_scene(fray, scene, W, H, wantAA=0)
    s.beginRender()
    ids, dist, st = s.primary_hits(stats=True)
    oi, od, ost = oracle.render(s.desc, abi.MODE_PRIMARY_ID)
    assert np.array_equal(ids, oi) and np.array_equal(dist, od)
    for k in COUNTERS:
        assert st[k] == ost[k], k                      # same work, not only the same answer
    s.close()


def test_bucket_partition_covers_the_frame_exactly_once(fray, abi, oracle, gpu):
    s = open_scene(fray, "boxed.fray", 200, 130, wantAA=0)
    s.beginRender()
    full, fd, _ = s.primary_hits()
    seen = np.zeros(full.shape, np.int32)
    for r in range(3):
        ids, dist, _ = s.primary_hits(bucket_first=r, bucket_stride=3)
        mine = ids != -9                               # untouched pixels keep the caller's fill value
        seen += mine
        assert np.array_equal(ids[mine], full[mine]) and np.array_equal(dist[mine], fd[mine])
    assert np.all(seen == 1)
    s.close()


WHITTED = [
    ("zaphod.fray", 320, 180, dict(wantAA=0, dof=0)),
    ("boxed.fray", 160, 120, dict(wantAA=0)),                               # KD meshes, 2 RectLights x 16 samples, Phong, bump
    ("boxed.fray", 61, 47, dict(wantAA=1)),                                 # the 5-sample AA table (main.cpp:55-61)
    ("forest.fray", 160, 120, dict(wantAA=0, dof=1, numDOFSamples=8, interactive=0)),   # thin lens + Phong + checker + bump
    ("zaphod.fray", 129, 86, dict(wantAA=1, dof=1, numDOFSamples=12)),     # as shipped, fewer samples
    ("hw12/sphtri.fray", 64, 48, dict(gi=0, wantAA=0)),                    # 3 x 225 light samples: 1350 random words per hit
    ("hw9/dragon.fray", 96, 64, dict(wantAA=0)),                            # glossy floor: 25 / 3 rejection-sampled reflections, recursion
    ("smallpt.fray", 96, 72, dict(gi=0, wantAA=0)),                         # mirror + glass spheres under Whitted
    ("smallpt.fray", 40, 30, dict(gi=0, wantAA=1, maxTraceDepth=2)),
    ("forest.fray", 96, 72, dict(wantAA=0, interactive=0, stereoSeparation=0.25)),   # anaglyph: two eyes, saturation 0.1, colour masks
    ("hw10/bokeh.fray", 96, 72, dict(wantAA=0, numDOFSamples=8)),           # as shipped but 8 lens samples: Cube - Cube CSG floor, Layered, Phong, cubemap
    ("hw9/axe_test.fray", 128, 96, dict()),                                  # as shipped (AA on): two KD meshes, checker floor, cubemap
    ("hw9/nonconvex.fray", 128, 96, dict()),                                 # as shipped: one small mesh over a checker floor (the other four figures are commented out in the file)
]


@pytest.mark.parametrize("scene,W,H,over", WHITTED, ids=lambda v: v if isinstance(v, str) else None)
def test_whitted_colour_vs_oracle(fray, abi, oracle, gpu, scene, W, H, over):
    s = open_scene(fray, scene, W, H, **over)
    s.beginRender()
    img, _ = s.render(seed=42)
    ref, ost = oracle.render(s.desc, abi.MODE_RENDER, seed=42)
    assert np.all(np.isfinite(img)) and ref.mean() > 1e-3
    assert np.all(rms(img, ref) <= RMS_TOL), rms(img, ref)
    same = float((img == ref).all(axis=2).mean())      # beyond the tolerance: the reference's evaluation order + correctly rounded trig
    print("%s %s: %.2f %% of the pixels bit-identical to the oracle" % (scene, over, 100 * same))
    assert same == 1.0, same                           # every Whitted case is the oracle's picture bit for bit; anything less is a regression
    img2, st = s.render(seed=42, stats=True)
    assert np.array_equal(img, img2)                   # instrumented kernels: same image
    for k in COUNTERS:
        assert st[k] == ost[k], k
    s.close()


PT = [
    ("cornell_box.fray", 96, 96, dict(numPaths=16)),
    ("cornell_box.fray", 50, 50, dict(numPaths=5, maxTraceDepth=2)),
    ("smallpt.fray", 96, 72, dict(numPaths=16)),                           # spheres, mirror + glass (Refl / Refr spawnRay)
    ("hw12/sphtri.fray", 96, 72, dict(numPaths=8)),                        # three RectLights
    ("zaphod.fray", 64, 43, dict(gi=1, numPaths=6, dof=1)),                # PointLight only: NEE contributes nothing; DOF + gi
    ("boxed.fray", 48, 36, dict(gi=1, numPaths=4)),                        # Phong under gi: the reference's default red BRDF
    ("cornell_box.fray", 60, 60, dict(numPaths=6, stereoSeparation=12.0)),  # anaglyph + gi: the right eye continues the left eye's streams
    ("smallpt.fray", 64, 48, dict(numPaths=5, stereoSeparation=1.5, dof=1, saturation=0.4)),
]


@pytest.mark.parametrize("scene,W,H,over", PT, ids=lambda v: v if isinstance(v, str) else None)
def test_path_traced_colour_vs_oracle(fray, abi, oracle, gpu, scene, W, H, over):
    s = open_scene(fray, scene, W, H, gi=1, **{k: v for k, v in over.items() if k != "gi"})
    s.beginRender()
    img, st = s.render(seed=42, stats=True)
    ref, ost = oracle.render(s.desc, abi.MODE_RENDER, seed=42)
    assert np.all(np.isfinite(img))
    if scene != "zaphod.fray":                          # PointLight only: the reference's path tracer sees no light at all
        assert ref.mean() > 1e-3
    else:
        assert ref.max() == 0 and img.max() == 0
    assert np.all(rms(img, ref) <= RMS_TOL), rms(img, ref)
    # Beyond the tolerance: with the samplers' correctly rounded sin / cos / acos (glibc's values in 99.85 % of calls) and the radiance
    # terms added innermost-first like the reference's recursion, a path-traced frame equals the oracle's BIT FOR BIT except where a
    # last-place difference of a direction flips a branch
    same = float((img == ref).all(axis=2).mean())
    print("%s %s: %.2f %% of the pixels bit-identical to the oracle" % (scene, over, 100 * same))
    assert same == 1.0, same                           # all eight cases, at this seed: a drop to 99.9 % is a regression, not noise
    eyes = 2 if s.camera.stereoSeparation > 0 else 1
    assert st["samples"] == ost["samples"] == W * H * s.samples_per_pixel() * eyes
    # The device's sin / cos / acos are correctly rounded, glibc's are in 99.85 % of calls: a direction may differ in its last place without
    # changing any hit.  Rays, node tests, KD visits and leaf references are the oracle's exactly; only the count of triangle tests moves by a
    # few per million (a box test at an edge going the other way)
    print("   work counters GPU - oracle:", {k: int(st[k]) - int(ost[k]) for k in COUNTERS})
    for k in COUNTERS:
        if k == "tri_tests":
            assert abs(st[k] - ost[k]) <= 2e-5 * ost[k] + 2, k
        else:
            assert st[k] == ost[k], k
    s.close()


LAYERED_SCENE = """
GlobalSettings {
	frameWidth 96
	frameHeight 72
	ambientLight (0.1, 0.1, 0.15)
	maxTraceDepth 4
	wantAA off
}
Camera camera {
	position (0, 6, -14)
	pitch -15
	fov 60
}
RectLight l1 {
	translate (0, 14, 0)
	scale (6, 6, 6)
	power 30
	xSubd 2
	ySubd 2
}
Plane floor {
	limit 30
}
Sphere ball {
	R 2
}
CheckerTexture checker {
	color1 (0.9, 0.2, 0.1)
	color2 (0.1, 0.2, 0.9)
	scaling 2
}
Lambert diffuse {
	texture checker
}
Phong shiny {
	color (0.3, 0.6, 0.3)
	specularExponent 40
}
Refl mirror {
	multiplier 0.9
}
Refl rough {
	glossiness 0.7
	numSamples 5
}
Refr refraction {
	ior 1.5
	multiplier 0.95
}
Fresnel fresnel {
	ior 1.5
}
Layered glass {
	layer refraction (1, 1, 1)
	layer mirror (1, 1, 1) fresnel
}
Layered lacquer {
	layer diffuse (1, 1, 1)
	layer shiny (0.3, 0.3, 0.3)
	layer rough (0.1, 0.15, 0.2)
}
Layered nested {
	layer lacquer (1, 1, 1)
	layer glass (0.5, 0.5, 0.5)
}
Node floorNode {
	geometry floor
	shader lacquer
}
Node glassBall {
	geometry ball
	shader glass
	translate (-3, 2, 0)
}
Node nestedBall {
	geometry ball
	shader nested
	translate (3, 2, 1)
	scale (1, 1.5, 1)
	rotate (30, 10, 0)
}
"""


def test_whitted_layered_glossy_refraction_recursion_vs_oracle(fray, abi, oracle, gpu, tmp_path):
    f = tmp_path / "layered.fray"
    f.write_text(LAYERED_SCENE)
    s = fray.Scene.parseScene(str(f))
    s.beginRender()
    img, st = s.render(seed=42, stats=True)
    ref, ost = oracle.render(s.desc, abi.MODE_RENDER, seed=42)
    assert ref.mean() > 0.05 and np.all(np.isfinite(img))
    assert np.all(rms(img, ref) <= RMS_TOL), rms(img, ref)
    for k in ("samples",):
        assert st[k] == ost[k]
    print("   work counters GPU - oracle:", {k: int(st[k]) - int(ost[k]) for k in ("closest_rays", "shadow_rays", "node_tests", "tri_tests")})
    for k in ("closest_rays", "shadow_rays", "node_tests"):
        assert st[k] == ost[k], k
    s.close()


CSG_SCENE = """
GlobalSettings {
	frameWidth 120
	frameHeight 90
	ambientLight (0.2, 0.2, 0.25)
	maxTraceDepth 4
	wantAA off
}
Camera camera {
	position (0, 9, -22)
	pitch -18
	fov 60
}
RectLight l1 {
	translate (2, 20, -4)
	scale (8, 8, 8)
	power 60
	xSubd 2
	ySubd 2
}
Cube c0 {
	halfSide 40
}
Cube c1 {
	halfSide 40.001
	O (0, -4, 0)
}
CsgMinus slab {
	left c0
	right c1
}
Cube box {
	O (0, 0, 0)
	halfSide 2
}
Sphere ball {
	R 2.6
}
Sphere small {
	O (1.2, 1.0, -1.0)
	R 1.6
}
Mesh dice {
	file "geom/truncated_cube.obj"
	faceted true
}
CsgAnd rounded {
	left box
	right ball
}
CsgMinus bitten {
	left ball
	right small
}
CsgPlus fused {
	left dice
	right small
}
CheckerTexture checker {
	color1 (0.8, 0.8, 0.2)
	color2 (0.2, 0.2, 0.8)
	scaling 3
}
Lambert tiles {
	texture checker
}
Phong red {
	color (0.9, 0.2, 0.2)
	specularExponent 30
}
Lambert grey {
	color (0.6, 0.6, 0.6)
}
Refl mirror {
	multiplier 0.8
}
Node floorNode {
	geometry slab
	shader tiles
	translate (0, -40, 0)
}
Node a {
	geometry rounded
	shader red
	translate (-6, 2.5, 0)
	rotate (30, 20, 0)
}
Node b {
	geometry bitten
	shader grey
	translate (0, 3, 2)
}
Node c {
	geometry fused
	shader mirror
	translate (6.5, 2.2, 0)
	scale (0.5, 0.5, 0.5)
	rotate (15, 0, 10)
}
Node plainCube {
	geometry box
	shader tiles
	translate (0, 2, -7)
	rotate (45, 0, 0)
}
"""


@pytest.mark.parametrize("gi", [0, 1])
def test_cube_and_csg_geometry_vs_oracle(fray, abi, oracle, gpu, tmp_path, gi):
    import shutil
    os.makedirs(tmp_path / "geom")
    shutil.copy(os.path.join(ROOT, "scenes", "geom", "truncated_cube.obj"), tmp_path / "geom")
    f = tmp_path / "csg.fray"
    f.write_text(CSG_SCENE)
    s = fray.Scene.parseScene(str(f))
    s.settings.gi, s.settings.numPaths = gi, 6
    s.beginRender()
    ids, dist, st = s.primary_hits(stats=True)
    oi, od, ost = oracle.render(s.desc, abi.MODE_PRIMARY_ID)
    assert set(np.unique(oi)) >= {0, 1, 2, 3, 4}                    # every node is visible
    assert np.array_equal(ids, oi) and np.array_equal(dist, od)     # bit-exact hit records through CSG
    for k in COUNTERS:
        assert st[k] == ost[k], k
    img, _ = s.render(seed=42)
    ref, _ = oracle.render(s.desc, abi.MODE_RENDER, seed=42)
    assert ref.mean() > 0.02 and np.all(np.isfinite(img))
    assert np.all(rms(img, ref) <= RMS_TOL), rms(img, ref)
    s.close()


def test_path_tracer_batching_and_seed_properties(fray, gpu):
    s = open_scene(fray, "cornell_box.fray", 80, 60, numPaths=7)
    s.beginRender()
    a, _ = s.render(seed=42)
    b, _ = s.render(seed=42, spp_chunk=1)              # one sample per batch
    c, _ = s.render(seed=42, spp_chunk=3)              # ragged last batch
    assert np.array_equal(a, b) and np.array_equal(a, c)
    d, _ = s.render(seed=7)
    assert not np.array_equal(a, d)
    acc = np.zeros_like(a)
    for r in range(2):
        part = np.zeros_like(a)
        s.render(seed=42, bucket_first=r, bucket_stride=2, out=part)
        acc += part
    assert np.array_equal(acc, a)                      # any tiling gives the same pixels
    s.close()


def test_full_size_properties_cornell_1080p(fray, oracle, abi, gpu):
    """At BASELINE's full frame size the oracle is too slow for whole-frame comparison: check
    determinism, tiling invariance and a strip of buckets against the oracle."""
    s = open_scene(fray, "cornell_box.fray", 1920, 1080, numPaths=4)
    s.beginRender()
    a, st = s.render(seed=42)
    b, _ = s.render(seed=42)
    assert np.array_equal(a, b) and np.all(np.isfinite(a))
    part = np.zeros_like(a)
    s.render(seed=42, bucket_first=5, bucket_stride=8, out=part)
    m = part.any(axis=2)
    assert np.array_equal(part[m], a[m])
    ref, _ = oracle.render(s.desc, abi.MODE_RENDER, seed=42, bucket_first=17, bucket_stride=40)   # 23 buckets
    mr = ref.any(axis=2)
    assert mr.sum() > 40000
    d = (a[mr].astype(np.float64) - ref[mr]) ** 2
    assert np.all(np.sqrt(d.mean(axis=0)) <= RMS_TOL)
    s.close()


def test_device_rng_matches_libstdcxx(fray, oracle, gpu):
    n = 1500          # doubles draw 3000 words: register window, materialised state, four twists
    for seed, hi in [(42, 0), (1, 15), (0xdeadbeef, 35), (123456789, 2), (7, 224)]:
        f, d, i = np.zeros(n, np.float32), np.zeros(n, np.float64), np.zeros(n, np.int32)
        assert fray.lib.frayhip_debug_rng(seed, n, f.ctypes.data, d.ctypes.data, i.ctypes.data, hi) == 0
        of, od, oi = np.zeros(n, np.float32), np.zeros(n, np.float64), np.zeros(n, np.int32)
        oracle.lib.fray_oracle_rng_stream(seed, n, of.ctypes.data, od.ctypes.data, oi.ctypes.data, hi)
        assert np.array_equal(f, of) and np.array_equal(d, od) and np.array_equal(i, oi)


def test_pack_unpack_buckets(fray, gpu):
    import torch
    W, H, world = 200, 130, 3
    frame = torch.rand((H, W, 3), device="cuda")
    out = torch.zeros_like(frame)
    for r in range(world):
        nb = fray.lib.frayhip_bucket_count(W, H, r, world)
        packed = torch.full((nb * 2304 * 3,), -1.0, device="cuda")
        assert fray.lib.frayhip_pack_buckets_device(frame.data_ptr(), packed.data_ptr(), W, H, 3, r, world, None) == 0
        assert fray.lib.frayhip_unpack_buckets_device(packed.data_ptr(), out.data_ptr(), W, H, 3, r, world, None) == 0
        torch.cuda.synchronize()
        p = packed.cpu().numpy().reshape(nb, 48, 48, 3)
        BW = (W - 1) // 48 + 1
        f = frame.cpu().numpy()
        for k in range(nb):
            b = r + k * world
            bx, by = bucket_xy(W, b)
            tile = np.zeros((48, 48, 3), np.float32)
            sub = f[by * 48:by * 48 + 48, bx * 48:bx * 48 + 48]
            tile[:sub.shape[0], :sub.shape[1]] = sub
            assert np.array_equal(p[k], tile)
    assert torch.equal(out, frame)


@pytest.mark.parametrize("gi", [0, 1])
def test_nested_csg_vs_oracle(fray, abi, oracle, gpu, gi):
    """CsgOp trees up to three levels deep (tests/scenes/csg_nested.fray; the oracle equals the reference's own
    CsgOp code on it: tests/golden/ref_csg_nested.npz)."""
    s = fray.Scene.parseScene(os.path.join(ROOT, "tests", "scenes", "csg_nested.fray"))
    s.settings.gi, s.settings.numPaths = gi, 5
    s.beginRender()
    ids, dist, st = s.primary_hits(stats=True)
    oi, od, ost = oracle.render(s.desc, abi.MODE_PRIMARY_ID)
    assert set(np.unique(oi)) >= {0, 1, 2, 3}
    assert np.array_equal(ids, oi) and np.array_equal(dist, od)
    for k in COUNTERS:
        assert st[k] == ost[k], k
    img, _ = s.render(seed=42)
    ref, _ = oracle.render(s.desc, abi.MODE_RENDER, seed=42)
    assert ref.mean() > 0.02 and np.all(np.isfinite(img))
    assert np.all(rms(img, ref) <= RMS_TOL), rms(img, ref)
    s.close()


def _csg_chain(tmp_path, levels):
    """A chain of `levels` CsgOp nodes, each level the union / difference / intersection of the level below and a sphere or cube of its own."""
    ops = ["CsgPlus", "CsgMinus", "CsgAnd"]
    txt = ("GlobalSettings {\n\tframeWidth 64\n\tframeHeight 48\n\tambientLight (0.3, 0.3, 0.3)\n}\nPointLight {\n\tpos (30, 40, -60)\n\tpower 12000\n}\n"
           "Camera camera {\n\tposition (0, 1, -9)\n\tfov 60\n\taspectRatio 1.3333\n}\nCube a {\n\thalfSide 2\n}\n")
    for k in range(1, levels + 1):
        txt += "Sphere s%d {\n\tO (%g, %g, %g)\n\tR %g\n}\n" % (k, 0.37 * (k % 5) - 0.8, 0.29 * (k % 3) - 0.3, -1.6 + 0.11 * k, 0.9 + 0.07 * (k % 4))
        txt += "%s l%d {\n\tleft %s\n\tright s%d\n}\n" % (ops[k % 3] if k % 4 else "CsgPlus", k, "a" if k == 1 else "l%d" % (k - 1), k)
    txt += "Lambert lam {\n\tcolor (0.8, 0.7, 0.6)\n}\nNode n {\n\tgeometry l%d\n\tshader lam\n}\n" % levels
    f = tmp_path / ("chain%d.fray" % levels)
    f.write_text(txt)
    return str(f)


@pytest.mark.parametrize("levels", [9, 16])
def test_csg_chains_up_to_sixteen_levels_vs_oracle(fray, abi, oracle, gpu, tmp_path, levels):
    """CsgOp::intersect recurses without bound in the reference (geometry.cpp:139-194); the device unrolls sixteen levels (round 2: eight).  Hit
    records, counters and the Whitted picture of a 9- and a 16-level chain equal the oracle's."""
    s = fray.Scene.parseScene(_csg_chain(tmp_path, levels))
    s.beginRender()
    ids, dist, st = s.primary_hits(stats=True)
    oi, od, ost = oracle.render(s.desc, abi.MODE_PRIMARY_ID)
    assert (oi == 0).sum() > 100                                 # the chain's solid is in the picture
    assert np.array_equal(ids, oi) and np.array_equal(dist, od)
    for k in COUNTERS:
        assert st[k] == ost[k], k
    img, _ = s.render(seed=42)
    ref, _ = oracle.render(s.desc, abi.MODE_RENDER, seed=42)
    assert np.array_equal(img, ref)
    s.close()


def test_unsupported_features_fail_loudly(fray, abi, gpu, tmp_path):
    s = fray.Scene.parseScene(_csg_chain(tmp_path, 17))
    with pytest.raises(fray.FrayError) as e:
        s.beginRender()                                         # seventeen CsgOp levels: one more than the device unrolls
    assert e.value.code == abi.E_UNSUPPORTED
    s3 = open_scene(fray, "boxed.fray", 32, 32)
    with pytest.raises(fray.FrayError):
        s3.render()                                             # beginRender() not called


@pytest.mark.parametrize("gi", [0, 1])
def test_deep_csg_and_many_intersections_vs_oracle(fray, abi, oracle, gpu, gi):
    """Six CsgOp levels and an operand a ray crosses up to 22 times (tests/scenes/csg_deep.fray): more than 16 intersections per sort,
    so ties are ordered by libstdc++'s introsort, which dev_sort.hpp restates; the oracle equals the reference's own CsgOp object
    code on this scene (tests/golden/ref_csg_deep.npz)."""
    s = fray.Scene.parseScene(os.path.join(ROOT, "tests", "scenes", "csg_deep.fray"))
    s.settings.gi, s.settings.numPaths = gi, 4
    s.beginRender()
    ids, dist, st = s.primary_hits(stats=True)
    oi, od, ost = oracle.render(s.desc, abi.MODE_PRIMARY_ID)
    assert set(np.unique(oi)) >= {0, 1, 2}
    assert np.array_equal(ids, oi) and np.array_equal(dist, od)
    for k in COUNTERS:
        assert st[k] == ost[k], k
    img, _ = s.render(seed=42)
    ref, _ = oracle.render(s.desc, abi.MODE_RENDER, seed=42)
    assert ref.mean() > 0.02 and np.all(np.isfinite(img))
    if gi:
        # Phong under gi is the reference's default red BRDF with pdf 1 (shading.h:124-134): a single diffuse bounce off the floor that
        # grazes the red solid's silhouette carries a contribution of ~3 per sample, and whether it grazes follows the last bit of the
        # bounce direction.  Every pixel but a handful must match; those few are counted, not averaged away.
        diff = np.abs(img.astype(np.float64) - ref)
        bad = (diff > 1e-4).any(axis=2)
        assert bad.sum() <= 4 and np.sqrt((diff[~bad] ** 2).mean()) <= 1e-6, int(bad.sum())
    else:
        assert np.all(rms(img, ref) <= RMS_TOL), rms(img, ref)
    s.close()


import glob as _glob

REF_FIXTURES = sorted(_glob.glob(os.path.join(ROOT, "tests", "golden", "ref_*.npz")))


@pytest.mark.parametrize("path", REF_FIXTURES, ids=lambda p: os.path.basename(p)[4:-4])
def test_gpu_colour_vs_reference_fixture(fray, gpu, path):
    """HIP render vs the committed image produced by the partial reference build (reference object
    code for shaders / lights / camera / geometry; tests/golden/ref_*.npz, oracle/make_golden.py)."""
    from test_oracle_vs_ref import load_case
    z, s = load_case(fray, path)
    s.beginRender()
    img, _ = s.render(seed=int(z["seed"]))
    if os.path.basename(path) == "ref_fuzz1009_pt.npz":
        # twin triangles of opposite orientation: a few paths follow the last bit of libm (see
        # test_fuzz_parity.py::test_coincident_opposite_triangles); every other pixel must match
        diff = np.abs(img.astype(np.float64) - z["image"])
        bad = (diff > 1e-5).any(axis=2)
        print("ref_fuzz1009_pt: %d of %d pixels differ from the reference fixture" % (int(bad.sum()), bad.size))
        assert bad.sum() == 0 and np.array_equal(img, z["image"]), int(bad.sum())     # since the device's own trig (round 2): every pixel
        s.close()
        return
    assert np.all(rms(img, z["image"]) <= RMS_TOL), rms(img, z["image"])
    # and beyond: the fixture is the output of the reference's own shader / light / camera / geometry object code
    same = float((img == z["image"]).all(axis=2).mean())
    print("%s: %.2f %% of the pixels bit-identical to the reference fixture" % (os.path.basename(path), 100 * same))
    if os.path.basename(path).startswith("ref_fuzz3"):
        # generated scenes (oracle/make_golden.py FUZZ_SEEDS): thin lenses and glossy samples take sin / cos of random angles, where glibc's sincos() is
        # not correctly rounded in 0.14 % of calls and the device's functions are: last-place differences in a few pixels, nothing more
        # Pinned per fixture (tests/golden/ref_fuzz3_last_place_pixels.json: how many pixels differ from the reference's picture at all, measured on the GPU
        # when the fixture was made; the device code is deterministic): a drift inside the tolerance shows up as another count.
        import json
        known = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_fuzz3_last_place_pixels.json")))
        n_diff = int((img != z["image"]).any(axis=2).sum())
        print("%s: %d pixels differ in the last place" % (os.path.basename(path), n_diff))
        assert np.all(np.abs(img.astype(np.float64) - z["image"]) <= 1e-5 * np.maximum(1.0, np.abs(z["image"])))
        assert n_diff == known[os.path.basename(path)], (n_diff, known[os.path.basename(path)])
    else:
        assert same == 1.0, same
    s.close()


@pytest.mark.parametrize("scene,W,H,over,strip", [
    ("smallpt.fray", 4096, 4096, dict(gi=1, numPaths=2), (123, 500)),                       # BASELINE configs[4] frame size
    ("forest.fray", 1920, 1080, dict(wantAA=0, dof=1, numDOFSamples=4, interactive=0), (7, 40)),   # configs[3]: KD meshes, DOF, cubemap
    ("zaphod.fray", 1920, 1080, dict(wantAA=0, dof=0), (11, 40)),                             # configs[1]
])
def test_full_size_frames_of_the_other_configs(fray, abi, oracle, gpu, scene, W, H, over, strip):
    """Full BASELINE frame sizes: determinism plus a strided strip of buckets against the oracle."""
    s = open_scene(fray, scene, W, H, **over)
    s.beginRender()
    a, _ = s.render(seed=42)
    b, _ = s.render(seed=42)
    assert np.array_equal(a, b) and np.all(np.isfinite(a))
    first, stride = strip
    ref, _ = oracle.render(s.desc, abi.MODE_RENDER, seed=42, bucket_first=first, bucket_stride=stride)
    BW, BH = (W - 1) // 48 + 1, (H - 1) // 48 + 1
    mask = np.zeros((H, W), bool)
    for bk in range(first, BW * BH, stride):
        bx, by = bucket_xy(W, bk)
        mask[by * 48:by * 48 + 48, bx * 48:bx * 48 + 48] = True
    assert mask.sum() > 20000
    d = (a[mask].astype(np.float64) - ref[mask]) ** 2
    assert np.all(np.sqrt(d.mean(axis=0)) <= RMS_TOL), np.sqrt(d.mean(axis=0))
    s.close()


def test_render_device_into_torch_tensor_on_a_side_stream(fray, abi, gpu):
    """frayhip_render_device: caller-owned device memory + a caller stream (what bench.py uses)."""
    import torch
    s = open_scene(fray, "cornell_box.fray", 96, 64, numPaths=5)
    s.beginRender()
    host, st0 = s.render(seed=42)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        frame = torch.full((64, 96, 3), -1.0, dtype=torch.float32, device="cuda")
        st = s.render_device(frame.data_ptr(), seed=42, stream=side.cuda_stream)
    side.synchronize()
    assert np.array_equal(frame.cpu().numpy(), host)
    assert st["trace_launches"] == st0["trace_launches"] > 0 and st["ms_trace"] > 0
    # hit records into device tensors as well
    ids = torch.zeros((64, 96), dtype=torch.int32, device="cuda")
    dist = torch.zeros((64, 96), dtype=torch.float64, device="cuda")
    s.render_device(None, mode=abi.MODE_PRIMARY_ID, d_id_ptr=ids.data_ptr(), d_dist_ptr=dist.data_ptr(), stream=side.cuda_stream)
    hi, hd, _ = s.primary_hits()
    assert np.array_equal(ids.cpu().numpy(), hi) and np.array_equal(dist.cpu().numpy(), hd)
    s.close()


def test_camera_and_settings_can_change_between_frames(fray, abi, oracle, gpu):
    s = open_scene(fray, "boxed.fray", 80, 60, wantAA=0)
    s.beginRender()
    a, _, _ = s.primary_hits()
    s.camera.yaw += 25.0
    s.camera.pos[1] += 3.0
    s.settings.frameWidth, s.settings.frameHeight = 64, 64
    s.beginFrame()
    b, bd, _ = s.primary_hits()
    oi, od, _ = oracle.render(s.desc, abi.MODE_PRIMARY_ID)
    assert b.shape == (64, 64) and np.array_equal(b, oi) and np.array_equal(bd, od)
    assert a.shape == (60, 80)
    s.close()


def test_cxx_host_example_renders_like_the_python_path(fray, gpu, tmp_path):
    """examples/fray_render: a C++ main() over the C ABI (no Python, no torch in that process)."""
    from conftest import run_in_clean_child
    exe = os.path.join(ROOT, "examples", "fray_render")
    if not os.path.exists(exe):
        pytest.fail("examples/fray_render is not built (make)")
    out = tmp_path / "c.bmp"
    # started from the fork server, not from this (GPU-initialised) process: conftest.py
    log = run_in_clean_child([exe, os.path.join(ROOT, "scenes", "cornell_box.fray"), str(out), "64", "48", "4"], str(tmp_path / "c.log"), timeout=300)
    assert "[exit code 0]" in log, log
    assert "Render took" in log and "Exited cleanly" in log
    s = open_scene(fray, "cornell_box.fray", 64, 48, numPaths=4)
    s.beginRender()
    img, _ = s.render(seed=42)
    ref = tmp_path / "p.bmp"
    assert fray.lib.frayhip_save_bmp(str(ref).encode(), img.ctypes.data, 64, 48) == 0
    assert open(out, "rb").read() == open(ref, "rb").read()
    s.close()


def test_device_libm_vs_glibc(fray, gpu):
    """sin / cos / acos are the reference's third-party arithmetic (glibc).  The device code carries its own, correctly rounded in
    all but a few calls per million (fray_amd/csrc/dev_trig.hpp, checked against 113-bit arithmetic by tests/native/trig_check.cpp);
    glibc's are the correctly rounded values in 99.85 % of calls, so the two must be IDENTICAL in more than 99.5 % of calls on the
    arguments the samplers produce (theta in [0, 2 pi), 2v-1 in [-1, 1)) and never more than one place apart.  (ROCm's own functions
    matched glibc in 96.9 % / 93.4 % of calls.)"""
    import math
    rng = np.random.default_rng(7)
    n = 1 << 16
    x = rng.random(n) * 2 * np.pi
    sn, cs, ac, arg = (np.zeros(n) for _ in range(4))
    assert fray.lib.frayhip_debug_libm(n, x.ctypes.data, sn.ctypes.data, cs.ctypes.data, ac.ctypes.data, arg.ctypes.data) == 0
    ref_arg = x - 2.0 * np.floor(x * 0.5) - 1.0
    assert np.array_equal(arg, ref_arg)

    def ulps(a, b):
        return np.abs(a.view(np.int64) - b.view(np.int64))
    # math.* calls the C library (glibc here, as in the reference build); numpy's vector loops are not glibc
    glibc = lambda f, v: np.array([f(t) for t in v.tolist()])
    for name, got, want in (("sin", sn, glibc(math.sin, x)), ("cos", cs, glibc(math.cos, x)), ("acos", ac, glibc(math.acos, arg))):
        u = ulps(got, want)
        assert u.max() <= 1, (name, int(u.max()))
        exact = float((u == 0).mean())
        print("%s: identical to glibc in %.3f %% of %d calls" % (name, 100 * exact, n))
        assert exact > 0.995, (name, exact)


@pytest.mark.parametrize("gi", [0, 1])
def test_negative_trace_depth_is_a_black_frame(fray, abi, oracle, gpu, gi):
    """raytrace() / pathtrace() return black before looking at the scene when depth 0 > maxTraceDepth (main.cpp:173-176, 248)."""
    s = open_scene(fray, "cornell_box.fray", 61, 47, gi=gi, numPaths=3, wantAA=0, maxTraceDepth=-1)
    s.beginRender()
    img, st = s.render(seed=42, stats=True)
    ref, ost = oracle.render(s.desc, abi.MODE_RENDER, seed=42)
    assert not img.any() and not ref.any()
    assert st["samples"] == ost["samples"] == 61 * 47 * s.samples_per_pixel()
    assert st["closest_rays"] == ost["closest_rays"] == 0
    s.close()


@pytest.mark.parametrize("depth,stereo", [(40, 0.0), (25, 1.5), (150, 0.0), (400, 0.0)])
def test_path_tracing_past_227_random_words_per_sample(fray, abi, oracle, gpu, depth, stereo):
    """maxTraceDepth >= 20: a path may draw more than 227 words from a generator (ten per Lambert bounce), which the three-register
    mt19937 streams do not cover; the bounce kernel then runs with per-path materialised generator state (MtPath).  The scene keeps
    throughput: paths reach the depth limit, several hundred words from each generator and, at depth 400, a second state twist."""
    s = fray.Scene.parseScene(os.path.join(ROOT, "tests", "scenes", "whitebox.fray"))
    s.settings.maxTraceDepth = depth
    s.camera.stereoSeparation = stereo
    s.beginRender()
    img, st = s.render(seed=42, stats=True)
    ref, ost = oracle.render(s.desc, abi.MODE_RENDER, seed=42)
    assert np.all(np.isfinite(img)) and ref.mean() > 0.05
    assert np.all(rms(img, ref) <= RMS_TOL), rms(img, ref)
    assert st["samples"] == ost["samples"]
    # paths really get that deep
    assert ost["closest_rays"] > 0.8 * min(depth, 300) * ost["samples"]
    img2, _ = s.render(seed=42)
    assert np.array_equal(img, img2)
    s.close()


# Kernel variant <8>: textures and an environment-free scene WITHOUT KD-tree meshes and without Cube / CSG (planes, spheres, a tree-less mesh;
# checker texture on a plane and on the sphere, whose uv come from atan2 / asin).  The variants <0> (cornell_box, smallpt) and <4> / <2>
# (everything else) are covered by the other tests; this one pins the fourth flag combination.
TEXTURED_PLAIN_SCENE = """
GlobalSettings {
	frameWidth 96
	frameHeight 72
	ambientLight (0.15, 0.15, 0.2)
	maxTraceDepth 4
	wantAA off
	%s
}
Camera camera {
	position (0, 6, -18)
	pitch -12
	fov 60
}
RectLight l1 {
	translate (2, 16, -4)
	scale (6, 6, 6)
	power 40
	xSubd 2
	ySubd 2
}
CheckerTexture chk {
	color1 (0.8, 0.7, 0.2)
	color2 (0.1, 0.2, 0.6)
	scaling 0.5
}
CheckerTexture chk2 {
	color1 (0.9, 0.9, 0.9)
	color2 (0.2, 0.6, 0.3)
	scaling 24
}
Lambert floorShader {
	color (1, 1, 1)
	texture chk
}
Lambert ballShader {
	color (1, 1, 1)
	texture chk2
}
Lambert plateShader {
	color (0.7, 0.3, 0.2)
}
Plane floor {
	y 0
	limit 60
}
Sphere ball {
	O (0, 0, 0)
	R 3
}
Mesh plates {
	file "%s"
	faceted true
	useKDTree false
}
Node floorNode {
	geometry floor
	shader floorShader
}
Node ballNode {
	geometry ball
	shader ballShader
	translate (-3, 3, 2)
	rotate (20, 35, 0)
}
Node platesNode {
	geometry plates
	shader plateShader
	translate (5, 2, 0)
	scale (2, 2, 2)
}
"""


@pytest.mark.parametrize("gi", [0, 1])
def test_textured_scene_without_kd_meshes_vs_oracle(fray, abi, oracle, gpu, tmp_path, gi):
    import shutil
    shutil.copy(os.path.join(os.path.dirname(os.path.abspath(__file__)), "scenes", "plates.obj"), tmp_path / "plates.obj")   # files are looked up next to the scene
    f = tmp_path / "textured_plain.fray"
    f.write_text(TEXTURED_PLAIN_SCENE % ("gi on\n\tpathsPerPixel 8" if gi else "gi off", "plates.obj"))
    s = fray.Scene.parseScene(str(f))
    s.beginRender()
    d = s.desc
    assert d.n_textures == 2 and not any(d.meshes[i].has_kd for i in range(d.n_meshes))     # the <8> variants' scene class
    img, st = s.render(seed=42, stats=True)
    plain = s.render(seed=42)
    plain = plain[0] if isinstance(plain, tuple) else plain
    ref, ost = oracle.render(s.desc, abi.MODE_RENDER, seed=42)
    assert ref.mean() > 0.02 and np.all(np.isfinite(img))
    assert np.array_equal(img, plain)                              # counting (<9>) and plain (<8>) kernels draw the same picture
    assert np.all(rms(img, ref) <= RMS_TOL), rms(img, ref)
    assert float((img == ref).all(axis=2).mean()) >= 0.995
    for k in ("samples", "closest_rays", "shadow_rays"):
        assert st[k] == ost[k], k
    s.close()


@pytest.mark.parametrize("what", ["normal_not_the_cross_product", "coordinates_beyond_the_filter"])
def test_kd_leaf_filter_guards_vs_oracle(fray, abi, oracle, gpu, what):
    """The certified FP32 filter of the KD leaves (fray_amd/csrc/dev_tricert.hpp) may only skip triangles the reference's arithmetic rejects.  Two
    scene descriptions where geometry and arithmetic part ways: (a) a mesh whose stored ABcrossAC is NOT the cross product of its AB and AC
    (Triangle::intersectFast, triangle.cpp:66-97, believes the stored vectors: lambda2 / lambda3 come out halved, rays 'hit' outside the triangles);
    (b) edge vectors scaled so that |AB|, |AC| leave the range the filter's FP32 products are proved for.  Both must fall back to the reference's
    arithmetic for every triangle: hit records bit-equal to the oracle's, counters equal."""
    s = open_scene(fray, "boxed.fray", 160, 120, wantAA=0)
    d = s.desc
    changed = 0
    for mi in range(d.n_meshes):
        m = d.meshes[mi]
        if not m.has_kd:
            continue
        for t in range(m.n_triangles):
            T = m.triangles[t]
            if what == "normal_not_the_cross_product":
                if t % 3 == 0:
                    for k in range(3):
                        T.ABcrossAC[k] *= 2.0
                    changed += 1
            elif t % 5 == 0:
                for k in range(3):                       # the same plane and (lambda2, lambda3) scaled by 2^-41: the oracle still hits some of them
                    T.AB[k] *= 2.0 ** 41
                    T.AC[k] *= 2.0 ** 41
                    T.ABcrossAC[k] *= 2.0 ** 82
                changed += 1
    assert changed > 1000
    s.beginRender()
    ids, dist, st = s.primary_hits(stats=True)
    rid, rdist, ost = oracle.render(s.desc, abi.MODE_PRIMARY_ID)
    assert np.array_equal(ids, rid) and np.array_equal(dist, rdist)
    for k in COUNTERS:
        assert st[k] == ost[k], k
    img, _ = s.render(seed=42)
    ref, _ = oracle.render(s.desc, abi.MODE_RENDER, seed=42)
    assert np.array_equal(img, ref)
    plain = open_scene(fray, "boxed.fray", 160, 120, wantAA=0)
    plain.beginRender()
    pid, _, _ = plain.primary_hits(stats=True)
    assert (pid != ids).mean() > 0.002                   # the change is visible: the case does exercise the guard
    plain.close()
    s.close()


@pytest.mark.parametrize("pitch,roll,yaw", [(-7.93, 7.93, 3.52), (-19.99, 1.79, -1.12), (-7.84, -2.51, 1.61)])
def test_camera_angles_where_sincos_is_not_sin_and_cos(fray, abi, oracle, gpu, pitch, roll, yaw):
    """Camera::beginFrame's rotation matrices (camera.cpp:48-49, matrix.cpp:29-62) take sin(angle) and cos(angle); the reference's g++ build merges
    each pair into glibc's sincos(), whose sine is not sin()'s in the last place for one angle in 700 -- these.  The product's camera code is compiled
    by clang, which keeps two calls, and 4 of 1 500 random scenes had hit distances one unit in the last place off until it asked for sincos() by
    name.  Hit records bit-equal to the oracle's, and to the reference object code's where that is built (oracle/_ref imports only sincos)."""
    s = open_scene(fray, "boxed.fray", 160, 120, wantAA=0, pitch=pitch, roll=roll, yaw=yaw)
    s.beginRender()
    ids, dist, st = s.primary_hits(stats=True)
    oi, od, ost = oracle.render(s.desc, abi.MODE_PRIMARY_ID)
    assert np.array_equal(ids, oi) and np.array_equal(dist, od)
    assert (oi >= 0).mean() > 0.3
    s.close()


@pytest.mark.parametrize("scene,W,H,over,fmax", [
    ("zaphod.fray", 1920, 1080, dict(wantAA=0, dof=0), 4),             # BASELINE configs[1]: one point light, nothing draws -> no k_seed, pixel written by the kernel
    ("zaphod.fray", 322, 215, dict(wantAA=1, dof=0), 4),               # five samples per pixel: the samples' colours go through k_pt_resolve
    ("zaphod.fray", 200, 130, dict(wantAA=0, dof=1, numDOFSamples=7), 4),   # the lens draws: seeded
    ("hw12/sphtri.fray", 160, 120, dict(gi=0, wantAA=0), 1024),        # three RectLights of 225 samples that draw, forced into the fused form
    ("hw12/sphtri.fray", 90, 60, dict(gi=0, wantAA=1, stereoSeparation=0.5), 1024),   # stereo: both eyes through the one copy of the code
])
def test_fused_whitted_shade_is_the_wavefronts_picture(fray, abi, oracle, gpu, scene, W, H, over, fmax):
    """Option "fused_whitted_max": k_wh_shade<ST, FUSED> (visible() in place, no queue, one launch) against the shade / visible / gather launches and the oracle."""
    s = open_scene(fray, scene, W, H, **over)
    s.beginRender()
    s.set_option("fused_whitted_max", 0)
    a, _ = s.render(seed=42)
    assert s.get_option("whitted_path") == 1
    s.set_option("fused_whitted_max", fmax)
    b, st = s.render(seed=42, stats=True)
    assert s.get_option("whitted_path") == 2 and st["shadow_launches"] == 0
    assert np.array_equal(a, b)
    ref, ost = oracle.render(s.desc, abi.MODE_RENDER, seed=42)
    assert np.array_equal(b, ref)
    assert st["shadow_rays"] == ost["shadow_rays"] and st["closest_rays"] == ost["closest_rays"]
    s.close()


@pytest.mark.parametrize("scene,W,H,over", [
    ("boxed.fray", 320, 240, dict(gi=1, numPaths=12)),                  # KD variants: first bounce, later bounces and the shadow kernel each spill another amount
    ("smallpt.fray", 256, 192, dict(gi=1, numPaths=12)),
    ("cornell_box.fray", 200, 150, dict(gi=1, numPaths=16)),
    ("../tests/scenes/csg_nested.fray", 160, 120, dict(gi=1, numPaths=6)),
])
def test_batch_lanes_never_change_the_picture(fray, gpu, scene, W, H, over):
    """A path-traced frame is cut into batches that run on up to four streams at once; the picture must be the one-lane picture every time, for the timed
    kernels, the counting kernels and the fp_contract kernels.  (Round 5 found the counting variants rendering wrong pixels in one frame in ten on three
    lanes: kernels with different amounts of spilled registers in scratch memory sharing the chip -- fray_amd/csrc/render_impl.hpp; five frames of each kind here.)"""
    s = open_scene(fray, scene, W, H, **over)
    s.beginRender()
    s.set_option("pt_lanes", 1)
    ref, _ = s.render(seed=7)
    s.set_option("pt_lanes", 4)
    for rep in range(5):
        for stats in (False, True):
            img, _ = s.render(seed=7, stats=stats)
            assert np.array_equal(img, ref), (rep, stats, int((img != ref).any(axis=2).sum()))
    s.set_option("fp_contract", 1)
    s.set_option("pt_lanes", 1)
    refc, _ = s.render(seed=7)
    s.set_option("pt_lanes", 4)
    for rep in range(5):
        img, _ = s.render(seed=7)
        assert np.array_equal(img, refc), (rep, "contracted", int((img != refc).any(axis=2).sum()))
    s.close()
