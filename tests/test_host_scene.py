"""The .fray parser / OBJ / BMP loaders behave like the reference's (scene.cpp:403-570, mesh.cpp:203-258,
bitmap.cpp:117-195) on the edge cases its grammar has."""
import ctypes as C
import os
import struct

import numpy as np
import pytest

from conftest import open_scene


def parse(fray, tmp_path, text, files=None):
    for name, data in (files or {}).items():
        p = tmp_path / name
        p.parent.mkdir(parents=True, exist_ok=True)
        p.write_bytes(data if isinstance(data, bytes) else data.encode())
    f = tmp_path / "scene.fray"
    f.write_text(text)
    return fray.Scene.parseScene(str(f))


BASE = """
Camera camera {
	position (1, 2, 3)
}
"""


def test_defaults_follow_the_reference(fray, tmp_path):
    s = parse(fray, tmp_path, BASE)
    st, cam = s.settings, s.camera
    assert (st.frameWidth, st.frameHeight, st.wantAA, st.maxTraceDepth, st.gi, st.numPaths, st.wantPrepass) == (800, 600, 1, 4, 0, 10, 1)
    assert (cam.fov, cam.aspectRatio, cam.fNumber, cam.focalPlaneDist, cam.numDOFSamples, cam.dof) == (90.0, 1.3333, 2.0, 5.0, 32, 0)
    assert list(cam.pos) == [1, 2, 3]
    assert s.samples_per_pixel() == 5


def test_comments_singletons_and_quotes(fray, tmp_path):
    text = """
// a comment
# another
GlobalSettings {
	frameWidth 123   // trailing comment
	frameHeight 77   # trailing hash comment
	wantAA off
	gi on
	pathsPerPixel 7
}
/* block comment opens at line start
Camera nope { position (9,9,9) }
*/
Camera camera {
	position (1, 2, 3)
	dof false
}
Plane p {
	y 2
	limit "64"
}
Lambert l { }
Lambert l2 {
	color (0.5, 0.25, 0.125)
}
Node n {
	geometry p
	shader l2
	weirdProperty 3
}
Node supernode {
	geometry p
}
"""
    # `Lambert l { }` on one line is NOT valid in the reference grammar (4 tokens): it must fail
    with pytest.raises(fray.FrayError):
        parse(fray, tmp_path, text)
    s = parse(fray, tmp_path, text.replace("Lambert l { }\n", ""))
    assert (s.settings.frameWidth, s.settings.frameHeight, s.settings.wantAA, s.settings.gi, s.settings.numPaths) == (123, 77, 0, 1, 7)
    assert list(s.camera.pos) == [1, 2, 3]
    assert s.desc.n_planes == 1 and s.desc.planes[0].height == 2 and s.desc.planes[0].limit == 64
    assert s.desc.n_nodes == 1            # the shader-less node is dropped from the render list (scene.cpp:563-568)
    assert list(s.desc.shaders[s.desc.nodes[0].shader].color) == [0.5, 0.25, 0.125]
    assert s.samples_per_pixel() == 7


def test_transform_lines_apply_in_file_order(fray, tmp_path):
    text = BASE + """
Plane p {
}
Lambert l {
}
Node a {
	geometry p
	shader l
	scale (2, 2, 2)
	rotate (90, 0, 0)
	translate (1, 0, 0)
}
Node b {
	geometry p
	shader l
	translate (1, 0, 0)
	rotate (90, 0, 0)
	scale (2, 2, 2)
}
"""
    s = parse(fray, tmp_path, text)
    a, b = s.desc.nodes[0].T, s.desc.nodes[1].T
    ma, mb = np.array(a.m).reshape(3, 3), np.array(b.m).reshape(3, 3)
    c, sn = np.cos(np.pi / 2), np.sin(np.pi / 2)
    ry = np.array([[c, 0, sn], [0, 1, 0], [-sn, 0, c]])         # rotationAroundY, matrix.cpp:41-50
    assert np.allclose(ma, 2 * np.eye(3) @ ry) and np.allclose(mb, ry @ (2 * np.eye(3)))
    assert list(a.offset) == [1, 0, 0] and list(b.offset) == [1, 0, 0]
    for T in (a, b):
        assert np.allclose(np.array(T.m).reshape(3, 3) @ np.array(T.invM).reshape(3, 3), np.eye(3))


def test_layered_lines_and_forward_references(fray, tmp_path):
    text = BASE + """
Layered glass {
	layer refr (1, 1, 1)
	layer refl (0.25, 0.5, 0.75) fresnel
}
Refr refr {
	ior 1.5
	multiplier 0.96
}
Refl refl {
	glossiness 0.5
	numSamples 7
}
Fresnel fresnel {
	ior 1.5
}
Sphere s {
	R 2
}
Node n {
	geometry s
	shader glass
}
"""
    s = parse(fray, tmp_path, text)
    sh = s.desc.shaders[0]
    assert sh.kind == 5 and sh.layer_count == 2
    l0, l1 = s.desc.layers[sh.layer_begin], s.desc.layers[sh.layer_begin + 1]
    assert (l0.shader, l0.texture, list(l0.opacity)) == (1, -1, [1, 1, 1])
    assert (l1.shader, l1.texture, list(l1.opacity)) == (2, 0, [0.25, 0.5, 0.75])
    refl = s.desc.shaders[2]
    assert refl.numSamples == 7 and refl.deflectionScaling == 10.0 ** (2 - 4 * 0.5)
    assert abs(s.desc.shaders[1].mult[0] - np.float32(0.96)) == 0


@pytest.mark.parametrize("bad", [
    "Wibble x {\n}\n",                                   # unknown class
    "Camera camera {\n",                                 # unfinished block
    "Camera camera {\n\tposition (1, 2)\n}\n",           # bad vector
    "Camera camera {\n\tposition (1,2,3)\n\tfov 500\n}\n",   # range check (camera.h:61)
    BASE + "Mesh m {\n\tfile \"missing.obj\"\n}\n",      # missing file
    BASE + "Node n {\n\tgeometry nope\n}\n",             # unresolved name
    "Plane p {\n}\n",                                     # no camera
])
def test_malformed_scenes_are_rejected_with_a_message(fray, tmp_path, bad):
    with pytest.raises(fray.FrayError) as e:
        parse(fray, tmp_path, bad)
    assert e.value.code == -2 and len(str(e.value)) > 20


def test_obj_fan_triangulation_and_dummy_indices(fray, tmp_path):
    obj = """# quad + triangle, one face with missing indices
v 0 0 0
v 1 0 0
v 1 1 0
v 0 1 0
vt 0 0
vt 1 0
vt 1 1
f 1/1 2/2 3/3 4
f 1 2 3
"""
    s = parse(fray, tmp_path, BASE + 'Mesh m {\n\tfile "m.obj"\n}\nLambert l {\n}\nNode n {\n\tgeometry m\n\tshader l\n}\n', {"m.obj": obj})
    m = s.desc.meshes[0]
    assert m.n_triangles == 3 and m.n_vertices == 5 and m.n_uvs == 4 and m.n_normals == 0
    assert m.faceted == 1 and m.has_kd == 0          # no normals -> faceted (mesh.cpp:70); <= 20 tris -> no tree
    t = [m.triangles[i] for i in range(3)]
    assert [list(x.v) for x in t] == [[1, 2, 3], [1, 3, 4], [1, 2, 3]]      # fan around the first corner
    assert list(t[1].t) == [1, 3, 0]                                        # missing uv index -> dummy 0
    assert list(m.bbox_min) == [0, 0, 0] and list(m.bbox_max) == [1, 1, 0]
    assert list(t[0].gnormal) == [0, 0, 1] and list(t[0].ABcrossAC) == [0, 0, 1]
    assert list(t[0].dNdx) == [0, 0, 0]              # no normals -> no tangent frame (mesh.cpp:306-309)


def bmp24(w, h, px):
    row = (w * 3 + 3) // 4 * 4
    data = b""
    for y in reversed(range(h)):
        r = b"".join(struct.pack("BBB", *px(x, y)[::-1]) for x in range(w))
        data += r + b"\0" * (row - len(r))
    hdr = b"BM" + struct.pack("<iii", 54 + len(data), 0, 54) + struct.pack("<iiiHHiiiiii", 40, w, h, 1, 24, 0, 0, 0, 0, 0, 0)
    return hdr + data


def test_bmp_loader_and_bump_differentiation(fray, tmp_path):
    px = lambda x, y: (x * 40, y * 60, 255 if (x + y) % 2 else 0)
    files = {"t.bmp": bmp24(5, 3, px)}
    text = BASE + 'BitmapTexture t {\n\tfile "t.bmp"\n\tscaling 4\n}\nBumpTexture b {\n\tfile "t.bmp"\n\tstrength 3\n}\n'
    s = parse(fray, tmp_path, text, files)
    t, b = s.desc.textures[0], s.desc.textures[1]
    assert (t.width, t.height, t.scaling) == (5, 3, 0.25) and (b.bumpIntensity, b.scaling) == (3.0, 1.0)
    tex = np.ctypeslib.as_array(s.desc.texels, shape=(s.desc.n_texels,))
    img = tex[t.texel_offset:t.texel_offset + 45].reshape(3, 5, 3)
    want = np.array([[[c / 255.0 for c in px(x, y)] for x in range(5)] for y in range(3)], np.float32)
    assert np.array_equal(img, want)
    d = tex[b.texel_offset:b.texel_offset + 45].reshape(3, 5, 3)
    inten = (want[..., 0] + want[..., 1] + want[..., 2]) / np.float32(3)
    assert np.array_equal(d[..., 0], inten - np.roll(inten, -1, axis=1))      # bitmap.cpp:300-315
    assert np.array_equal(d[..., 1], inten - np.roll(inten, -1, axis=0))
    assert np.all(d[..., 2] == 0)


def test_shipped_scenes_parse(fray):
    for name, nodes, lights in [("boxed.fray", 9, 2), ("zaphod.fray", 1, 1), ("cornell_box.fray", 7, 1), ("forest.fray", 4, 1),
                                ("smallpt.fray", 7, 1), ("hw9/dragon.fray", 2, 1), ("hw12/sphtri.fray", 1, 3),
                                ("hw10/bokeh.fray", 2, 1), ("hw9/axe_test.fray", 3, 1), ("hw9/nonconvex.fray", 2, 1)]:   # all ten scenes fray ships (nonconvex: four of its five figures sit inside a block comment)
        s = open_scene(fray, name)
        assert (s.desc.n_nodes, s.desc.n_lights) == (nodes, lights), name
        s.close()
    s = open_scene(fray, "cornell_box.fray")
    L = s.desc.lights[0]
    assert L.kind == 1 and (L.xSubd, L.ySubd) == (4, 4) and L.area == float(np.float32(130) * np.float32(105))
    assert list(L.center) == [278, 547.7, 279.5]


def test_exr_piz_cubemap_decodes(fray, oracle):
    """forest.fray's environment: six 256x256 half-RGBA PIZ faces.  The Huffman stage checks itself (exact bit and symbol counts); the image
    statistics catch a wrong wavelet / LUT stage; the hashes were produced by oracle/exr_piz_reader.py, an INDEPENDENT reader written from the OpenEXR
    file-layout / PIZ description (oracle/make_exr_hashes.py) -- the reference decodes through the OpenEXR library, which this image lacks, so two
    implementations sharing only the specification are the strongest pin there is; see also the two tests below."""
    s = open_scene(fray, "forest.fray")
    e = s.desc.environment
    assert e.present == 1 and e.loaded == 1
    assert list(e.width) == [256] * 6 and list(e.height) == [256] * 6
    tex = np.ctypeslib.as_array(s.desc.texels, shape=(s.desc.n_texels,))
    for f in range(6):
        img = tex[e.texel_offset[f]:e.texel_offset[f] + 256 * 256 * 3].reshape(256, 256, 3)
        assert np.isfinite(img).all() and img.min() > 0 and img.max() < 1000
        # natural image: neighbouring rows are strongly correlated, and far smoother than shuffled data
        assert np.corrcoef(img[100, :, 1], img[101, :, 1])[0, 1] > 0.7
        assert np.abs(np.diff(img, axis=0)).mean() < 0.5 * np.abs(img - np.roll(img, 97, axis=0)).mean()
    # the independent reader's hashes of the decoded texels (oracle/make_exr_hashes.py): a single differing bit of a face fails here
    import json
    pins = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "exr_face_hashes.json")))["forest.fray"]
    for f in range(6):
        face = np.ascontiguousarray(tex[e.texel_offset[f]:e.texel_offset[f] + 256 * 256 * 3])
        assert oracle.fnv(face) == pins[f]["fnv"], pins[f]["face"]
    # the sky face is the brightest, the ground face the darkest
    means = [float(tex[e.texel_offset[f]:e.texel_offset[f] + 256 * 256 * 3].mean()) for f in range(6)]
    assert max(range(6), key=lambda f: means[f]) == 4 and min(range(6), key=lambda f: means[f]) == 1   # POSY / NEGY
    s.close()


def test_malformed_assets_fail_cleanly_or_fall_back_to_the_dummy_element(fray, tmp_path):
    # OBJ indices that nothing backs read the dummy element 0 (the reference would index out of bounds)
    obj = "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\nf 1 2 99\nf -4 2 3\nf 1/7/9 2 3\n"
    s = parse(fray, tmp_path, BASE + 'Mesh m {\n\tfile "m.obj"\n}\n', {"m.obj": obj})
    m = s.desc.meshes[0]
    assert [list(m.triangles[i].v) for i in range(4)] == [[1, 2, 3], [1, 2, 0], [0, 2, 3], [1, 2, 3]]
    assert list(m.triangles[3].t) == [0, 0, 0] and list(m.triangles[3].n) == [0, 0, 0]
    # a BMP whose header promises more pixels than the file holds is rejected, not allocated
    good = bmp24(4, 4, lambda x, y: (x * 60, y * 60, 0))
    lying = bytearray(good)
    lying[18:22] = struct.pack("<i", 30000)
    lying[22:26] = struct.pack("<i", 30000)
    for data in (bytes(lying), good[:40], b"XX" + good[2:]):
        with pytest.raises(fray.FrayError):
            parse(fray, tmp_path, BASE + 'BitmapTexture t {\n\tfile "t.bmp"\n}\n', {"t.bmp": data})
    # a corrupt cubemap face leaves the environment declared but not loaded (no crash, misses are black)
    exr = bytearray(open(os.path.join(os.path.dirname(__file__), "..", "scenes", "env", "forest", "posx.exr"), "rb").read())
    exr[5000:5100] = bytes(100)
    files = {"env/%s.exr" % f: bytes(exr) for f in ("negx", "negy", "negz", "posx", "posy", "posz")}
    s = parse(fray, tmp_path, BASE + 'CubemapEnvironment e {\n\tfolder "env"\n}\n', files)
    assert s.desc.environment.present == 1 and s.desc.environment.loaded == 0


def test_loaders_under_address_and_ub_sanitizers(tmp_path):
    """The host loaders (EXR / PIZ, BMP, OBJ + KD build, .fray) built with -fsanitize=address,undefined and fed the shipped
    assets, truncations and bit flips of them, and the hand-made headers that used to overflow: a chunk offset of
    2^64 - 4 (offset + 8 wrapped), a dataWindow whose width overflows int32, a BMP that claims 32768 x 32768 pixels in
    a few hundred bytes, an 8-bit BMP with a negative colour count."""
    import random
    import shutil
    import subprocess
    root = os.path.join(os.path.dirname(__file__), "..")
    exe = tmp_path / "san_loaders"
    src = [os.path.join(root, "tests", "native", "san_loaders.cpp")] + [os.path.join(root, "fray_amd", "csrc", f)
                                                                        for f in ("host_scene.cpp", "host_loaders.cpp", "host_exr.cpp")]
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                    "-I" + os.path.join(root, "include"), "-I" + os.path.join(root, "fray_amd", "csrc")] + src + ["-o", str(exe)], check=True)
    rnd = random.Random(7)
    corpus = []

    def put(name, data):
        f = tmp_path / name
        f.write_bytes(data)
        corpus.append(str(f))

    exr = open(os.path.join(root, "scenes", "env", "forest", "posx.exr"), "rb").read()
    put("good.exr", exr)
    # header geometry of this file: locate the dataWindow attribute and the chunk offset table behind the header's end
    dw = exr.index(b"dataWindow\0box2i\0") + len(b"dataWindow\0box2i\0") + 4
    hdr_end = exr.index(b"\0\0", exr.index(b"lineOrder")) if b"lineOrder" in exr else 0
    bad = bytearray(exr)
    bad[dw:dw + 16] = struct.pack("<iiii", -2147483648, 0, 2147483647, 255)          # x1 - x0 + 1 overflows int32
    put("window_overflow.exr", bytes(bad))
    bad = bytearray(exr)
    bad[dw:dw + 16] = struct.pack("<iiii", 0, -2147483648, 255, 2147483647)
    put("window_overflow_y.exr", bytes(bad))
    # every 8-byte little-endian word that equals a plausible chunk offset becomes 2^64 - 4 in turn (the table sits right after the header)
    first_chunk = min(o for o in (struct.unpack_from("<Q", exr, k)[0] for k in range(300, 600, 1)) if 300 < o < len(exr))
    table = next(k for k in range(300, 600) if struct.unpack_from("<Q", exr, k)[0] == first_chunk)
    for j in range(3):
        bad = bytearray(exr)
        bad[table + 8 * j:table + 8 * j + 8] = struct.pack("<Q", 2 ** 64 - 4 - j)
        put("offset_wrap_%d.exr" % j, bytes(bad))
    for j in range(40):                                                              # truncations and bit flips
        bad = bytearray(exr[:rnd.randrange(8, len(exr))]) if j % 2 else bytearray(exr)
        for _ in range(rnd.randrange(1, 6)):
            bad[rnd.randrange(len(bad))] ^= 1 << rnd.randrange(8)
        put("fuzz_%02d.exr" % j, bytes(bad))
    good = bmp24(4, 4, lambda x, y: (x * 60, y * 60, 0))
    put("good.bmp", good)
    lying = bytearray(good)
    lying[22:26] = struct.pack("<i", 32768)
    lying[18:22] = struct.pack("<i", 1)                                               # one row fits, 32768 rows do not
    put("lying_height.bmp", bytes(lying))
    pal = bytearray(good)
    pal[28:30] = struct.pack("<H", 8)
    pal[46:50] = struct.pack("<i", -5)                                               # negative colour count
    put("negative_colours.bmp", bytes(pal))
    pal[46:50] = struct.pack("<i", 0)
    put("short_palette.bmp", bytes(pal))
    for j in range(20):
        bad = bytearray(good)
        for _ in range(rnd.randrange(1, 5)):
            bad[rnd.randrange(len(bad))] = rnd.randrange(256)
        put("fuzz_%02d.bmp" % j, bytes(bad))
    put("odd.obj", b"v 0 0 0\nv 1 0 0\nv 0 1 0\nvn\nvt 1\nf 1 2 3\nf 1 2 99\nf -4 2 3\nf 1/7/9 2 3\nf\nf 1\n" + b"v " + b"9" * 12000 + b"\n")
    shutil.copy(os.path.join(root, "scenes", "geom", "heart.obj"), tmp_path / "heart.obj")
    corpus.append(str(tmp_path / "heart.obj"))
    r = subprocess.run([str(exe)] + corpus, capture_output=True, text=True, timeout=300,
                       env={**os.environ, "ASAN_OPTIONS": "detect_leaks=0:allocator_may_return_null=1", "UBSAN_OPTIONS": "print_stacktrace=1"})
    assert r.returncode == 0, r.stderr[-3000:]
    lines = dict((os.path.basename(l.split()[1]), l.split()[0]) for l in r.stdout.splitlines() if l)
    assert lines["good.exr"] == "loaded" and lines["good.bmp"] == "loaded" and lines["heart.obj"] == "loaded"
    for name in ("window_overflow.exr", "window_overflow_y.exr", "offset_wrap_0.exr", "offset_wrap_1.exr", "offset_wrap_2.exr",
                 "lying_height.bmp", "negative_colours.bmp"):
        assert lines[name] == "rejected", name


def test_device_trig_and_sort_restatements_on_the_host(tmp_path):
    """Two device headers that restate library arithmetic compile for the host too and are checked against the libraries here:
    dev_trig.hpp (sin / cos / acos) must be correctly rounded -- compared with 113-bit libquadmath -- in all but 2 calls per 10^4 (it
    is ~5 per 10^6), and dev_sort.hpp must order every input, ties included, exactly as this toolchain's std::sort does."""
    import subprocess
    root = os.path.join(os.path.dirname(__file__), "..")
    inc = "-I" + os.path.join(root, "fray_amd", "csrc")
    exe = str(tmp_path / "trig_check")
    subprocess.run(["g++", "-std=c++17", "-O2", "-ffp-contract=off", inc, os.path.join(root, "tests", "native", "trig_check.cpp"), "-o", exe, "-lquadmath"], check=True)
    r = subprocess.run([exe, "1000000"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout
    exe = str(tmp_path / "sort_check")
    subprocess.run(["g++", "-std=c++17", "-O2", inc, os.path.join(root, "tests", "native", "sort_check.cpp"), "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "heapsort fallbacks exercised" in r.stdout and " 0 heapsort" not in r.stdout, r.stdout


def test_certified_box_test_never_contradicts_the_reference_arithmetic(tmp_path):
    """dev_boxcert.hpp (the interval form of BBox::testIntersect the KD walk decides with) compiled for the host: over adversarial rays -- aimed
    at faces, edges and corners with offsets down to 1e-17, starting inside inside()'s 1e-6 shell, nearly axis-parallel, down chains of midpoint
    splits with the child intervals derived incrementally as the device does -- a box classified SURELY TRUE / SURELY FALSE is never decided the
    other way by the reference's own arithmetic (bbox.h:79-134, restated in the harness).  With the margins set to zero the same harness
    reports tens of thousands of contradictions (profiles/r03_experiments/README.md), so it does see them."""
    import subprocess
    root = os.path.join(os.path.dirname(__file__), "..")
    exe = str(tmp_path / "boxcert_check")
    subprocess.run(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-I" + os.path.join(root, "fray_amd", "csrc"),
                    os.path.join(root, "tests", "native", "boxcert_check.cpp"), "-o", exe], check=True)
    r = subprocess.run([exe, "400000"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "mismatches 0" in r.stdout, r.stdout + r.stderr
    n_true = int(r.stdout.split("surely true")[1].split()[0])
    assert n_true > 100000, r.stdout            # the harness must actually classify


def test_certified_box_miss_is_never_a_hit_of_the_reference_routines(tmp_path):
    """dev_misscert.hpp (what lets a ray that passes a CsgOp node's tree by skip CsgOp::intersect, which has no bounding volume in the reference) compiled for
    the host: over random and adversarial rays -- grazing faces, edges and corners at offsets around the certificate's margin, starts within 1e-7 of the
    surface and 1e4 away, directions with exact zeros -- a ray certified to miss a geometry's exact bounding box is never reported hit by the reference's own
    Sphere::intersect, Cube::intersect or BBox::testIntersect (geometry.cpp:52-137, bbox.h:79-134, restated in the harness).  Built without the margins the
    same harness finds contradictions by the ten thousand, so it does see them."""
    import subprocess
    root = os.path.join(os.path.dirname(__file__), "..")
    src, inc = os.path.join(root, "tests", "native", "misscert_check.cpp"), "-I" + os.path.join(root, "fray_amd", "csrc")
    exe = str(tmp_path / "misscert_check")
    subprocess.run(["g++", "-std=c++17", "-O2", "-ffp-contract=off", inc, src, "-o", exe], check=True)
    r = subprocess.run([exe, "3000000"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "contradictions 0" in r.stdout, r.stdout + r.stderr
    assert int(r.stdout.split("certified")[1].split()[0]) > 1000000, r.stdout      # the harness must actually certify
    exe0 = str(tmp_path / "misscert_check0")
    subprocess.run(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-DFRAY_MISSCERT_SCALE=0", "-DNO_HOST_MARGIN", inc, src, "-o", exe0], check=True)
    r0 = subprocess.run([exe0, "1000000"], capture_output=True, text=True, timeout=300)
    assert r0.returncode == 1 and int(r0.stdout.split("contradictions")[1].split()[0]) > 1000, r0.stdout


def test_certified_triangle_filter_never_rejects_what_the_reference_accepts(tmp_path):
    """dev_tricert.hpp (the FP32 "surely misses" filter the KD leaves run before Triangle::intersectFast's arithmetic) compiled for the host: over
    adversarial rays -- aimed at edges and vertices with offsets down to 1e-17 of the triangle, grazing its plane, starting on it, from up to 1e5
    triangle sizes away, at slivers and at triangles of 1e-9 .. 1e6 units far from the mesh's reference point -- a triangle the filter calls
    surely rejected is never accepted by the reference's own arithmetic (triangle.cpp:66-97, restated in the harness, minDist = INF, no culling).
    With the error bound scaled down to 1e-7 the same harness reports contradictions, so it does see them."""
    import subprocess
    root = os.path.join(os.path.dirname(__file__), "..")
    exe = str(tmp_path / "tricert_check")
    subprocess.run(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-I" + os.path.join(root, "fray_amd", "csrc"),
                    os.path.join(root, "tests", "native", "tricert_check.cpp"), "-o", exe], check=True)
    r = subprocess.run([exe, "1500000"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "mismatches 0" in r.stdout, r.stdout + r.stderr
    assert int(r.stdout.split("surely rejected")[1].split()[0]) > 300000, r.stdout       # the harness must actually filter
    assert int(r.stdout.split("reference accepts")[1].split()[0]) > 500000, r.stdout     # ... and aim at triangles
    r = subprocess.run([exe, "1500000", "1e-7"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 1 and "mismatches 0" not in r.stdout, r.stdout


# ---- the product's parser against the reference's parser OBJECT CODE (tests/golden/ref_parse.json, made by oracle/make_parse_golden.py from oracle/_ref:
# ---- scene.o with every fillProperties, the OBJ / BMP loaders, Transform) -- not against this project's reading of the grammar
def _ref_parse_golden():
    import json
    return json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ref_parse.json")))


def _compare_with_reference_dump(fray, path, ref_lines):
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
    from oracle import scene_dump
    if ref_lines is None:
        with pytest.raises(fray.FrayError):        # the reference rejects this text: so must the product
            fray.Scene.parseScene(path)
        return
    s = fray.Scene.parseScene(path)
    diff = scene_dump.first_difference(scene_dump.dump(s.desc), [l.split() for l in ref_lines])
    s.close()
    assert diff is None, "line %d, token %d: product %s, reference %s" % diff


@pytest.mark.parametrize("name", sorted(_ref_parse_golden()["cases"]))
def test_parser_edge_cases_equal_the_reference_parsers_scene(fray, tmp_path, name):
    """Comments, quotes, singleton blocks, the one-line block the grammar rejects, transform lines in file order (nodes and a RectLight), Layered lines with
    forward references, randfloat / randint macros (drawn from the generator initRandom(42) leaves in table entry 0), CsgOp trees with every texture and
    shader kind, OBJ fans with missing / negative / zero indices, every camera and settings field: settings, camera, lights and every
    render-list node's transform (27 doubles), geometry tree and shader tree must equal what the reference's scene.o made of the same text."""
    from parser_cases import write_case
    _compare_with_reference_dump(fray, write_case(name, str(tmp_path)), _ref_parse_golden()["cases"][name])


@pytest.mark.parametrize("name", sorted(_ref_parse_golden()["scenes"]))
def test_repository_scene_files_parse_like_the_reference(fray, name):
    from conftest import SCENES
    _compare_with_reference_dump(fray, os.path.join(SCENES, name), _ref_parse_golden()["scenes"][name])


def test_exr_decoder_equals_the_independent_reader_texel_for_texel(fray):
    """fray_amd/csrc/host_exr.cpp against oracle/exr_piz_reader.py on the six shipped cubemap faces: the same float32 RGB arrays, bit for bit."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
    from conftest import SCENES
    from oracle import exr_piz_reader
    s = open_scene(fray, "forest.fray")
    e = s.desc.environment
    tex = np.ctypeslib.as_array(s.desc.texels, shape=(s.desc.n_texels,))
    for f, name in enumerate(("negx", "negy", "negz", "posx", "posy", "posz")):
        mine = tex[e.texel_offset[f]:e.texel_offset[f] + e.width[f] * e.height[f] * 3].reshape(e.height[f], e.width[f], 3)
        theirs = exr_piz_reader.read_rgb(os.path.join(SCENES, "env", "forest", name + ".exr"))
        assert theirs.shape == mine.shape and np.array_equal(mine, theirs), name
    s.close()


@pytest.mark.parametrize("kind,h,w", [("smooth", 64, 64), ("smooth", 37, 21), ("smooth", 33, 70), ("few", 45, 45), ("few", 5, 3), ("few", 1, 1), ("noise", 40, 17),
                                      ("const", 70, 9), ("smooth", 1, 50), ("smooth", 50, 1), ("few", 96, 31)])
def test_exr_piz_files_of_other_shapes_decode_to_what_was_written(fray, tmp_path, kind, h, w):
    """The shipped faces are 256 x 256 with wide value ranges: they never take the wavelet's 1-D steps for odd rows / columns, its 14-bit form (fewer
    than 2^14 distinct values), the Huffman run-length code, or blocks stored raw.  oracle/exr_piz_writer.py makes PIZ files that do (odd sizes down to
    1 x 1, eight distinct values, constant areas, incompressible noise), and BOTH decoders -- the product's and the independent reader -- must return
    exactly the half-float values that were written."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
    from oracle import exr_piz_reader, exr_piz_writer
    rng = np.random.default_rng(h * 1000 + w)

    def make():
        if kind == "smooth":
            y, x = np.mgrid[0:h, 0:w]
            base = np.sin(x * 0.2) * np.cos(y * 0.13) + 1.5 + rng.normal(size=(h, w)) * 0.05
            return {c: (base * (i + 1)).astype(np.float32) for i, c in enumerate("RGBA")}
        if kind == "few":
            vals = np.array([0, 0.25, 0.5, 1, 2, 4, 8.5, 100], np.float32)
            img = {c: vals[rng.integers(0, len(vals), size=(h, w))] for c in "RGB"}
            img["R"][:h // 2] = 0.5
            return img
        if kind == "noise":
            return {c: rng.random((h, w)).astype(np.float32) * 1000 for c in "RGBA"}
        return {c: np.full((h, w), 3.0, np.float32) for c in "RGB"}

    (tmp_path / "env").mkdir()
    written = {}
    for face in ("negx", "negy", "negz", "posx", "posy", "posz"):
        written[face] = exr_piz_writer.write_exr(str(tmp_path / "env" / (face + ".exr")), make())
        _, _, got = exr_piz_reader.read_exr(str(tmp_path / "env" / (face + ".exr")))
        for c, v in written[face].items():
            assert np.array_equal(got[c], v), (face, c)
    (tmp_path / "scene.fray").write_text('Camera camera {\n\tposition (0,0,0)\n}\nCubemapEnvironment environment {\n\tfolder "env"\n}\n')
    s = fray.Scene.parseScene(str(tmp_path / "scene.fray"))
    e = s.desc.environment
    assert e.present == 1 and e.loaded == 1
    tex = np.ctypeslib.as_array(s.desc.texels, shape=(s.desc.n_texels,))
    for f, face in enumerate(("negx", "negy", "negz", "posx", "posy", "posz")):
        assert (e.width[f], e.height[f]) == (w, h)
        mine = tex[e.texel_offset[f]:e.texel_offset[f] + w * h * 3].reshape(h, w, 3)
        want = np.stack([written[face].get(c, np.zeros((h, w), np.float32)) for c in "RGB"], axis=2)
        assert np.array_equal(mine, want), face
    s.close()
