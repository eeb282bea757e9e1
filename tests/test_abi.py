"""The C-ABI library loads without a GPU and exports exactly what include/frayhip.h declares."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT


def header_functions():
    src = open(os.path.join(ROOT, "include", "frayhip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(frayhip_\w+)\s*\(", src)))


def test_library_exports_every_declared_symbol(fray, abi):
    names = header_functions()
    assert names, "no declarations found"
    assert sorted(abi.SYMBOLS) == names, "fray_amd/abi.py and include/frayhip.h disagree on the entry points"
    for n in names:
        assert hasattr(fray.lib, n), n


def test_struct_layouts_match(fray, abi):
    for name, t in abi.STRUCTS.items():
        assert fray.lib.frayhip_sizeof(name.encode()) == C.sizeof(t), name
    assert fray.lib.frayhip_sizeof(b"nope") == -1
    assert fray.lib.frayhip_abi_version() == abi.ABI_VERSION


def test_every_header_struct_is_mirrored(abi):
    src = open(os.path.join(ROOT, "include", "frayhip.h")).read()
    structs = set(re.findall(r"typedef struct (frayhip_\w+)\s*\{", src))
    assert structs == set(abi.STRUCTS), structs ^ set(abi.STRUCTS)


def test_bucket_count(fray):
    f = fray.lib.frayhip_bucket_count
    assert f(1920, 1080, 0, 1) == 40 * 23
    assert f(48, 48, 0, 1) == 1 and f(49, 48, 0, 1) == 2
    assert sum(f(1920, 1080, r, 8) for r in range(8)) == 920
    assert f(100, 100, 3, 2) < 0 and f(0, 10, 0, 1) < 0


def test_to_rgb32(fray):
    import numpy as np
    rgb = np.array([[0, 0.5, 1.0], [-1, 2, 0.25], [1 / 255.0, 0.4999 / 255, 0.5001 / 255]], np.float32)
    out = np.zeros(3, np.uint32)
    assert fray.lib.frayhip_to_rgb32(rgb.ctypes.data, out.ctypes.data, 3) == 0
    assert out[0] == (0 << 16) | (128 << 8) | 255      # floor(0.5*255 + 0.5) = 128
    assert out[1] == (0 << 16) | (255 << 8) | 64
    assert out[2] == (1 << 16) | (0 << 8) | 1


def test_errors_do_not_raise_across_the_abi(fray, abi):
    hs = C.c_void_p()
    rc = fray.lib.frayhip_scene_parse(b"/nonexistent/scene.fray", C.byref(hs))
    assert rc == abi.E_PARSE and b"Cannot open" in fray.lib.frayhip_last_error()
    assert fray.lib.frayhip_scene_parse(None, C.byref(hs)) == abi.E_ARG


def test_product_does_not_reach_into_the_oracle():
    """Nothing under fray_amd/ (nor bench.py outside its cpu_baseline leg) may import the oracle."""
    for dp, _, files in os.walk(os.path.join(ROOT, "fray_amd")):
        for fn in files:
            if fn.endswith((".py", ".cpp", ".hpp", ".h", ".hip")):
                text = open(os.path.join(dp, fn), errors="replace").read()
                assert "oracle" not in text.lower() or fn in ("dev_rng.hpp",), (fn, "mentions the oracle")


def test_save_bmp_round_trips_through_the_loader(fray, tmp_path):
    """frayhip_save_bmp (Bitmap::saveBMP) -> the scene loader's BMP reader (Bitmap::loadBMP)."""
    import numpy as np
    W, H = 7, 5                                         # row size not a multiple of 4
    rgb = np.linspace(-0.1, 1.1, W * H * 3, dtype=np.float32).reshape(H, W, 3)
    path = tmp_path / "t.bmp"
    assert fray.lib.frayhip_save_bmp(str(path).encode(), rgb.ctypes.data, W, H) == 0
    assert os.path.getsize(path) == 54 + ((W * 3 + 3) // 4 * 4) * H
    (tmp_path / "s.fray").write_text('Camera camera {\n\tposition (0,0,0)\n}\nBitmapTexture t {\n\tfile "t.bmp"\n}\n')
    s = fray.Scene.parseScene(str(tmp_path / "s.fray"))
    t = s.desc.textures[0]
    tex = np.ctypeslib.as_array(s.desc.texels, shape=(s.desc.n_texels,))[t.texel_offset:t.texel_offset + W * H * 3].reshape(H, W, 3)
    want = np.floor(np.clip(rgb, 0, 1) * np.float32(255) + np.float32(0.5)) / np.float32(255)
    assert np.array_equal(tex, want.astype(np.float32))


def test_scene_create_range_checks_a_description_before_touching_the_gpu(fray, abi):
    """A description may come from any host: a bad index must be an error message, never a wild
    device read.  Validation runs before the first HIP call, so this needs no GPU."""
    import ctypes as C
    from conftest import open_scene

    def expect_rejected(mutate, needle, scene="boxed.fray"):
        s = open_scene(fray, scene)
        mutate(s.desc)
        with pytest.raises(fray.FrayError) as e:
            s.beginRender()
        assert e.value.code == abi.E_ARG and needle in str(e.value), str(e.value)
        s.close()

    def poke(ptr_field, index, attr, value):
        def f(d):
            setattr(getattr(d, ptr_field)[index], attr, value)
        return f

    expect_rejected(poke("nodes", 0, "shader", 99), "node reference")
    expect_rejected(poke("nodes", 1, "geom", -3), "node reference")
    expect_rejected(poke("nodes", 0, "bump_tex", 1000), "node reference")
    expect_rejected(poke("geoms", 0, "index", 12345), "geometry reference")
    expect_rejected(poke("geoms", 0, "kind", 7), "geometry reference")
    expect_rejected(poke("shaders", 0, "texture", 77), "shader texture")
    expect_rejected(poke("lights", 0, "xSubd", 0), "RectLight")
    expect_rejected(poke("textures", 0, "kind", 9), "texture kind")

    def bad_tri(d):
        d.meshes[0].triangles[5].v[1] = d.meshes[0].n_vertices
    expect_rejected(bad_tri, "vertex index")

    def bad_kd_child(d):
        d.meshes[0].kdnodes[0].child0 = d.meshes[0].n_kdnodes
    expect_rejected(bad_kd_child, "KD child link")

    def bad_kd_leaf(d):
        m = d.meshes[0]
        for k in range(m.n_kdnodes):
            if m.kdnodes[k].axis == 3 and m.kdnodes[k].tri_count > 0:
                m.kdnodes[k].tri_begin = m.n_trirefs
                return
    expect_rejected(bad_kd_leaf, "KD leaf range")

    def bad_ref(d):
        d.meshes[0].trirefs[3] = -1
    expect_rejected(bad_ref, "KD triangle reference")

    def bad_env(d):
        d.environment.texel_offset[2] = d.n_texels
    expect_rejected(bad_env, "environment face", scene="forest.fray")

    def bad_frame(d):
        d.settings.frameWidth = 0
    expect_rejected(bad_frame, "frame size")


def test_bucket_numbering_is_the_documented_one(fray):
    """frayhip_bucket_xy: bucket b sits in bucket row b / BW and column (b % BW + 3 * row) % BW (include/frayhip.h, FRAYHIP_BUCKET_SKEW): the
    rule the tests restate (conftest.bucket_xy) is the library's for every bucket of a few frame sizes, it is a bijection, and a stride
    that divides the frame's width in buckets (8 ranks, 1920 pixels) deals every rank a bucket in every residue class of columns."""
    import ctypes as C
    from conftest import bucket_xy
    for W, H in [(1920, 1080), (640, 480), (100, 75), (4096, 4096), (47, 49)]:
        BW, BH = (W - 1) // 48 + 1, (H - 1) // 48 + 1
        seen = set()
        for b in range(BW * BH):
            bx, by = C.c_int(), C.c_int()
            assert fray.lib.frayhip_bucket_xy(W, H, b, C.byref(bx), C.byref(by)) == 0
            assert (bx.value, by.value) == bucket_xy(W, b)
            assert 0 <= bx.value < BW and 0 <= by.value < BH
            seen.add((bx.value, by.value))
        assert len(seen) == BW * BH
        bx, by = C.c_int(), C.c_int()
        assert fray.lib.frayhip_bucket_xy(W, H, BW * BH, C.byref(bx), C.byref(by)) != 0
    cols = {r: set() for r in range(8)}
    for b in range(40 * 23):
        x, _ = bucket_xy(1920, b)
        cols[b % 8].add(x % 8)
    assert all(len(c) == 8 for c in cols.values())


def test_build_recipe_keeps_what_correctness_depends_on(fray):
    """Build facts the pictures depend on.  (1) Round 3 compiled the Cube / CSG kernel variants with -amdgpu-spill-sgpr-to-vgpr=false
    because, with SGPR spills in VGPR lanes, two equivalent source changes made k_whitted<2> / k_pt_shadow<2> render wrong pictures; those variants
    were the only ones with out-of-line calls (sixteen CsgOp levels).  Round 4 rebuilt CsgOp::intersect as one loop over an explicit stack: NO device
    function is called anywhere any more, the flag is gone, and both facts are checked here -- a call creeping back in must be a decision.  The cause was
    narrowed down in round 4 (tools/repro/README.md): LLVM's greedy allocator for the VGPRs that hold spilled SGPRs; round 3's tree renders every Cube / CSG
    fuzz scene wrong with it and right with -mllvm -wwm-regalloc=basic, which the whole library is now built with.  (2) The
    host code takes sine and cosine of an angle from ONE sincos() call like the reference's g++ build (clang would call sin() and cos(), whose sine
    differs in the last place for one angle in 700): the library must import sincos and neither sin nor cos."""
    import glob
    import re
    mk = open(os.path.join(ROOT, "Makefile")).read()
    assert "spill-sgpr-to-vgpr" not in mk
    # the allocator of the VGPRs that hold spilled SGPRs: LLVM's default (greedy) one is what miscompiled round 3's kernels (tools/repro/README.md)
    flags = [l for l in mk.splitlines() if l.startswith("HIPFLAGS")]
    assert flags and "-mllvm -wwm-regalloc=basic" in flags[0]
    reports = sorted(glob.glob(os.path.join(ROOT, "fray_amd", "csrc", "variant*.resources.txt")))
    # eight flag words of render_variant.hip + six of render_contract.hip (option "fp_contract": every flag word but the two Cube / CSG ones)
    assert len(reports) == 14 and sum(os.path.basename(r).startswith("variantC") for r in reports) == 6, reports
    # the contracted kernels are built with fused multiply-adds and NOTHING ELSE is: one rule of the Makefile carries the flag
    assert mk.count("-ffp-contract=fast") == 1 and "variantC%.o" in mk.split("-ffp-contract=fast")[0].splitlines()[-2]
    for r in reports:
        names = re.findall(r"Function Name: (\S+)", open(r).read())
        assert names, r
        for n in names:
            assert re.match(r"_ZL\d+k_", n), (os.path.basename(r), n, "is not a kernel: an out-of-line device function")
    import subprocess
    so = os.path.join(ROOT, "fray_amd", "libfrayhip.so")
    syms = subprocess.run(["nm", "-D", "--undefined-only", so], capture_output=True, text=True, check=True).stdout
    names = {l.split()[-1].split("@")[0] for l in syms.splitlines() if l.strip()}
    assert "sincos" in names and "sin" not in names and "cos" not in names, sorted(n for n in names if n in ("sin", "cos", "sincos"))


def test_loopback_stand_in_exports_what_the_library_binds():
    """tests/native/librccl_loopback.so (test infrastructure for tests/test_gpu_gather_loopback.py) must offer every RCCL entry point
    fray_amd/csrc/capi_comm.hip binds, and FRAYHIP_RCCL_LIBRARY must make the library bind it (no GPU needed for either)."""
    import subprocess
    import sys
    so = os.path.join(ROOT, "tests", "native", "librccl_loopback.so")
    assert os.path.exists(so), "not built (make)"
    bound = re.findall(r'sym\("(nccl\w+)"\)', open(os.path.join(ROOT, "fray_amd", "csrc", "capi_comm.hip")).read())
    assert len(bound) == 10
    lib = C.CDLL(so)
    for n in bound:
        assert hasattr(lib, n), n
    code = "import fray_amd; print(fray_amd.lib.frayhip_comm_available(), fray_amd.lib.frayhip_comm_library().decode())"
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env={**os.environ, "FRAYHIP_RCCL_LIBRARY": so}, capture_output=True, text=True, timeout=300)
    assert r.stdout.split() == ["1", so], r.stdout + r.stderr
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env={**os.environ, "FRAYHIP_RCCL_LIBRARY": "/nonexistent/librccl.so"}, capture_output=True, text=True, timeout=300)
    assert r.stdout.split() == ["0"], r.stdout + r.stderr            # a named build that cannot be loaded is an error, not a reason to bind another
