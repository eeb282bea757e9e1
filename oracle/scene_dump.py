"""Test infrastructure: the text oracle/ref_dump.cpp writes for the reference's `Scene scene`, written from the PRODUCT's frayhip_scene_desc
(what fray_amd's own parser, loaders and flattening made of the same file).  Numbers are printed as C's "%a" prints them on the other side;
tests compare token by token (tokens that parse as hex floats by value and sign, everything else as text)."""
import struct


def _tok(v):
    return float(v).hex()


def _vec(a, n=3):
    return [_tok(a[i]) for i in range(n)]


def _xform(T):
    return ["T"] + _vec(T.offset) + _vec(T.m, 9) + _vec(T.invM, 9)


def _texture(d, i):
    if i < 0:
        return ["tex-none"]
    t = d.textures[i]
    if t.kind == 0:
        return ["tex-checker"] + _vec(t.color1) + _vec(t.color2) + [_tok(t.scaling)]
    if t.kind == 1:
        return ["tex-bitmap", str(t.width), str(t.height), _tok(t.scaling)]
    if t.kind == 2:
        return ["tex-bump", str(t.width), str(t.height), _tok(t.scaling), _tok(t.bumpIntensity)]
    if t.kind == 3:
        return ["tex-fresnel", _tok(t.ior)]
    return ["tex-unknown"]


def _shader(d, i, depth=0):
    if i < 0:
        return ["shader-none"]
    if depth > 40:
        return ["shader-too-deep"]
    s = d.shaders[i]
    if s.kind == 0:
        return ["const"] + _vec(s.color)
    if s.kind == 1:
        return ["lambert"] + _vec(s.color) + _texture(d, s.texture)
    if s.kind == 2:
        return ["phong"] + _vec(s.color) + _vec(s.specularColor) + [_tok(s.exponent), _tok(s.specularMultiplier)] + _texture(d, s.texture)
    if s.kind == 3:
        return ["refl"] + _vec(s.mult) + [_tok(s.glossiness), _tok(s.deflectionScaling), str(s.numSamples)]
    if s.kind == 4:
        return ["refr"] + _vec(s.mult) + [_tok(s.ior)]
    if s.kind == 5:
        out = ["layered", str(s.layer_count), "["]
        for k in range(s.layer_count):
            L = d.layers[s.layer_begin + k]
            out += ["layer"] + _vec(L.opacity) + _texture(d, L.texture) + _shader(d, L.shader, depth + 1)
        return out + ["]"]
    return ["shader-unknown"]


def _geometry(d, gi, depth=0):
    if gi < 0:
        return ["geom-none"]
    if depth > 40:
        return ["geom-too-deep"]
    g = d.geoms[gi]
    if g.kind == 0:
        p = d.planes[g.index]
        return ["plane", _tok(p.limit), _tok(p.height)]
    if g.kind == 1:
        s = d.spheres[g.index]
        return ["sphere"] + _vec(s.O) + [_tok(s.R)]
    if g.kind == 2:
        c = d.cubes[g.index]
        return ["cube"] + _vec(c.O) + [_tok(c.halfSide)]
    if g.kind == 3:
        m = d.meshes[g.index]
        out = ["mesh", str(m.n_vertices), str(m.n_normals), str(m.n_uvs), str(m.n_triangles), "faceted", str(m.faceted), "culling", str(m.backfaceCulling), "kd", str(m.has_kd)]
        out += _vec(m.bbox_min) + _vec(m.bbox_max)
        if m.n_vertices:
            out += [_tok(m.vertices[k]) for k in range(3)] + [_tok(m.vertices[3 * (m.n_vertices - 1) + k]) for k in range(3)]
        if m.n_triangles:
            for t in (m.triangles[0], m.triangles[m.n_triangles - 1]):
                out += ["tri"] + [str(x) for x in list(t.v) + list(t.n) + list(t.t)] + _vec(t.gnormal) + _vec(t.AB) + _vec(t.AC) + _vec(t.ABcrossAC)
        return out
    if g.kind == 4:
        c = d.csgs[g.index]
        return [("csg-plus", "csg-and", "csg-minus")[c.op], "("] + _geometry(d, c.left, depth + 1) + [","] + _geometry(d, c.right, depth + 1) + [")"]
    return ["geom-unknown"]


def dump(desc):
    """The scene description as a list of lines, each a list of tokens."""
    d = desc
    s, c = d.settings, d.camera
    lines = [["settings", str(s.frameWidth), str(s.frameHeight), "aa", str(s.wantAA), "gi", str(s.gi), "paths", str(s.numPaths), "depth", str(s.maxTraceDepth),
              "prepass", str(s.wantPrepass), _tok(s.saturation)] + _vec(s.ambientLight)]
    lines.append(["camera"] + _vec(c.pos) + [_tok(v) for v in (c.yaw, c.pitch, c.roll, c.fov, c.aspectRatio, c.focalPlaneDist, c.fNumber, c.stereoSeparation)] +
                 ["dof", str(c.dof), "autofocus", str(c.autofocus), "samples", str(c.numDOFSamples)] + _vec(c.leftMask) + _vec(c.rightMask))
    lines.append(["environment", str(d.environment.present)])
    lines.append(["lights", str(d.n_lights)])
    for i in range(d.n_lights):
        L = d.lights[i]
        if L.kind == 0:
            lines.append(["light", "point"] + _vec(L.color) + [_tok(L.power)] + _vec(L.pos))
        else:
            lines.append(["light", "rect"] + _vec(L.color) + [_tok(L.power), str(L.xSubd), str(L.ySubd)] + _xform(L.T) + _vec(L.center) + [_tok(L.area)])
    lines.append(["nodes", str(d.n_nodes)])
    for i in range(d.n_nodes):
        n = d.nodes[i]
        lines.append(["node"] + _xform(n.T) + ["|"] + _geometry(d, n.geom) + ["|"] + _shader(d, n.shader) + ["|", "bump"] + _texture(d, n.bump_tex))
    return lines


def parse_text(text):
    return [l.split() for l in text.splitlines() if l.strip()]


def _same(a, b):
    fa = fb = None
    try:
        if "0x" in a or "0x" in b or a in ("inf", "-inf", "nan") or b in ("inf", "-inf", "nan"):
            fa, fb = float.fromhex(a), float.fromhex(b)
    except ValueError:
        return a == b
    if fa is None:
        return a == b
    return struct.pack("<d", fa) == struct.pack("<d", fb)


def first_difference(mine, theirs):
    """None when the two dumps (lists of token lists) agree; else (line, token index, mine, theirs)."""
    for li in range(max(len(mine), len(theirs))):
        if li >= len(mine) or li >= len(theirs):
            return (li, -1, mine[li] if li < len(mine) else None, theirs[li] if li < len(theirs) else None)
        a, b = mine[li], theirs[li]
        for ti in range(max(len(a), len(b))):
            if ti >= len(a) or ti >= len(b) or not _same(a[ti], b[ti]):
                return (li, ti, a[max(0, ti - 2):ti + 3], b[max(0, ti - 2):ti + 3])
    return None
