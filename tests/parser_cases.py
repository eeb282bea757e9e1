"""Scene texts that exercise the edge cases of the reference's grammar (scene.cpp:403-570, 609-653; mesh.cpp:203-258): shared by
tests/test_host_scene.py (the product's parser) and oracle/make_parse_golden.py (the reference's parser object code, which writes
tests/golden/ref_parse.json)."""
import os
import struct

BASE = """
Camera camera {
	position (1, 2, 3)
}
"""

QUAD_OBJ = """# quad + triangle, one face with missing indices
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

NEG_OBJ = """v 0 0 0
v 2 0 0
v 2 2 0
v 0 2 1
vn 0 0 1
vn 0 1 0
f 1//1 2//1 3//2
f -1 0 2
f 1//1 2//1 4//2
"""
# (An index BEYOND the end of its array -- "f 1/9/1" with no vt lines -- is kept raw by the reference, mesh.cpp:233-247, and read out of bounds when a
# texture asks for uvs; the product maps it to the dummy element 0.  Not a comparable case: left out.)


def bmp24(w, h, px):
    row = (w * 3 + 3) // 4 * 4
    data = b""
    for y in reversed(range(h)):
        r = b"".join(struct.pack("BBB", *px(x, y)[::-1]) for x in range(w))
        data += r + b"\0" * (row - len(r))
    hdr = b"BM" + struct.pack("<iii", 54 + len(data), 0, 54) + struct.pack("<iiiHHiiiiii", 40, w, h, 1, 24, 0, 0, 0, 0, 0, 0)
    return hdr + data


CASES = {
    "defaults": (BASE, {}),
    "comments_singletons_quotes": ("""
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
""", {}),
    "one_line_block_is_rejected": (BASE + "Lambert l { }\n", {}),
    "transform_order": (BASE + """
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
Node c {
	geometry p
	shader l
	rotate (17.5, -33, 71)
	scale (1, 2.5, 0.25)
	rotate (5, 5, 5)
	translate (-3, 0.125, 9)
	translate (1, 1, 1)
}
RectLight r {
	scale (3, 1, 2)
	rotate (10, 20, 30)
	translate (0, 9, 0)
	xSubd 3
	ySubd 2
	power 11
	color (0.25, 0.5, 1)
}
PointLight q {
	pos (4, 5, 6)
	power 7.5
}
""", {}),
    "layered_and_forward_references": (BASE + """
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
""", {}),
    "macros": (BASE + """
GlobalSettings {
	frameWidth randint(100,200)
	frameHeight randint(50, 60)
	ambientLight (randfloat(0,1), randfloat(0.25,0.5), 0.125)
}
Sphere s {
	O (randfloat(-5,5), randfloat(-5,5), randfloat(1, 2))
	R randfloat(0.5,1.5)
}
Cube c {
	halfSide randfloat(1,1)
	O (randint(-3,3), randint(7,7), randint(0,100))
}
Phong ph {
	color (randfloat(0,1), randfloat(0,1), randfloat(0,1))
	specularExponent randint(5,50)
}
Node n {
	geometry s
	shader ph
	translate (randfloat(-1,1), randint(2,4), 0)
	rotate (randfloat(0,360), 0, randfloat(-10,10))
}
Node m {
	geometry c
	shader ph
	scale (randfloat(1,2), randfloat(1,2), randfloat(1,2))
}
""", {}),
    "csg_textures_shaders": (BASE + """
Cube a {
	halfSide 2
}
Sphere b {
	R 2.5
	O (0.5, 0, 0)
}
Plane fl {
	y -1
	limit 30
}
CsgMinus d {
	left a
	right b
}
CsgAnd e {
	left d
	right a
}
CsgPlus f {
	left e
	right b
}
CheckerTexture chk {
	color1 (0.9, 0.8, 0.7)
	color2 (0.1, 0.2, 0.3)
	scaling 2.5
}
BitmapTexture pic {
	file "t.bmp"
	scaling 4
}
BumpTexture dents {
	file "t.bmp"
	strength 3
	scaling 0.5
}
Lambert l1 {
	texture chk
}
Lambert l2 {
	texture pic
	color (0.5, 0.5, 0.5)
}
Phong p1 {
	color (0.8, 0.3, 0.2)
	specularExponent 33
	specularMultiplier 0.75
	specularColor (0.1, 0.9, 0.4)
	texture chk
}
Const k {
	color (0.2, 0.9, 0.4)
}
Refl mirror {
	multiplier 0.85
}
Node n1 {
	geometry f
	shader l1
}
Node n2 {
	geometry fl
	shader l2
	bump dents
}
Node n3 {
	geometry d
	shader p1
	scale (1, 2, 1)
}
Node n4 {
	geometry b
	shader k
}
Node n5 {
	geometry a
	shader mirror
	translate (5, 0, 0)
}
""", {"t.bmp": bmp24(5, 3, lambda x, y: (x * 40, y * 60, 255 if (x + y) % 2 else 0))}),
    "obj_fan_and_dummy_indices": (BASE + 'Mesh m {\n\tfile "m.obj"\n}\nLambert l {\n}\nNode n {\n\tgeometry m\n\tshader l\n}\n', {"m.obj": QUAD_OBJ}),
    "obj_negative_and_zero_indices": (BASE + 'Mesh m {\n\tfile "m.obj"\n\tbackfaceCulling false\n\tfaceted true\n}\nLambert l {\n}\nNode n {\n\tgeometry m\n\tshader l\n\trotate (45, 0, 0)\n}\n',
                                                   {"m.obj": NEG_OBJ}),
    "camera_and_settings_fields": ("""
GlobalSettings {
	frameWidth 321
	frameHeight 123
	ambientLight (0.1, 0.2, 0.3)
	maxTraceDepth 9
	saturation 0.25
	wantPrepass off
	wantAA on
	numThreads 3
}
Camera camera {
	position (1.5, -2.25, 3.125)
	yaw 12.5
	pitch -7.75
	roll 3
	fov 77
	aspectRatio 2.5
	dof on
	numSamples 9
	fNumber 5.6
	focalPlaneDist 12.5
	autofocus off
	stereoSeparation 0.35
	leftMask (1, 0.5, 0)
	rightMask (0, 0.5, 1)
}
""", {}),
}


def write_case(name, folder):
    """Writes the case's files into `folder`; returns the path of its scene file."""
    text, files = CASES[name]
    for fn, data in files.items():
        p = os.path.join(folder, fn)
        os.makedirs(os.path.dirname(p), exist_ok=True)
        with open(p, "wb") as f:
            f.write(data if isinstance(data, bytes) else data.encode())
    path = os.path.join(folder, "scene.fray")
    with open(path, "w") as f:
        f.write(text)
    return path
