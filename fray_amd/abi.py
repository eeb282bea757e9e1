"""ctypes mirror of include/frayhip.h (the C ABI of libfrayhip.so).

Field order and types follow the header one for one; tests/test_abi.py checks the struct sizes
against the library's own sizeof table so that the two cannot drift apart silently.
"""
import ctypes as C

i32, u32, i64, u64, f32, f64 = C.c_int32, C.c_uint32, C.c_int64, C.c_uint64, C.c_float, C.c_double
P = C.POINTER

ABI_VERSION = 3

OK, E_ARG, E_PARSE, E_NODEVICE, E_UNSUPPORTED, E_NOMEM, E_HIP = 0, -1, -2, -3, -4, -5, -6
GEOM_PLANE, GEOM_SPHERE, GEOM_CUBE, GEOM_MESH, GEOM_CSG = range(5)
TEX_CHECKER, TEX_BITMAP, TEX_BUMP, TEX_FRESNEL = range(4)
SHADER_CONST, SHADER_LAMBERT, SHADER_PHONG, SHADER_REFL, SHADER_REFR, SHADER_LAYERED = range(6)
LIGHT_POINT, LIGHT_RECT = range(2)
MODE_PRIMARY_ID, MODE_RENDER = 0, 1
FRAME_STATS = 1


class Transform(C.Structure):
    _fields_ = [("offset", f64 * 3), ("m", f64 * 9), ("invM", f64 * 9)]


class GeomRef(C.Structure):
    _fields_ = [("kind", i32), ("index", i32)]


class Node(C.Structure):
    _fields_ = [("geom", i32), ("shader", i32), ("bump_tex", i32), ("_pad", i32), ("T", Transform)]


class Plane(C.Structure):
    _fields_ = [("limit", f64), ("height", f64)]


class Sphere(C.Structure):
    _fields_ = [("O", f64 * 3), ("R", f64)]


class Cube(C.Structure):
    _fields_ = [("O", f64 * 3), ("halfSide", f64)]


class Csg(C.Structure):
    _fields_ = [("op", i32), ("left", i32), ("right", i32), ("_pad", i32)]


class Triangle(C.Structure):
    _fields_ = [("v", i32 * 3), ("n", i32 * 3), ("t", i32 * 3), ("_pad", i32),
                ("gnormal", f64 * 3), ("dNdx", f64 * 3), ("dNdy", f64 * 3),
                ("AB", f64 * 3), ("AC", f64 * 3), ("ABcrossAC", f64 * 3)]


class KDNode(C.Structure):
    _fields_ = [("axis", i32), ("child0", i32), ("parent", i32), ("tri_begin", i32),
                ("tri_count", i32), ("_pad", i32), ("split", f64)]


class Mesh(C.Structure):
    _fields_ = [("n_vertices", i32), ("n_normals", i32), ("n_uvs", i32), ("n_triangles", i32),
                ("n_kdnodes", i32), ("n_trirefs", i32),
                ("faceted", i32), ("backfaceCulling", i32), ("has_kd", i32), ("_pad", i32),
                ("bbox_min", f64 * 3), ("bbox_max", f64 * 3),
                ("vertices", P(f64)), ("normals", P(f64)), ("uvs", P(f64)),
                ("triangles", P(Triangle)), ("kdnodes", P(KDNode)), ("trirefs", P(i32)),
                ("kd_max_depth", i32), ("kd_depth_sum", i32)]


class Texture(C.Structure):
    _fields_ = [("kind", i32), ("width", i32), ("height", i32), ("_pad", i32),
                ("color1", f32 * 3), ("color2", f32 * 3),
                ("scaling", f64), ("bumpIntensity", f64), ("ior", f64), ("texel_offset", i64)]


class Shader(C.Structure):
    _fields_ = [("kind", i32), ("texture", i32), ("color", f32 * 3), ("specularColor", f32 * 3),
                ("mult", f32 * 3), ("numSamples", i32),
                ("exponent", f64), ("specularMultiplier", f64), ("glossiness", f64),
                ("deflectionScaling", f64), ("ior", f64), ("layer_begin", i32), ("layer_count", i32)]


class Layer(C.Structure):
    _fields_ = [("shader", i32), ("texture", i32), ("opacity", f32 * 3), ("_pad", i32)]


class Light(C.Structure):
    _fields_ = [("kind", i32), ("xSubd", i32), ("ySubd", i32), ("_pad", i32),
                ("color", f32 * 3), ("power", f32), ("pos", f64 * 3), ("T", Transform),
                ("center", f64 * 3), ("area", f64)]


class Camera(C.Structure):
    _fields_ = [("pos", f64 * 3), ("yaw", f64), ("pitch", f64), ("roll", f64), ("fov", f64),
                ("aspectRatio", f64), ("focalPlaneDist", f64), ("fNumber", f64),
                ("stereoSeparation", f64),
                ("dof", i32), ("autofocus", i32), ("numDOFSamples", i32), ("_pad", i32),
                ("leftMask", f32 * 3), ("rightMask", f32 * 3)]


class Settings(C.Structure):
    _fields_ = [("frameWidth", i32), ("frameHeight", i32), ("ambientLight", f32 * 3),
                ("wantAA", i32), ("gi", i32), ("maxTraceDepth", i32), ("dbg", i32),
                ("saturation", f32), ("wantPrepass", i32), ("numPaths", i32), ("numThreads", i32),
                ("interactive", i32), ("fullscreen", i32)]


class Environment(C.Structure):
    _fields_ = [("present", i32), ("loaded", i32), ("width", i32 * 6), ("height", i32 * 6),
                ("texel_offset", i64 * 6)]


class SceneDesc(C.Structure):
    _fields_ = [("abi_version", i32),
                ("n_nodes", i32), ("n_geoms", i32), ("n_planes", i32), ("n_spheres", i32),
                ("n_cubes", i32), ("n_csgs", i32), ("n_meshes", i32),
                ("n_shaders", i32), ("n_layers", i32), ("n_textures", i32), ("n_lights", i32),
                ("n_texels", i64),
                ("nodes", P(Node)), ("geoms", P(GeomRef)), ("planes", P(Plane)),
                ("spheres", P(Sphere)), ("cubes", P(Cube)), ("csgs", P(Csg)), ("meshes", P(Mesh)),
                ("shaders", P(Shader)), ("layers", P(Layer)), ("textures", P(Texture)),
                ("lights", P(Light)), ("texels", P(f32)),
                ("environment", Environment), ("camera", Camera), ("settings", Settings)]


class Frame(C.Structure):
    _fields_ = [("mode", i32), ("seed", u32), ("bucket_first", i32), ("bucket_stride", i32),
                ("spp_chunk", i32), ("flags", i32)]


class Stats(C.Structure):
    _fields_ = [("closest_rays", u64), ("shadow_rays", u64), ("node_tests", u64),
                ("kd_inner_visits", u64), ("leaf_refs", u64), ("tri_tests", u64),
                ("prim_tests", u64), ("smooth_hits", u64), ("samples", u64),
                ("texture_fetches", u64),
                ("ms_total", f64), ("ms_kernels", f64), ("ms_trace", f64),
                ("trace_launches", u64), ("alg_bytes_trace", f64),
                ("ms_shadow", f64), ("shadow_launches", u64), ("alg_bytes_shadow", f64),
                ("alg_flops_trace", f64), ("alg_flops_shadow", f64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


STRUCTS = {"frayhip_transform": Transform, "frayhip_geom_ref": GeomRef, "frayhip_node": Node,
           "frayhip_plane": Plane, "frayhip_sphere": Sphere, "frayhip_cube": Cube,
           "frayhip_csg": Csg, "frayhip_triangle": Triangle, "frayhip_kdnode": KDNode,
           "frayhip_mesh": Mesh, "frayhip_texture": Texture, "frayhip_shader": Shader,
           "frayhip_layer": Layer, "frayhip_light": Light, "frayhip_camera": Camera,
           "frayhip_settings": Settings, "frayhip_environment": Environment,
           "frayhip_scene_desc": SceneDesc, "frayhip_frame": Frame, "frayhip_stats": Stats}

# Every symbol include/frayhip.h declares: name -> (restype, argtypes)
VP = C.c_void_p
SYMBOLS = {
    "frayhip_scene_parse": (C.c_int, [C.c_char_p, P(VP)]),
    "frayhip_host_scene_desc": (P(SceneDesc), [VP]),
    "frayhip_host_scene_free": (None, [VP]),
    "frayhip_init": (C.c_int, [C.c_int]),
    "frayhip_scene_create": (C.c_int, [P(SceneDesc), P(VP)]),
    "frayhip_scene_destroy": (None, [VP]),
    "frayhip_scene_set_view": (C.c_int, [VP, P(Camera), P(Settings)]),
    "frayhip_scene_set_option": (C.c_int, [VP, C.c_char_p, i64]),
    "frayhip_scene_get_option": (C.c_int, [VP, C.c_char_p, P(i64)]),
    "frayhip_render": (C.c_int, [VP, P(Frame), VP, VP, VP, P(Stats)]),
    "frayhip_render_device": (C.c_int, [VP, P(Frame), VP, VP, VP, VP, P(Stats)]),
    "frayhip_bucket_count": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "frayhip_pack_buckets_device": (C.c_int, [VP, VP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, VP]),
    "frayhip_unpack_buckets_device": (C.c_int, [VP, VP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, VP]),
    "frayhip_bucket_xy": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "frayhip_comm_available": (C.c_int, []),
    "frayhip_comm_library": (C.c_char_p, []),
    "frayhip_comm_unique_id": (C.c_int, [VP]),
    "frayhip_comm_create": (C.c_int, [VP, C.c_int, C.c_int, P(VP)]),
    "frayhip_comm_from_nccl": (C.c_int, [VP, C.c_int, C.c_int, P(VP)]),
    "frayhip_comm_ranks": (C.c_int, [VP]),
    "frayhip_comm_destroy": (None, [VP]),
    "frayhip_gather_buckets": (C.c_int, [VP, VP, C.c_int, C.c_int, C.c_int, C.c_int, VP]),
    "frayhip_to_rgb32": (C.c_int, [VP, VP, C.c_int]),
    "frayhip_save_bmp": (C.c_int, [C.c_char_p, VP, C.c_int, C.c_int]),
    "frayhip_debug_rng": (C.c_int, [u32, C.c_int, VP, VP, VP, C.c_int]),
    "frayhip_debug_libm": (C.c_int, [C.c_int, VP, VP, VP, VP, VP]),
    "frayhip_last_error": (C.c_char_p, []),
    "frayhip_abi_version": (C.c_int, []),
    "frayhip_sizeof": (C.c_int, [C.c_char_p]),
}


def bind(lib):
    """Attach restype/argtypes for every declared symbol; raises AttributeError if one is missing."""
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib
