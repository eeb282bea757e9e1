/*
 * frayhip.h -- C ABI of the MI355X-native renderer for fray's per-pixel ray-trace hot path.
 *
 * The reference (anrieff/fray) has no FFI; its seams are C++ globals and virtuals (SURVEY.md
 * section 8b).  Every entry point below names the reference interface it stands behind.
 * All structs are POD, little-endian, FP64 geometry / FP32 colour, int32 indices; nothing here
 * mentions torch, HIP or C++ types.  No exception crosses this boundary: every call returns
 * 0 on success or a negative FRAYHIP_E_* code, with text available from frayhip_last_error().
 *
 * Citations are relative to the reference tree (src/...).
 */
#ifndef FRAYHIP_H
#define FRAYHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 3: the bucket numbering changed meaning (bucket b sits in column (b % BW + FRAYHIP_BUCKET_SKEW * row) % BW, see frayhip_bucket_xy): a host built
 * against version 2 that packs or unpacks buckets itself (row-major, bx = b % BW) would scatter tiles to the wrong places, so it must not pass the
 * version check.  frayhip_bucket_xy and frayhip_comm_available were added with the same change. */
#define FRAYHIP_ABI_VERSION 3

/* ---- error codes ------------------------------------------------------------------------ */
enum {
    FRAYHIP_OK            = 0,
    FRAYHIP_E_ARG         = -1,  /* bad argument (null pointer, bad size, bad mode)          */
    FRAYHIP_E_PARSE       = -2,  /* scene file syntax error / missing file (scene.cpp:544-554) */
    FRAYHIP_E_NODEVICE    = -3,  /* no usable HIP device (none present, bad id, no driver)    */
    FRAYHIP_E_UNSUPPORTED = -4,  /* scene uses an element the device path does not implement  */
    FRAYHIP_E_NOMEM       = -5,  /* host or device allocation failed                           */
    FRAYHIP_E_HIP         = -6,  /* any other HIP runtime failure (launch, copy, event); text in frayhip_last_error() */
};

/* ---- scene description (flattened `Scene`, scene.h:280-299) ------------------------------ */

/* Transform, matrix.h:72-98: row-vector convention v*M; m and invM are row-major 3x3. */
typedef struct frayhip_transform {
    double offset[3];
    double m[9];
    double invM[9];
} frayhip_transform;

enum { FRAYHIP_GEOM_PLANE = 0, FRAYHIP_GEOM_SPHERE = 1, FRAYHIP_GEOM_CUBE = 2,
       FRAYHIP_GEOM_MESH = 3, FRAYHIP_GEOM_CSG = 4 };

/* Every geometry of the scene in declaration order (scene.geometries); `index` selects the
 * record in the per-kind array. */
typedef struct frayhip_geom_ref { int32_t kind, index; } frayhip_geom_ref;

/* Node, geometry.h:158-176.  Only nodes WITH a shader are listed (scene.cpp:563-568), in file
 * order; the position in this array is the hit id. */
typedef struct frayhip_node {
    int32_t geom;        /* index into geoms[]                       */
    int32_t shader;      /* index into shaders[]                     */
    int32_t bump_tex;    /* index into textures[] or -1              */
    int32_t _pad;
    frayhip_transform T;
} frayhip_node;

typedef struct frayhip_plane  { double limit, height; } frayhip_plane;          /* geometry.h:56-70  */
typedef struct frayhip_sphere { double O[3]; double R; } frayhip_sphere;       /* geometry.h:72-90  */
typedef struct frayhip_cube   { double O[3]; double halfSide; } frayhip_cube;  /* geometry.h:92-115 */
enum { FRAYHIP_CSG_PLUS = 0, FRAYHIP_CSG_AND = 1, FRAYHIP_CSG_MINUS = 2 };
typedef struct frayhip_csg    { int32_t op, left, right, _pad; } frayhip_csg;  /* left/right: geoms[] */

/* Triangle, triangle.h:30-41 (same members, same meaning). */
typedef struct frayhip_triangle {
    int32_t v[3], n[3], t[3];
    int32_t _pad;
    double gnormal[3], dNdx[3], dNdy[3], AB[3], AC[3], ABcrossAC[3];
} frayhip_triangle;

/* KDTreeNode, mesh.h:35-53, linearised in the order buildKD visits nodes (pre-order).
 * axis 0..2 = inner node, 3 = leaf (Axis::AXIS_NONE).  Children are allocated pairwise in the
 * reference (mesh.cpp:57) and stay adjacent here: right child = child0 + 1. */
typedef struct frayhip_kdnode {
    int32_t axis;
    int32_t child0;      /* inner: index of left child; leaf: -1            */
    int32_t parent;      /* -1 for the root                                 */
    int32_t tri_begin;   /* leaf: first entry in trirefs[]                  */
    int32_t tri_count;   /* leaf: number of entries                         */
    int32_t _pad;
    double  split;       /* inner: splitPos                                 */
} frayhip_kdnode;

/* Mesh, mesh.h:55-100.  vertices/normals/uvs hold xyz triples; element 0 of each is the
 * dummy the OBJ loader inserts (mesh.cpp:209-211); normals is empty (n_normals == 0) when the
 * file has none (mesh.cpp:252). */
typedef struct frayhip_mesh {
    int32_t n_vertices, n_normals, n_uvs, n_triangles, n_kdnodes, n_trirefs;
    int32_t faceted, backfaceCulling, has_kd, _pad;
    double  bbox_min[3], bbox_max[3];
    const double*           vertices;
    const double*           normals;
    const double*           uvs;
    const frayhip_triangle* triangles;
    const frayhip_kdnode*   kdnodes;
    const int32_t*          trirefs;
    int32_t kd_max_depth, kd_depth_sum;   /* statistics printed by mesh.cpp:91 */
} frayhip_mesh;

enum { FRAYHIP_TEX_CHECKER = 0, FRAYHIP_TEX_BITMAP = 1, FRAYHIP_TEX_BUMP = 2, FRAYHIP_TEX_FRESNEL = 3 };
/* Texture family, shading.h:33-110, 211-222.  Bitmap texels are float RGB triples in
 * texels[texel_offset ...], row-major, as Bitmap::data (bitmap.h:31-35).  For BUMP the texels are
 * already differentiated (bitmap.cpp:300-315).  `scaling` is stored as the reference uses it
 * at sample time (BitmapTexture inverts it at parse, shading.h:66-67). */
typedef struct frayhip_texture {
    int32_t kind, width, height, _pad;
    float   color1[3], color2[3];
    double  scaling, bumpIntensity, ior;
    int64_t texel_offset;        /* in floats */
} frayhip_texture;

enum { FRAYHIP_SHADER_CONST = 0, FRAYHIP_SHADER_LAMBERT = 1, FRAYHIP_SHADER_PHONG = 2,
       FRAYHIP_SHADER_REFL = 3, FRAYHIP_SHADER_REFR = 4, FRAYHIP_SHADER_LAYERED = 5 };
/* Shader family, shading.h:112-255. */
typedef struct frayhip_shader {
    int32_t kind;
    int32_t texture;             /* diffuseTex -> textures[] or -1                        */
    float   color[3];
    float   specularColor[3];
    float   mult[3];             /* Refl / Refr multiplier colour                         */
    int32_t numSamples;          /* Refl                                                  */
    double  exponent, specularMultiplier;
    double  glossiness, deflectionScaling;   /* Refl::beginFrame, shading.h:197-201       */
    double  ior;
    int32_t layer_begin, layer_count;        /* Layered -> layers[]                       */
} frayhip_shader;

typedef struct frayhip_layer { int32_t shader, texture; float opacity[3]; int32_t _pad; } frayhip_layer;

enum { FRAYHIP_LIGHT_POINT = 0, FRAYHIP_LIGHT_RECT = 1 };
/* Light family, lights.h:32-99.  center/area are what RectLight::beginFrame computes
 * (lights.cpp:37-46); area keeps the reference's float*float rounding. */
typedef struct frayhip_light {
    int32_t kind, xSubd, ySubd, _pad;
    float   color[3], power;
    double  pos[3];
    frayhip_transform T;
    double  center[3];
    double  area;
} frayhip_light;

/* Camera, camera.h:37-55 (scene-file parameters; the per-frame corner vectors are derived by
 * the renderer exactly as Camera::beginFrame does, camera.cpp:34-57). */
typedef struct frayhip_camera {
    double pos[3];
    double yaw, pitch, roll, fov, aspectRatio, focalPlaneDist, fNumber, stereoSeparation;
    int32_t dof, autofocus, numDOFSamples, _pad;
    float  leftMask[3], rightMask[3];
} frayhip_camera;

/* GlobalSettings, scene.h:252-278 / scene.cpp:783-814. */
typedef struct frayhip_settings {
    int32_t frameWidth, frameHeight;
    float   ambientLight[3];
    int32_t wantAA, gi, maxTraceDepth, dbg;
    float   saturation;
    int32_t wantPrepass, numPaths, numThreads, interactive, fullscreen;
} frayhip_settings;

/* CubemapEnvironment, environment.h:50-78: faces in CubeOrder negx,negy,negz,posx,posy,posz. */
typedef struct frayhip_environment {
    int32_t present;             /* scene declares an environment                         */
    int32_t loaded;              /* all six faces decoded                                 */
    int32_t width[6], height[6];
    int64_t texel_offset[6];     /* in floats, into texels[]                              */
} frayhip_environment;

typedef struct frayhip_scene_desc {
    int32_t abi_version;
    int32_t n_nodes, n_geoms, n_planes, n_spheres, n_cubes, n_csgs, n_meshes;
    int32_t n_shaders, n_layers, n_textures, n_lights;
    int64_t n_texels;            /* floats */
    const frayhip_node*     nodes;
    const frayhip_geom_ref* geoms;
    const frayhip_plane*    planes;
    const frayhip_sphere*   spheres;
    const frayhip_cube*     cubes;
    const frayhip_csg*      csgs;
    const frayhip_mesh*     meshes;
    const frayhip_shader*   shaders;
    const frayhip_layer*    layers;
    const frayhip_texture*  textures;
    const frayhip_light*    lights;
    const float*            texels;
    frayhip_environment environment;
    frayhip_camera      camera;
    frayhip_settings    settings;
} frayhip_scene_desc;

/* ---- frame request ----------------------------------------------------------------------- */
enum {
    FRAYHIP_MODE_PRIMARY_ID = 0, /* camera ray through integer (x,y), closest hit only:
                                    the node/light loops of main.cpp:250-271               */
    FRAYHIP_MODE_RENDER     = 1, /* what render() does (main.cpp:373-405): Whitted, or path
                                    tracing when settings.gi                               */
};

/* Frame size is settings.frameWidth x frameHeight (the reference takes it from the SDL
 * surface created with those values, sdl.cpp:77-88, main.cpp:508).  Work is split in the
 * reference's 48x48 buckets (sdl.cpp:243-262); a call renders the buckets b with
 * b % bucket_stride == bucket_first, so N ranks with stride N cover the frame.  Pixels of other
 * buckets are left untouched in the output buffers.
 * Bucket b of a frame BW = ceil(W / 48) buckets wide is the one in bucket row by = b / BW and bucket
 * column bx = (b % BW + FRAYHIP_BUCKET_SKEW * by) % BW (frayhip_bucket_xy): every bucket row is rotated
 * against the one above it, so that a stride that divides BW (8 ranks on a 1920-wide frame: BW = 40)
 * deals diagonal stripes, not eight fixed sets of columns -- with plain row-major numbering rank r
 * would own the same five columns in every row, and the slowest rank's share of forest.fray was 8 %
 * above the mean. */
#define FRAYHIP_BUCKET_SKEW 3
typedef struct frayhip_frame {
    int32_t  mode;
    uint32_t seed;               /* RNG contract seed; the reference uses 42   */
    int32_t  bucket_first, bucket_stride;
    int32_t  spp_chunk;          /* path-tracing samples kept in flight per pixel; 0 = auto */
    int32_t  flags;              /* FRAYHIP_FRAME_* bits                                   */
} frayhip_frame;

enum {
    FRAYHIP_FRAME_STATS = 1,     /* also count rays / node tests / ... into frayhip_stats (uses the
                                    instrumented kernel variants: slower, same results)     */
};

typedef struct frayhip_stats {
    uint64_t closest_rays;       /* closest-hit queries (main.cpp:182-199 / 254-271)       */
    uint64_t shadow_rays;        /* visible() queries (main.cpp:64-80)                     */
    uint64_t node_tests;         /* Node::intersect calls                                  */
    uint64_t kd_inner_visits;    /* inner KD nodes entered                                 */
    uint64_t leaf_refs;          /* triangle indices read from leaves                      */
    uint64_t tri_tests;          /* Mesh::intersectTriangle calls                          */
    uint64_t prim_tests;         /* plane / sphere / cube / rect-light tests               */
    uint64_t smooth_hits;        /* winning hits that interpolated normals/uvs             */
    uint64_t samples;            /* camera samples                                         */
    uint64_t texture_fetches;
    double   ms_total;           /* wall time of the call                                  */
    double   ms_kernels;         /* device time between first and last kernel (HIP events) */
    double   ms_trace;           /* device time inside the dominant kernel: k_pt_bounce /
                                    k_whitted / k_primary (HIP events around each launch)   */
    uint64_t trace_launches;     /* number of launches summed into ms_trace                */
    double   alg_bytes_trace;    /* SURVEY 8(d) byte model evaluated on those launches       */
    /* path tracing only: the next-event shadow rays run in their own kernel (k_pt_shadow) */
    double   ms_shadow;          /* device time inside k_pt_shadow                           */
    uint64_t shadow_launches;
    double   alg_bytes_shadow;   /* byte model of the shadow-ray kernel                      */
    /* algorithmic FP64 operations of the same launches (Node::intersect ~90, triangle test
       ~45, box test ~30, primitive ~30 operations each), for the FP64-issue roofline      */
    double   alg_flops_trace;
    double   alg_flops_shadow;
} frayhip_stats;

/* ---- host scene layer (stands behind Scene::parseScene + Scene::beginRender,
 *      scene.cpp:751-767, main.cpp:503-514).  No GPU needed. -------------------------------- */
typedef struct frayhip_host_scene frayhip_host_scene;

/* Parses a .fray file (plus the OBJ/BMP/EXR files it names, relative to the scene file's
 * directory), runs the beginRender work (KD build, bump differentiate) and flattens. */
int  frayhip_scene_parse(const char* fray_path, frayhip_host_scene** out);
/* Mutable flattened view; the caller may edit .settings and .camera before creating a device
 * scene, which is how the reference's overrides are applied (scene.settings.* = ...). */
frayhip_scene_desc* frayhip_host_scene_desc(frayhip_host_scene* hs);
void frayhip_host_scene_free(frayhip_host_scene* hs);

/* ---- device side (stands behind render(), main.cpp:373) ----------------------------------- */
typedef struct frayhip_scene frayhip_scene;

/* Select the HIP device for this process (one process per GPU). */
int  frayhip_init(int device_id);
/* Deep-copies the description into device memory. */
int  frayhip_scene_create(const frayhip_scene_desc* desc, frayhip_scene** out);
void frayhip_scene_destroy(frayhip_scene* s);
/* Replaces the camera and the global settings of an uploaded scene without touching the geometry
 * (the reference's interactive loop mutates scene.camera between frames, main.cpp:437-491;
 * Camera::beginFrame re-derives everything per frame anyway).  Either pointer may be NULL. */
int  frayhip_scene_set_view(frayhip_scene* s, const frayhip_camera* camera, const frayhip_settings* settings);

/* Tunables of an uploaded scene (value ranges checked, FRAYHIP_E_ARG otherwise):
 *   "pt_lanes"      1..4   path-tracing batches in flight at once, each on its own HIP stream (default 4;
 *                          1 serialises every launch, which is what a per-kernel profile wants)
 *   "pt_budget_mib" MiB of device memory a path-traced frame may use for its queues (about 340 B per path in
 *                          flight; default 24576): a frame is cut into batches of samples that fit
 *   "speculate_fans" 0 / 1  glossy reflections of eight or more samples at depth 0 (Reflection::shade, shading.cpp:172-204) in a scene
 *                          whose lights draw no random numbers: the fan's directions are drawn ahead and its rays traced as work items of
 *                          their own, then looked up while none of them drew (default 1; the picture is the same either way,
 *                          hw9/dragon.fray 1080p 16.4 -> 7.8 ms)
 *   "fused_whitted_max" 0..1024  a Whitted frame of a scene without recursive shaders and without KD meshes whose lights take at most this many
 *                          samples per hit (default 4) asks visible() inside the shading kernel -- one launch instead of seed + shade + visible +
 *                          gather + resolve: zaphod.fray 1080p 0.30 -> 0.17 ms; 0 = always the separate launches (the picture is the same)
 *   "fp_contract"   0 / 1  0 (default): the reference's arithmetic everywhere (no fused multiply-add, IEEE division and square root, correctly
 *                          rounded sin / cos / acos): hit records AND colours are the CPU reference build's, bit for bit.  1: path tracing only -- every
 *                          bounce AFTER a camera sample's first closest hit and every next-event visibility query run kernels built with
 *                          -ffp-contract=fast, reciprocal / reciprocal-square-root with two refinement steps, plain-double sin / cos, and
 *                          sqrt(1 - c^2) for sin(acos c).  Primary hit records (MODE_PRIMARY_ID) and a sample's first bounce stay exact; shaded colour
 *                          stays inside 1e-4 RMS per channel (measured: 0 of 2 073 600 pixels of the 1080p x 64 spp Cornell frame differ at all, since
 *                          FP64 differences of 1e-16 vanish where geometry becomes an FP32 colour factor); cornell 1080p x 64 spp 92.4 -> 81.6 ms
 * The environment variables FRAYHIP_PT_LANES / FRAYHIP_PT_BUDGET_MIB / FRAYHIP_SPECULATE_FANS / FRAYHIP_FP_CONTRACT preset them at frayhip_scene_create. */
int  frayhip_scene_set_option(frayhip_scene* s, const char* name, int64_t value);
/* Reads an option back, or one of the last frame's read-only figures: "fans_filed" (camera samples whose first fan was drawn ahead),
 * "fan_children" (rays traced ahead), "fan_children_looked_up" (results used), "fans_given_up" (fans in which a ray drew a random
 * number after all, so that the rest of the fan was traced in place), "contracted_launches" (launches of the last frame that ran a kernel of
 * the "fp_contract" build), "whitted_path" (how the last Whitted frame ran: 0 = the recursive kernel, 1 = shade / visible / gather launches, 2 = fused), "pt_budget_effective_mib" (the queue budget frames currently plan with: pt_budget_mib clamped to the device's
 * free memory, halved when an allocation failed and the frame could be planned again). */
int  frayhip_scene_get_option(frayhip_scene* s, const char* name, int64_t* value);

/* Threads: a frayhip_scene renders one frame at a time (it owns one workspace and one set of
 * counters), as the reference calls render() from one thread at a time (main.cpp:407-412,448);
 * different scenes may be driven from different threads.  frayhip_last_error() is per thread. */
/* Blocking render.  Any output pointer may be NULL.  Host buffers, row-major:
 *   rgb      W*H*3 float  -- `vfb` (main.cpp:53,360), linear, unclamped
 *   hit_id   W*H   int32  -- node index, -1 miss, -2-i light i   (MODE_PRIMARY_ID)
 *   hit_dist W*H   double -- world distance, 1e99 on a miss       (MODE_PRIMARY_ID)          */
int  frayhip_render(frayhip_scene* s, const frayhip_frame* f,
                    float* rgb, int32_t* hit_id, double* hit_dist, frayhip_stats* st);
/* Same, with DEVICE pointers (hipMalloc'ed by the caller, e.g. a torch tensor's data_ptr);
 * work is enqueued on `hip_stream` (a hipStream_t, NULL = default stream) and the call
 * returns after the stream has been synchronised.  A path-traced frame also runs batches on
 * streams of its own; they start after everything already enqueued on `hip_stream` and are
 * joined back into it before the call returns, so the caller sees one stream's ordering. */
int  frayhip_render_device(frayhip_scene* s, const frayhip_frame* f,
                           float* d_rgb, int32_t* d_hit_id, double* d_hit_dist,
                           void* hip_stream, frayhip_stats* st);

/* Multi-GPU tile exchange helpers.  pack: gathers this rank's buckets from a
 * full-frame device buffer into a compact bucket-major buffer of
 * frayhip_bucket_count(W,H,first,stride) * 48*48*channels floats; unpack is the inverse and
 * is run by the gathering rank once per peer. */
int  frayhip_bucket_count(int width, int height, int bucket_first, int bucket_stride);
int  frayhip_bucket_xy(int width, int height, int bucket, int* bx, int* by);   /* bucket column and row of bucket b (the rule above) */
int  frayhip_pack_buckets_device(const float* d_frame, float* d_packed, int width, int height,
                                 int channels, int bucket_first, int bucket_stride, void* hip_stream);
int  frayhip_unpack_buckets_device(const float* d_packed, float* d_frame, int width, int height,
                                   int channels, int bucket_first, int bucket_stride, void* hip_stream);

/* The exchange itself, inside the library: rank r (of `world`) has rendered the buckets b with b % world == r into
 * its own full-size device frame (frayhip_frame.bucket_first = r, bucket_stride = world); frayhip_gather_buckets
 * moves every other rank's buckets into rank `root`'s frame -- pack, direct peer -> root RCCL transfers over xGMI
 * (grouped ncclRecv on the root, one ncclSend per peer), unpack -- enqueued on `hip_stream` of each rank.  Collective:
 * every rank of the communicator calls it.  One process per GPU; frayhip_init(device) first.
 *   frayhip_comm_unique_id   rank 0 obtains the 128-byte ncclUniqueId and hands it to the other ranks by whatever
 *                            means the host has (MPI, a file, torch.distributed, a socket)
 *   frayhip_comm_create      every rank, with the same id; world == 1 needs no id and no RCCL
 *   frayhip_comm_from_nccl   wraps an ncclComm_t the host already owns (not destroyed by frayhip_comm_destroy); FRAYHIP_E_ARG when the
 *                            communicator's own size / rank (ncclCommCount / ncclCommUserRank) are not the caller's
 *   frayhip_comm_ranks       the number of ranks RCCL itself sees in the communicator (ncclCommCount; 1 for a world of one made without
 *                            RCCL), negative error code on failure: what a host prints to show that the exchange really spans N GPUs
 *   frayhip_comm_available   1 when RCCL can be bound in this process, 0 otherwise: frayhip_comm_create blocks inside
 *                            ncclCommInitRank until EVERY rank has entered it, so the ranks agree on this first
 *   frayhip_comm_library     the file the RCCL entry points were bound from ("" when none could be): which RCCL carries the frames
 * RCCL is bound at run time (an RCCL the process already holds is used, none is loaded beside it; the environment variable
 * FRAYHIP_RCCL_LIBRARY, read at the first comm call, names a specific build to bind instead): FRAYHIP_E_UNSUPPORTED
 * when the host has none.  Gathers on one communicator share its staging buffer; a gather waits, on its own stream,
 * for the previous gather of that communicator, so they may be issued on different streams. */
#define FRAYHIP_COMM_ID_BYTES 128
typedef struct frayhip_comm frayhip_comm;
int  frayhip_comm_available(void);
const char* frayhip_comm_library(void);
int  frayhip_comm_unique_id(void* id128);
int  frayhip_comm_create(const void* id128, int rank, int world, frayhip_comm** out);
int  frayhip_comm_from_nccl(void* nccl_comm, int rank, int world, frayhip_comm** out);
int  frayhip_comm_ranks(frayhip_comm* c);
void frayhip_comm_destroy(frayhip_comm* c);
int  frayhip_gather_buckets(frayhip_comm* c, float* d_frame, int width, int height, int channels, int root, void* hip_stream);

/* vfb -> RGB32 with clamp, no gamma (displayVFB, sdl.cpp:63-74; Color::toRGB32, color.h:59-65). */
int  frayhip_to_rgb32(const float* rgb, uint32_t* out, int n_pixels);

/* Writes a float RGB frame as a 24-bit BMP the way Bitmap::saveBMP does (bitmap.cpp:197-236:
 * clamp + round per toRGB32, bottom-up rows padded to 4 bytes). */
int  frayhip_save_bmp(const char* path, const float* rgb, int width, int height);

/* Test hook: runs the device restatement of the reference's random numbers (std::mt19937 +
 * libstdc++ distributions, random_generator.cpp:41-80) for one seed and returns, for i < n,
 * the i-th randfloat() of a fresh generator, the i-th randdouble() of a second and the i-th
 * randint(0, int_hi) of a third (host buffers, any may be NULL; n <= 4096, which crosses the
 * generator's 227-word register window and two full state twists). */
int  frayhip_debug_rng(uint32_t seed, int n, float* floats, double* doubles, int32_t* ints, int int_hi);

/* Test hook: the device's sin / cos / acos behind hemisphereSample / unitDiscSample (main.cpp:92-116,
 * random_generator.cpp:71-80; the reference takes them from glibc, the device code has its own correctly rounded
 * ones, fray_amd/csrc/dev_trig.hpp) -- sincos(x[i]) and acos(fold(x[i])), so that a test can state how often they
 * equal the host's.  fold(x) = x - 2*floor(x/2) - 1 in [-1, 1) is returned in acos_arg.  Host buffers of n doubles,
 * any output may be NULL. */
int  frayhip_debug_libm(int n, const double* x, double* sin_out, double* cos_out, double* acos_out, double* acos_arg);

const char* frayhip_last_error(void);
int  frayhip_abi_version(void);
/* sizeof() of a struct of this header by name ("frayhip_mesh", ...), -1 if unknown: lets a
 * foreign-language binding check its mirror of the layouts. */
int  frayhip_sizeof(const char* struct_name);

#ifdef __cplusplus
}
#endif
#endif /* FRAYHIP_H */
