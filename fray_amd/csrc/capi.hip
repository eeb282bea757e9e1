// C ABI, device side: scene upload, frame set-up (Camera::beginFrame), kernel launches, timing.
// Stands behind render() of the reference (src/main.cpp:373-405).
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstddef>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>
#include <functional>
#include <cstdlib>


#include "render_state.hpp"
#include "dev_rng.hpp"
#include "dev_pack.hpp"

namespace {

using frayhip_detail::set_error;

// One device allocation holding every read-only table of a scene.
struct Arena {
    std::vector<unsigned char> host;
    size_t add(const void* p, size_t bytes, size_t align = 256)
    {
        size_t off = (host.size() + align - 1) / align * align;
        host.resize(off + bytes);
        if (bytes) memcpy(host.data() + off, p, bytes);
        return off;
    }
};

void put3(double* o, const double* p) { o[0] = p[0]; o[1] = p[1]; o[2] = p[2]; }
void putX(DXform& X, const frayhip_transform& T)
{
    put3(X.off, T.offset);
    memcpy(X.m, T.m, sizeof X.m);
    memcpy(X.inv, T.invM, sizeof X.inv);
}

}  // namespace

namespace frayhip_detail {

// Camera::beginFrame, camera.cpp:34-57 (host, FP64; sin/cos/tan from the host libm like the reference).
void matmul3(const double* a, const double* b, double* c)
{
    for (int i = 0; i < 9; i++) c[i] = 0.0;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            for (int k = 0; k < 3; k++) c[i * 3 + j] += a[i * 3 + k] * b[k * 3 + j];
}
void rowmul(const double* v, const double* m, double* o)
{
    for (int j = 0; j < 3; j++) o[j] = v[0] * m[j] + v[1] * m[3 + j] + v[2] * m[6 + j];
}
DCamera camera_begin_frame(const frayhip_camera& c, int W, int H)
{
    const double PI = 3.141592653589793238;
    auto rad = [&](double a) { return a / 180.0 * PI; };
    DCamera f{};
    const double aspect = c.aspectRatio;
    const double bc[3] = {-aspect - 0.0, 1.0 - 0.0, 1.0 - 1.0};
    const double lenBC = sqrt(bc[0] * bc[0] + bc[1] * bc[1] + bc[2] * bc[2]);
    const double m = tan(rad(c.fov / 2)) / lenBC;
    const double tl[3] = {-aspect * m, +m, 1}, tr[3] = {+aspect * m, +m, 1}, bl[3] = {-aspect * m, -m, 1};
    // rotationAroundZ / X / Y (matrix.cpp:29-62) take `sin(angle)` and `cos(angle)`; the reference's build (g++ -O2) merges the pair into ONE call of
    // glibc's sincos(), whose sine is not sin()'s in the last place for one angle in 700 (-7.93, -7.84, -19.99 degrees ...).  This file is compiled by
    // clang, which keeps two calls: ask for sincos() by name, so that the camera is the reference's whatever compiles it.
    double S, C;
    sincos(rad(c.roll), &S, &C);
    const double rz[9] = {C, -S, 0, S, C, 0, 0, 0, 1};
    sincos(rad(c.pitch), &S, &C);
    const double rx[9] = {1, 0, 0, 0, C, -S, 0, S, C};
    sincos(rad(c.yaw), &S, &C);
    const double ry[9] = {C, 0, S, 0, 1, 0, -S, 0, C};
    double t[9], rot[9];
    matmul3(rz, rx, t);
    matmul3(t, ry, rot);
    rowmul(tl, rot, f.topLeft);
    rowmul(tr, rot, f.topRight);
    rowmul(bl, rot, f.bottomLeft);
    const double ez[3] = {0, 0, 1}, ey[3] = {0, 1, 0}, ex[3] = {1, 0, 0};
    rowmul(ez, rot, f.frontDir);
    rowmul(ey, rot, f.upDir);
    rowmul(ex, rot, f.rightDir);
    put3(f.pos, c.pos);
    f.w = W; f.h = H;
    f.apertureSize = 1.0 / c.fNumber;
    f.focalPlaneDist = c.focalPlaneDist;
    f.stereoSeparation = c.stereoSeparation;
    memcpy(f.leftMask, c.leftMask, sizeof f.leftMask);
    memcpy(f.rightMask, c.rightMask, sizeof f.rightMask);
    f.dof = c.dof;
    return f;
}

// Grid of a persistent kernel (k_primary, k_whitted: waves claim tiles from DCursors): exactly the
// blocks that are resident at once -- a 256-thread block is one wave per SIMD, so `wavesPerSimd`
// blocks per compute unit.
int persistent_grid(size_t n, int wavesPerSimd)
{
    static int cus = 0;
    if (!cus) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
        else cus = 256;
    }
    size_t blocks = (n + 255) / 256, cap = (size_t)cus * wavesPerSimd;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

#ifndef FRAY_BOUNCE_BLOCKS
#define FRAY_BOUNCE_BLOCKS 2048
#endif
#ifndef FRAY_BOUNCE_PATHS_PER_BLOCK
#define FRAY_BOUNCE_PATHS_PER_BLOCK 8192
#endif
#ifndef FRAY_BOUNCE_BLOCKS_ALONE
#define FRAY_BOUNCE_BLOCKS_ALONE 8192
#endif
static_assert(FRAY_BOUNCE_BLOCKS * 4 <= FRAY_MAXSEG && FRAY_BOUNCE_BLOCKS_ALONE * 4 <= FRAY_MAXSEG, "every wave of the bounce / shadow grid owns one segment of the queue tables (QMeta)");
// Every wave of the bounce / shadow grid gets an equal share of the queue.  With four batches in flight the frame is fastest at 2 048 blocks (smaller shares add
// instructions, and the other lanes' blocks fill a launch's tail anyway: profiles/r04_experiments/README.md K).  The Cube / CSG variants run their batches one at a time
// (`alone`), where the tail is the chip standing empty: 8 192 blocks, csg_nested path traced 79.8 -> 67.4 ms.
// Beside other batches a block should get at least 8 192 paths (2 048 per wave): one rank's share of an 8-rank frame (5.7 M paths per batch) is fastest at 768-1 024 blocks
// (12.6-12.7 ms against 13.3 at 2 048, 15.9 at 4 096), a quarter frame at 1 024-1 536, the whole frame (16.6 M) at 2 048.
int bounce_grid(size_t n, bool alone)
{
    size_t blocks = (n + 255) / 256;
    const size_t cap = alone ? (size_t)FRAY_BOUNCE_BLOCKS_ALONE : std::min<size_t>(FRAY_BOUNCE_BLOCKS, std::max<size_t>(256, n / FRAY_BOUNCE_PATHS_PER_BLOCK));
    if (blocks > cap) blocks = cap;
    return (int)(blocks < 1 ? 1 : blocks);
}

int seed_grid(size_t n)          // k_seed: many short threads
{
    // forest DOF 256: 155.2 ms at 2 048 blocks, 153.6 at 32 768 (a thread runs few chains and the last waves leave together).  FRAYHIP_SEED_BLOCKS is a
    // development knob, read once per process and validated like the others: 1..65535, anything else keeps the default
    static const long cap = [] { const char* e = getenv("FRAYHIP_SEED_BLOCKS"); const long v = e ? atol(e) : 0; return v >= 1 && v <= 65535 ? v : 32768L; }();
    size_t blocks = (n + 255) / 256;
    if (blocks > (size_t)cap) blocks = (size_t)cap;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

int grid_for(size_t n)
{
    size_t blocks = (n + 255) / 256;
    const size_t cap = 256 * 8;   // 256 CUs x 8 blocks of 256 threads: the chip's full wave capacity
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

int ensure_work(frayhip_scene* sc, size_t bytes)
{
    if (sc->work_bytes >= bytes) return FRAYHIP_OK;
    if (sc->d_work) (void)hipFree(sc->d_work);
    sc->d_work = nullptr;
    sc->work_bytes = 0;
    if (hipMalloc(&sc->d_work, bytes) != hipSuccess) { set_error("frayhip_render: out of device memory for the work queues (" + std::to_string(bytes >> 20) + " MiB)"); return FRAYHIP_E_NOMEM; }
    sc->work_bytes = bytes;
    return FRAYHIP_OK;
}

// The queue budget a frame may plan with: the tunable (pt_budget_mib), but never more than four fifths of what the device could give when the
// clamp was taken (free memory plus the workspace this scene already holds) -- on a smaller or busy GPU, or with several ranks sharing one for a
// rehearsal, the frame is then cut into smaller batches instead of failing with E_NOMEM.  The clamp is taken ONCE (first use after scene creation
// or after a change of the option): asking hipMemGetInfo every frame made the batch size follow other processes' allocations, and every growth of
// the workspace is a hipFree + hipMalloc in the middle of a run.  Ranks that share a GPU all see the same free memory at the same moment and may
// each plan with four fifths of it: an allocation that fails all the same halves the budget and the frame is planned again (ensure_work_or_shrink).
size_t work_budget(frayhip_scene* sc)
{
    if (!sc->ptBudgetEff) {
        size_t freeB = 0, totalB = 0;
        size_t b = sc->ptBudgetBytes;
        if (hipMemGetInfo(&freeB, &totalB) == hipSuccess) b = std::min(b, (freeB + sc->work_bytes) / 5 * 4);
        sc->ptBudgetEff = std::max<size_t>(b, (size_t)64 << 20);
    }
    return sc->ptBudgetEff;
}

// ensure_work for a plan made under work_budget(): FRAYHIP_OK, an error, or FRAYHIP_RETRY_SMALLER after halving the budget (the caller plans again).
// `canRetry`: the caller will plan again with the smaller budget (it is not pinned by frayhip_frame.spp_chunk and is not already at one sample per
// batch); only then is the halved budget kept -- a failure the caller can do nothing about must not shrink every later frame's batches (it is
// reported as FRAYHIP_E_NOMEM and the budget stays; option "pt_budget_effective_mib" reads what frames currently plan with).
int ensure_work_or_shrink(frayhip_scene* sc, size_t bytes, bool canRetry)
{
    const int rc = ensure_work(sc, bytes);
    if (rc != FRAYHIP_E_NOMEM) return rc;
    (void)hipGetLastError();
    if (!canRetry || work_budget(sc) <= ((size_t)64 << 20)) return rc;          // nothing to plan again, or already at the floor: give up with the allocation's message
    sc->ptBudgetEff = std::max<size_t>(sc->ptBudgetEff / 2, (size_t)64 << 20);
    return FRAYHIP_RETRY_SMALLER;
}

hipEvent_t pool_event(std::vector<hipEvent_t>& pool, size_t i)
{
    while (pool.size() <= i) {
        hipEvent_t e = nullptr;
        hipError_t rc = hipEventCreate(&e);
        if (rc != hipSuccess) { set_error(std::string("hipEventCreate: ") + hipGetErrorString(rc)); return nullptr; }
        pool.push_back(e);
    }
    return pool[i];
}

}  // namespace frayhip_detail

namespace {

using frayhip_detail::kStatsBytes;
using frayhip_detail::grid_for;

bool shader_uses_uv(const frayhip_scene_desc& d, int s, int depth = 0)
{
    if (s < 0 || s >= d.n_shaders || depth > 40) return false;
    const frayhip_shader& sh = d.shaders[s];
    auto texUV = [&](int t) { return t >= 0 && t < d.n_textures && d.textures[t].kind != FRAYHIP_TEX_FRESNEL; };
    if (texUV(sh.texture)) return true;
    if (sh.kind == FRAYHIP_SHADER_LAYERED)
        for (int i = 0; i < sh.layer_count; i++) {
            const frayhip_layer& L = d.layers[sh.layer_begin + i];
            if (texUV(L.texture) || shader_uses_uv(d, L.shader, depth + 1)) return true;
        }
    return false;
}

bool create_lanes(frayhip_scene* sc)
{
    if (hipEventCreateWithFlags(&sc->evLaneStart, hipEventDisableTiming) != hipSuccess) return false;
    for (int k = 0; k < FRAY_PT_LANES; k++) {
        if (k > 0 && hipStreamCreateWithFlags(&sc->laneStream[k], hipStreamNonBlocking) != hipSuccess) return false;
        if (hipEventCreateWithFlags(&sc->evResolved[k], hipEventDisableTiming) != hipSuccess) return false;
    }
    return true;
}

// A description can come from any host, not only from frayhip_scene_parse: every
// index the kernels will follow is range-checked here, because an out-of-range one would be a wild
// device read.  Returns an empty string when the description is sound.
std::string validate_desc(const frayhip_scene_desc& d)
{
    auto bad = [](const char* what, long long i) { return std::string("frayhip_scene_create: ") + what + " (element " + std::to_string(i) + ")"; };
    const int32_t counts[] = {d.n_nodes, d.n_geoms, d.n_planes, d.n_spheres, d.n_cubes, d.n_csgs, d.n_meshes, d.n_shaders, d.n_layers, d.n_textures, d.n_lights};
    for (int32_t c : counts) if (c < 0) return "frayhip_scene_create: negative element count";
    if (d.n_texels < 0) return "frayhip_scene_create: negative texel count";
    const void* arrays[] = {d.nodes, d.geoms, d.planes, d.spheres, d.cubes, d.csgs, d.meshes, d.shaders, d.layers, d.textures, d.lights};
    for (int k = 0; k < 11; k++) if (counts[k] > 0 && !arrays[k]) return "frayhip_scene_create: null array with a non-zero count";
    if (d.n_texels > 0 && !d.texels) return "frayhip_scene_create: null texel pool";
    const int32_t perKind[5] = {d.n_planes, d.n_spheres, d.n_cubes, d.n_meshes, d.n_csgs};
    for (int i = 0; i < d.n_geoms; i++) {
        const frayhip_geom_ref& g = d.geoms[i];
        if (g.kind < 0 || g.kind > 4 || g.index < 0 || g.index >= perKind[g.kind]) return bad("geometry reference out of range", i);
    }
    for (int i = 0; i < d.n_csgs; i++) {
        const frayhip_csg& c = d.csgs[i];
        if (c.op < 0 || c.op > 2 || c.left < 0 || c.left >= d.n_geoms || c.right < 0 || c.right >= d.n_geoms) return bad("CSG operand out of range", i);
    }
    auto texel_range_ok = [&](int64_t off, int32_t w, int32_t h) {
        if (w < 0 || h < 0 || off < 0) return false;
        return off + (int64_t)w * h * 3 <= d.n_texels;
    };
    for (int i = 0; i < d.n_textures; i++) {
        const frayhip_texture& t = d.textures[i];
        if (t.kind < 0 || t.kind > 3) return bad("unknown texture kind", i);
        if ((t.kind == FRAYHIP_TEX_BITMAP || t.kind == FRAYHIP_TEX_BUMP) && (t.width <= 0 || t.height <= 0)) return bad("bitmap texture without texels", i);   // the lookup wraps modulo width / height
        if ((t.kind == FRAYHIP_TEX_BITMAP || t.kind == FRAYHIP_TEX_BUMP) && !texel_range_ok(t.texel_offset, t.width, t.height)) return bad("texture texels outside the pool", i);
    }
    for (int i = 0; i < d.n_layers; i++) {
        const frayhip_layer& L = d.layers[i];
        if (L.shader < 0 || L.shader >= d.n_shaders || L.texture < -1 || L.texture >= d.n_textures) return bad("layer reference out of range", i);
    }
    for (int i = 0; i < d.n_shaders; i++) {
        const frayhip_shader& sh = d.shaders[i];
        if (sh.kind < 0 || sh.kind > 5) return bad("unknown shader kind", i);
        if (sh.texture < -1 || sh.texture >= d.n_textures) return bad("shader texture out of range", i);
        if (sh.kind == FRAYHIP_SHADER_LAYERED && (sh.layer_begin < 0 || sh.layer_count < 0 || (int64_t)sh.layer_begin + sh.layer_count > d.n_layers)) return bad("layer range out of bounds", i);
        if (sh.kind == FRAYHIP_SHADER_REFL && sh.numSamples < 0) return bad("negative numSamples", i);
    }
    for (int i = 0; i < d.n_nodes; i++) {
        const frayhip_node& n = d.nodes[i];
        if (n.geom < 0 || n.geom >= d.n_geoms || n.shader < 0 || n.shader >= d.n_shaders || n.bump_tex < -1 || n.bump_tex >= d.n_textures) return bad("node reference out of range", i);
    }
    for (int i = 0; i < d.n_lights; i++) {
        const frayhip_light& L = d.lights[i];
        if (L.kind < 0 || L.kind > 1) return bad("unknown light kind", i);
        if (L.kind == FRAYHIP_LIGHT_RECT && (L.xSubd <= 0 || L.ySubd <= 0 || (int64_t)L.xSubd * L.ySubd > (1 << 20))) return bad("bad RectLight subdivision", i);
    }
    if (d.environment.present && d.environment.loaded)
        for (int f = 0; f < 6; f++)
            if (d.environment.width[f] <= 0 || d.environment.height[f] <= 0 || !texel_range_ok(d.environment.texel_offset[f], d.environment.width[f], d.environment.height[f]))
                return bad("environment face outside the texel pool", f);
    for (int i = 0; i < d.n_meshes; i++) {
        const frayhip_mesh& m = d.meshes[i];
        if (m.n_vertices < 0 || m.n_normals < 0 || m.n_uvs < 0 || m.n_triangles < 0 || m.n_kdnodes < 0 || m.n_trirefs < 0) return bad("negative mesh count", i);
        if ((m.n_vertices && !m.vertices) || (m.n_normals && !m.normals) || (m.n_uvs && !m.uvs) || (m.n_triangles && !m.triangles) ||
            (m.n_kdnodes && !m.kdnodes) || (m.n_trirefs && !m.trirefs)) return bad("null mesh array with a non-zero count", i);
        for (int t = 0; t < m.n_triangles; t++) {
            const frayhip_triangle& T = m.triangles[t];
            for (int k = 0; k < 3; k++) {
                if (T.v[k] < 0 || T.v[k] >= m.n_vertices) return bad("triangle vertex index out of range in mesh", i);
                if (m.n_normals > 0 && (T.n[k] < 0 || T.n[k] >= m.n_normals)) return bad("triangle normal index out of range in mesh", i);
                if (m.n_uvs > 0 && (T.t[k] < 0 || T.t[k] >= m.n_uvs)) return bad("triangle uv index out of range in mesh", i);
            }
        }
        if (m.has_kd) {
            if (m.n_kdnodes <= 0) return bad("has_kd without nodes in mesh", i);
            std::vector<int> depth((size_t)m.n_kdnodes, 0);
            for (int k = 0; k < m.n_kdnodes; k++) {
                const frayhip_kdnode& n = m.kdnodes[k];
                if (n.parent < -1 || n.parent >= k || (k == 0) != (n.parent == -1)) return bad("KD parent link is not a tree in mesh", i);
                // the walk's stack of pending children holds one entry per level (dev_trace.hpp); the reference's builder stops at 65 (constants.h:39)
                if (k > 0 && (depth[k] = depth[n.parent] + 1) >= FRAY_KD_MAX_DEPTH) return bad(("KD tree deeper than the walk's stack (" + std::to_string(FRAY_KD_MAX_DEPTH) + " levels) in mesh").c_str(), i);
                if (n.axis == 3) {
                    if (n.tri_begin < 0 || n.tri_count < 0 || (int64_t)n.tri_begin + n.tri_count > m.n_trirefs) return bad("KD leaf range out of bounds in mesh", i);
                } else if (n.axis >= 0 && n.axis <= 2) {
                    // children lie after their parent (pre-order) and point back to it: the stackless walk climbs through these links
                    if (n.child0 <= k || (int64_t)n.child0 + 1 >= m.n_kdnodes || m.kdnodes[n.child0].parent != k || m.kdnodes[n.child0 + 1].parent != k)
                        return bad("KD child link is not a tree in mesh", i);
                } else return bad("bad KD axis in mesh", i);
            }
            for (int r = 0; r < m.n_trirefs; r++) if (m.trirefs[r] < 0 || m.trirefs[r] >= m.n_triangles) return bad("KD triangle reference out of range in mesh", i);
        }
    }
    if (d.settings.frameWidth <= 0 || d.settings.frameHeight <= 0) return "frayhip_scene_create: bad frame size";
    return std::string();
}

}  // namespace

extern "C" {

int frayhip_init(int device_id)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) { set_error("frayhip_init: no HIP device available"); return FRAYHIP_E_NODEVICE; }
    if (device_id < 0 || device_id >= n) { set_error("frayhip_init: bad device id"); return FRAYHIP_E_ARG; }
    HIP_TRY(hipSetDevice(device_id));
    return FRAYHIP_OK;
}

int frayhip_scene_create(const frayhip_scene_desc* desc, frayhip_scene** out)
{
    if (!desc || !out) { set_error("frayhip_scene_create: null argument"); return FRAYHIP_E_ARG; }
    if (desc->abi_version != FRAYHIP_ABI_VERSION) { set_error("frayhip_scene_create: ABI version mismatch"); return FRAYHIP_E_ARG; }
    const frayhip_scene_desc& d = *desc;
    {
        const std::string why = validate_desc(d);
        if (!why.empty()) { set_error(why); return FRAYHIP_E_ARG; }
    }
    // ---- what the device path implements ----
    {   // CsgOp trees: the device unrolls the Geometry::intersect recursion FRAY_CSG_DEPTH levels deep
        std::vector<int> levels(d.n_csgs, 0);      // 0 = not computed yet
        std::function<int(int, int)> depth = [&](int i, int guard) -> int {
            if (guard > d.n_csgs) return 1 << 20;   // a cycle cannot come out of the parser, but a hand-made description could hold one
            if (levels[i]) return levels[i];
            int m = 1;
            for (int side = 0; side < 2; side++) {
                const frayhip_geom_ref& g = d.geoms[side == 0 ? d.csgs[i].left : d.csgs[i].right];
                if (g.kind == FRAYHIP_GEOM_CSG) m = std::max(m, 1 + depth(g.index, guard + 1));
            }
            return levels[i] = m;
        };
        for (int i = 0; i < d.n_csgs; i++)
            if (depth(i, 0) > FRAY_CSG_DEPTH) {
                set_error("frayhip_scene_create: CSG operands nested more than " + std::to_string(FRAY_CSG_DEPTH) + " levels deep are not implemented on the device path");
                return FRAYHIP_E_UNSUPPORTED;
            }
    }
    frayhip_scene* sc = new frayhip_scene();
    for (int i = 0; i < d.n_nodes; i++) {
        int k = d.geoms[d.nodes[i].geom].kind;
        if (k == FRAYHIP_GEOM_CUBE || k == FRAYHIP_GEOM_CSG) sc->extGeometry = true;   // selects the <ST | 2> kernel variants
    }
    Arena A;
    // nodes
    std::vector<DNode> nodes(d.n_nodes);
    for (int i = 0; i < d.n_nodes; i++) {
        const frayhip_node& n = d.nodes[i];
        putX(nodes[i].T, n.T);
        nodes[i].geomKind = d.geoms[n.geom].kind;
        nodes[i].geomIndex = d.geoms[n.geom].index;
        nodes[i].shader = n.shader;
        nodes[i].bumpTex = n.bump_tex;
        nodes[i].xfClass = i;
        {
            static const double I9[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, Z3[3] = {0, 0, 0};       // +0.0 everywhere: a -0.0 in the file's transform is not the identity
            nodes[i].xfIdentity = (!memcmp(n.T.offset, Z3, sizeof Z3) && !memcmp(n.T.m, I9, sizeof I9) && !memcmp(n.T.invM, I9, sizeof I9)) ? 1 : 0;
        }
        for (int j = 0; j < i; j++)
            if (!memcmp(d.nodes[j].T.offset, n.T.offset, sizeof n.T.offset) && !memcmp(d.nodes[j].T.invM, n.T.invM, sizeof n.T.invM) &&
                !memcmp(d.nodes[j].T.m, n.T.m, sizeof n.T.m)) {
                nodes[i].xfClass = nodes[j].xfClass;
                break;
            }
        int sk = d.shaders[n.shader].kind;
        if (sk == FRAYHIP_SHADER_REFL || sk == FRAYHIP_SHADER_REFR || sk == FRAYHIP_SHADER_LAYERED) sc->whittedNeedsRecursion = true;
    }
    size_t oNodes = A.add(nullptr, 0);                  // filled after the meshes: tree-less nodes hold a device pointer
    A.host.resize(oNodes + nodes.size() * sizeof(DNode));
    size_t oNodesX = A.add(nullptr, 0);
    A.host.resize(oNodesX + nodes.size() * sizeof(DNodeX));
    size_t oGates = A.add(nullptr, 0);                  // the path tracer's scheduling hint (DGate), filled with the nodes
    A.host.resize(oGates + FRAY_MAX_GATES * sizeof(DGate));
    std::vector<DPlane> planes(d.n_planes);
    for (int i = 0; i < d.n_planes; i++) { planes[i].limit = d.planes[i].limit; planes[i].height = d.planes[i].height; }
    size_t oPlanes = A.add(planes.data(), planes.size() * sizeof(DPlane));
    std::vector<DSphere> spheres(d.n_spheres);
    for (int i = 0; i < d.n_spheres; i++) { put3(spheres[i].O, d.spheres[i].O); spheres[i].R = d.spheres[i].R; }
    size_t oSpheres = A.add(spheres.data(), spheres.size() * sizeof(DSphere));
    std::vector<DCube> cubes(d.n_cubes);
    for (int i = 0; i < d.n_cubes; i++) { put3(cubes[i].O, d.cubes[i].O); cubes[i].halfSide = d.cubes[i].halfSide; }
    size_t oCubes = A.add(cubes.data(), cubes.size() * sizeof(DCube));
    std::vector<DCsg> csgs(d.n_csgs);
    for (int i = 0; i < d.n_csgs; i++) {
        const frayhip_csg& g = d.csgs[i];
        csgs[i].op = g.op;
        csgs[i].leftKind = d.geoms[g.left].kind; csgs[i].leftIndex = d.geoms[g.left].index; csgs[i].leftGeom = g.left;
        csgs[i].rightKind = d.geoms[g.right].kind; csgs[i].rightIndex = d.geoms[g.right].index; csgs[i].rightGeom = g.right;
        csgs[i].flat = (csgs[i].leftKind <= FRAYHIP_GEOM_CUBE && csgs[i].rightKind <= FRAYHIP_GEOM_CUBE) ? 1 : 0;
    }
    size_t oCsgs = A.add(csgs.data(), csgs.size() * sizeof(DCsg));
    // meshes
    std::vector<DMesh> meshes(d.n_meshes);
    struct MeshOff { size_t tris, attrs, kd, kdBox, refs, ltris, ltris32; };
    std::vector<MeshOff> moff(d.n_meshes);
    for (int mi = 0; mi < d.n_meshes; mi++) {
        const frayhip_mesh& m = d.meshes[mi];
        DMesh& M = meshes[mi];
        put3(M.bmin, m.bbox_min);
        put3(M.bmax, m.bbox_max);
        M.boxMax = 0;
        for (int k = 0; k < 3; k++) M.boxMax = std::max(M.boxMax, std::max(std::fabs(m.bbox_min[k]), std::fabs(m.bbox_max[k])));
        M.nTris = m.n_triangles;
        M.hasKd = m.has_kd;
        if (m.has_kd) sc->kdMeshes = true;
        M.smooth = !(m.faceted || m.n_normals == 0);
        M.culling = m.backfaceCulling;
        M.hasUV = m.n_uvs != 0;
        std::vector<DTri> tris(m.n_triangles);
        std::vector<DTriAttr> attrs(m.n_triangles);
        for (int t = 0; t < m.n_triangles; t++) {
            const frayhip_triangle& T = m.triangles[t];
            DTri& o = tris[t];
            put3(o.g, T.gnormal);
            put3(o.A, m.vertices + 3 * (size_t)T.v[0]);
            put3(o.N, T.ABcrossAC);
            put3(o.AC, T.AC);
            put3(o.AB, T.AB);
            o.index = t; o.pad = 0;
            DTriAttr& a = attrs[t];
            memset(&a, 0, sizeof a);
            if (M.smooth) {
                put3(a.nA, m.normals + 3 * (size_t)T.n[0]);
                put3(a.nB, m.normals + 3 * (size_t)T.n[1]);
                put3(a.nC, m.normals + 3 * (size_t)T.n[2]);
            }
            if (M.hasUV) {
                const double* tA = m.uvs + 3 * (size_t)T.t[0]; const double* tB = m.uvs + 3 * (size_t)T.t[1]; const double* tC = m.uvs + 3 * (size_t)T.t[2];
                a.tA[0] = tA[0]; a.tA[1] = tA[1]; a.tB[0] = tB[0]; a.tB[1] = tB[1]; a.tC[0] = tC[0]; a.tC[1] = tC[1];
            }
            put3(a.dNdx, T.dNdx);
            put3(a.dNdy, T.dNdy);
        }
        // KD nodes: add each node's own box.  Boxes are derived top-down exactly as BBox::split
        // does (copy parent, overwrite one coordinate).
        std::vector<DKd> kd(m.n_kdnodes);
        std::vector<DKdBox> kdBox(m.n_kdnodes);
        if (m.n_kdnodes > 0) {
            for (int k = 0; k < 3; k++) { kdBox[0].lo[k] = m.bbox_min[k]; kdBox[0].hi[k] = m.bbox_max[k]; }
            for (int n = 0; n < m.n_kdnodes; n++) {   // parents precede children in the array
                const frayhip_kdnode& K = m.kdnodes[n];
                DKd& o = kd[n];
                o.child0 = K.child0; o.meta = K.axis;
                if (K.axis != 3) {
                    o.split = K.split;
                    o.meta |= (m.kdnodes[K.child0].axis == 3 ? 4 : 0) | (m.kdnodes[K.child0 + 1].axis == 3 ? 8 : 0);
                    kdBox[K.child0] = kdBox[n];
                    kdBox[K.child0 + 1] = kdBox[n];
                    kdBox[K.child0].hi[K.axis] = K.split;
                    kdBox[K.child0 + 1].lo[K.axis] = K.split;
                } else {
                    o.triBegin = K.tri_begin; o.triCount = K.tri_count;
                }
            }
        }
        moff[mi].tris = A.add(tris.data(), tris.size() * sizeof(DTri));
        moff[mi].attrs = A.add(attrs.data(), attrs.size() * sizeof(DTriAttr));
        moff[mi].kd = A.add(kd.data(), kd.size() * sizeof(DKd));
        moff[mi].kdBox = A.add(kdBox.data(), kdBox.size() * sizeof(DKdBox));
        moff[mi].refs = A.add(m.trirefs, (size_t)m.n_trirefs * sizeof(int32_t));
        std::vector<DTri> ltris((size_t)m.n_trirefs);
        for (int r = 0; r < m.n_trirefs; r++) ltris[r] = tris[m.trirefs[r]];
        moff[mi].ltris = A.add(ltris.data(), ltris.size() * sizeof(DTri));
        // the same leaf order again as FP32 records of the certified filter (dev_tricert.hpp), relative to the centre of the mesh's box
        for (int k = 0; k < 3; k++) M.ref[k] = (m.bbox_min[k] + m.bbox_max[k]) * 0.5;
        std::vector<DTri32> l32(ltris.size());
        for (size_t r = 0; r < ltris.size(); r++) tricert_make(l32[r], ltris[r].A, ltris[r].AB, ltris[r].AC, ltris[r].N, M.ref);
        moff[mi].ltris32 = A.add(l32.data(), l32.size() * sizeof(DTri32));
    }
    size_t oTexels = A.add(d.texels, (size_t)d.n_texels * sizeof(float));
    std::vector<DTexture> tex(d.n_textures);
    for (int i = 0; i < d.n_textures; i++) {
        const frayhip_texture& t = d.textures[i];
        DTexture& o = tex[i];
        o.kind = t.kind; o.width = t.width; o.height = t.height; o.pad = 0;
        memcpy(o.color1, t.color1, sizeof o.color1);
        memcpy(o.color2, t.color2, sizeof o.color2);
        o.scaling = t.scaling; o.bumpIntensity = t.bumpIntensity; o.ior = t.ior;
        o.texels = nullptr;   // patched below
    }
    std::vector<DShader> shaders(d.n_shaders);
    for (int i = 0; i < d.n_shaders; i++) {
        const frayhip_shader& s = d.shaders[i];
        DShader& o = shaders[i];
        o.kind = s.kind; o.texture = s.texture;
        memcpy(o.color, s.color, sizeof o.color);
        memcpy(o.specularColor, s.specularColor, sizeof o.specularColor);
        memcpy(o.mult, s.mult, sizeof o.mult);
        o.numSamples = s.numSamples;
        o.exponent = s.exponent; o.specularMultiplier = s.specularMultiplier; o.glossiness = s.glossiness;
        o.deflectionScaling = s.deflectionScaling; o.ior = s.ior;
        o.layerBegin = s.layer_begin; o.layerCount = s.layer_count;
        o.usesUV = shader_uses_uv(d, i);
        o.pad = 0;
    }
    size_t oShaders = A.add(shaders.data(), shaders.size() * sizeof(DShader));
    std::vector<DLayer> layers(d.n_layers);
    for (int i = 0; i < d.n_layers; i++) {
        layers[i].shader = d.layers[i].shader; layers[i].texture = d.layers[i].texture;
        memcpy(layers[i].opacity, d.layers[i].opacity, sizeof layers[i].opacity);
        layers[i].pad = 0;
    }
    size_t oLayers = A.add(layers.data(), layers.size() * sizeof(DLayer));
    std::vector<DLight> lights(d.n_lights);
    bool anyLightDraws = false;
    for (int i = 0; i < d.n_lights; i++) {
        const frayhip_light& l = d.lights[i];
        DLight& o = lights[i];
        o.kind = l.kind; o.xSubd = l.xSubd; o.ySubd = l.ySubd; o.pad = 0;
        memcpy(o.color, l.color, sizeof o.color);
        o.power = l.power;
        put3(o.pos, l.pos);
        putX(o.T, l.T);
        put3(o.center, l.center);
        o.area = l.area;
        sc->lightSampleCount += l.kind == FRAYHIP_LIGHT_RECT ? l.xSubd * l.ySubd : 1;
        if (l.kind == FRAYHIP_LIGHT_RECT) sc->lightDraws = true;
        if (l.kind == FRAYHIP_LIGHT_RECT) anyLightDraws = true;          // RectLight::getNthSample draws two words per sample (lights.cpp:62-63)
        o.areaXsize = 1.0 / l.xSubd;
        o.areaYsize = 1.0 / l.ySubd;
    }
    // glossy fans may be drawn ahead (dev_whitted.hpp) where nothing under them is likely to draw: no light that samples, a fan of eight or more
    sc->specFanMax = 0;
    if (!anyLightDraws)
        for (int i = 0; i < d.n_shaders; i++)
            if (d.shaders[i].kind == FRAYHIP_SHADER_REFL && d.shaders[i].glossiness != 1.0 && d.shaders[i].numSamples >= 8)
                sc->specFanMax = std::max(sc->specFanMax, (int)d.shaders[i].numSamples);
    size_t oLights = A.add(lights.data(), lights.size() * sizeof(DLight));
    size_t oMeshes = A.add(nullptr, 0);              // reserve aligned slots for the tables that hold device pointers
    A.host.resize(oMeshes + meshes.size() * sizeof(DMesh));
    size_t oTex = A.add(nullptr, 0);
    A.host.resize(oTex + tex.size() * sizeof(DTexture));

    if (hipMalloc(&sc->d_arena, A.host.size() ? A.host.size() : 256) != hipSuccess) {
        set_error("frayhip_scene_create: hipMalloc failed (no device?)");
        delete sc;
        return FRAYHIP_E_NODEVICE;
    }
    unsigned char* base = (unsigned char*)sc->d_arena;
    for (int mi = 0; mi < d.n_meshes; mi++) {
        meshes[mi].tris = (const FRAY_RO DTri*)(base + moff[mi].tris);
        meshes[mi].attrs = (const FRAY_RO DTriAttr*)(base + moff[mi].attrs);
        meshes[mi].kd = (const FRAY_RO DKd*)(base + moff[mi].kd);
        meshes[mi].kdBox = (const FRAY_RO DKdBox*)(base + moff[mi].kdBox);
        meshes[mi].refs = (const FRAY_RO int32_t*)(base + moff[mi].refs);
        meshes[mi].ltris = (const FRAY_RO DTri*)(base + moff[mi].ltris);
        meshes[mi].ltris32 = (const FRAY_RO DTri32*)(base + moff[mi].ltris32);
    }
    if (!meshes.empty()) memcpy(A.host.data() + oMeshes, meshes.data(), meshes.size() * sizeof(DMesh));
    std::vector<DNodeX> nodesX(nodes.size());
    for (int i = 0; i < d.n_nodes; i++) {
        DNode& N = nodes[i];
        DNodeX& X = nodesX[i];
        N.tlTris = 0; N.tlCulling = 0; N.tlPtr = nullptr; N.boxMax = 0; N.gated = 0; N.padN = 0;
        for (int k = 0; k < 3; k++) N.bmin[k] = N.bmax[k] = X.bminE[k] = X.bmaxE[k] = 0;
        if (N.geomKind == FRAYHIP_GEOM_MESH && !meshes[N.geomIndex].hasKd) {
            const DMesh& M = meshes[N.geomIndex];
            N.tlTris = M.nTris; N.tlCulling = M.culling; N.tlPtr = M.tris;
            put3(N.bmin, M.bmin); put3(N.bmax, M.bmax);
            N.boxMax = M.boxMax;
            for (int k = 0; k < 3; k++) { X.bminE[k] = N.bmin[k] - 1e-6; X.bmaxE[k] = N.bmax[k] + 1e-6; }
        }
    }
    // Bounds of a geometry tree in its own (local) space: every point an intersection of the tree can lie on.  Plus: both operands; Minus: the left one
    // (a ray that has no intersection with the left operand is never inside the difference); And: either operand alone bounds the result, the smaller
    // box is taken.  A Plane operand makes its tree unbounded (ok = false) unless the operator hides it.
    struct GB { bool ok; double lo[3], hi[3]; };
    std::function<GB(int, int)> bounds = [&](int g, int depth) -> GB {
        GB b{true, {0, 0, 0}, {0, 0, 0}};
        if (g < 0 || g >= d.n_geoms || depth > FRAY_CSG_DEPTH + 1) { b.ok = false; return b; }
        const int kind = d.geoms[g].kind, idx = d.geoms[g].index;
        if (kind == FRAYHIP_GEOM_PLANE) {
            // never bounded here: Plane::intersect divides 0 by 0 for a horizontal ray that starts at the plane's height and then reports a hit at NaN
            // (geometry.cpp:35-41: no comparison with NaN is true), wherever the ray is -- no box holds that
            b.ok = false; return b;
        } else if (kind == FRAYHIP_GEOM_SPHERE) {
            for (int k = 0; k < 3; k++) { b.lo[k] = d.spheres[idx].O[k] - std::fabs(d.spheres[idx].R); b.hi[k] = d.spheres[idx].O[k] + std::fabs(d.spheres[idx].R); }
        } else if (kind == FRAYHIP_GEOM_CUBE) {
            for (int k = 0; k < 3; k++) { b.lo[k] = d.cubes[idx].O[k] - std::fabs(d.cubes[idx].halfSide); b.hi[k] = d.cubes[idx].O[k] + std::fabs(d.cubes[idx].halfSide); }
        } else if (kind == FRAYHIP_GEOM_MESH) {
            for (int k = 0; k < 3; k++) { b.lo[k] = d.meshes[idx].bbox_min[k]; b.hi[k] = d.meshes[idx].bbox_max[k]; }
        } else if (kind == FRAYHIP_GEOM_CSG) {
            const frayhip_csg& C = d.csgs[idx];
            const GB L = bounds(C.left, depth + 1), R = bounds(C.right, depth + 1);
            auto vol = [](const GB& q) { return (q.hi[0] - q.lo[0]) * (q.hi[1] - q.lo[1]) * (q.hi[2] - q.lo[2]); };
            if (C.op == FRAYHIP_CSG_MINUS) return L;
            if (C.op == FRAYHIP_CSG_AND) { if (L.ok && R.ok) return vol(L) <= vol(R) ? L : R; return L.ok ? L : R; }
            if (!L.ok || !R.ok) { b.ok = false; return b; }
            for (int k = 0; k < 3; k++) { b.lo[k] = std::min(L.lo[k], R.lo[k]); b.hi[k] = std::max(L.hi[k], R.hi[k]); }
        } else b.ok = false;
        for (int k = 0; k < 3; k++) if (!(std::isfinite(b.lo[k]) && std::isfinite(b.hi[k]) && b.lo[k] <= b.hi[k])) b.ok = false;
        return b;
    };
    std::vector<GB> csgLocal(d.n_nodes, GB{false, {0, 0, 0}, {0, 0, 0}});
    for (int i = 0; i < d.n_nodes; i++) {
        DNodeX& X = nodesX[i];
        for (int k = 0; k < 3; k++) X.cc[k] = X.ch[k] = 0;
        X.cM = 0; X.csgBox = 0; X.padX = 0;
        if (nodes[i].geomKind != FRAYHIP_GEOM_CSG) continue;
        const GB b = bounds(d.nodes[i].geom, 0);
        if (!b.ok) continue;
        csgLocal[i] = b;
        for (int k = 0; k < 3; k++) {
            X.cc[k] = 0.5 * (b.lo[k] + b.hi[k]);
            X.ch[k] = std::max(b.hi[k] - X.cc[k], X.cc[k] - b.lo[k]) * (1.0 + 1e-12) + 1e-5;       // the true extents and dev_misscert.hpp's constant margin
            X.cM = std::max(X.cM, std::fabs(X.cc[k]) + X.ch[k]);
        }
        X.csgBox = X.cM < 1e9 ? 1 : 0;
    }
    if (!nodes.empty()) memcpy(A.host.data() + oNodes, nodes.data(), nodes.size() * sizeof(DNode));
    if (!nodesX.empty()) memcpy(A.host.data() + oNodesX, nodesX.data(), nodesX.size() * sizeof(DNodeX));
    // gates: world-space boxes of the meshes whose brute-force triangle loops are worth skipping for a whole wave (dev_scene.hpp DGate):
    // the eight corners of the mesh's box through the node's transform (Transform::transformPoint, matrix.cpp:137-146), a hair wider
    int nGates = 0;
    bool gatesExact = false;
    {
        DGate gates[FRAY_MAX_GATES];
        int gateNode[FRAY_MAX_GATES];
        // ... and of the CsgOp nodes whose tree is bounded (their machine is the most expensive thing a ray can enter), unless the box is so large
        // against the others that nearly every ray enters it anyway (a floor slab): larger than 30 times the smallest such box in some extent
        double smallest = 1e300;
        for (int i = 0; i < d.n_nodes; i++)
            if (csgLocal[i].ok) for (int k = 0; k < 3; k++) smallest = std::min(smallest, std::max(csgLocal[i].hi[k] - csgLocal[i].lo[k], 1e-9));
        for (int i = 0; i < d.n_nodes && nGates < FRAY_MAX_GATES; i++) {
            const DNode& N = nodes[i];
            double bmin[3], bmax[3];
            if (N.tlTris >= FRAY_GATE_MIN_TRIS) { put3(bmin, N.bmin); put3(bmax, N.bmax); }
            else if (csgLocal[i].ok) {
                bool huge = false;
                for (int k = 0; k < 3; k++) { bmin[k] = csgLocal[i].lo[k]; bmax[k] = csgLocal[i].hi[k]; huge = huge || bmax[k] - bmin[k] > 30.0 * smallest; }
                if (huge) continue;
            } else continue;
            DGate g;
            for (int k = 0; k < 3; k++) { g.lo[k] = 1e300; g.hi[k] = -1e300; }
            for (int c = 0; c < 8; c++) {
                const double p[3] = {c & 1 ? bmax[0] : bmin[0], c & 2 ? bmax[1] : bmin[1], c & 4 ? bmax[2] : bmin[2]};
                for (int k = 0; k < 3; k++) {
                    const double w = p[0] * N.T.m[k] + p[1] * N.T.m[3 + k] + p[2] * N.T.m[6 + k] + N.T.off[k];
                    g.lo[k] = std::min(g.lo[k], w); g.hi[k] = std::max(g.hi[k], w);
                }
            }
            // an untransformed node: the box is the geometry's own, in the space the reference tests it in -- the producers' FP32 certificate applies
            g.exact = N.xfIdentity ? 1 : 0;
            g.Mf = 0;
            for (int k = 0; k < 3; k++) {
                const double c = 0.5 * (bmin[k] + bmax[k]), half = std::max(bmax[k] - c, c - bmin[k]);
                g.cf[k] = (float)c;
                g.hf[k] = std::nextafterf((float)(half * (1.0 + 1e-12) + 1e-5 + std::fabs(c - (double)g.cf[k])), INFINITY);
                g.Mf = std::max(g.Mf, std::nextafterf(std::fabs(g.cf[k]) + g.hf[k], INFINITY));
            }
            if (!(g.Mf < 1e9f)) g.exact = 0;
            for (int k = 0; k < 3; k++) { const double e = 1e-6 * (1.0 + std::fabs(g.lo[k]) + std::fabs(g.hi[k])); g.lo[k] -= e; g.hi[k] += e; }
            gateNode[nGates] = i;
            gates[nGates++] = g;
        }
        gatesExact = nGates > 0;
        for (int q = 0; q < nGates; q++) gatesExact = gatesExact && gates[q].exact;
        if (gatesExact) {
            for (int q = 0; q < nGates; q++) nodes[gateNode[q]].gated = 1;
            memcpy(A.host.data() + oNodes, nodes.data(), nodes.size() * sizeof(DNode));        // (the nodes were copied before the gates were known)
        }
        if (nGates) memcpy(A.host.data() + oGates, gates, (size_t)nGates * sizeof(DGate));
    }
    for (int i = 0; i < d.n_textures; i++) tex[i].texels = (const FRAY_RO float*)(base + oTexels) + d.textures[i].texel_offset;
    if (!tex.empty()) memcpy(A.host.data() + oTex, tex.data(), tex.size() * sizeof(DTexture));
    hipError_t e = hipMemcpy(sc->d_arena, A.host.data(), A.host.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) { set_error(std::string("frayhip_scene_create: upload failed: ") + hipGetErrorString(e)); (void)hipFree(sc->d_arena); delete sc; return FRAYHIP_E_NODEVICE; }
    sc->arena_bytes = A.host.size();

    DScene& S = sc->S;
    S.nodes = (const FRAY_RO DNode*)(base + oNodes);
    S.nodesX = (const FRAY_RO DNodeX*)(base + oNodesX);
    S.gates = (const FRAY_RO DGate*)(base + oGates);
    S.nGates = nGates; S.gatesExact = gatesExact ? 1 : 0;
    S.planes = (const FRAY_RO DPlane*)(base + oPlanes);
    S.spheres = (const FRAY_RO DSphere*)(base + oSpheres);
    S.cubes = (const FRAY_RO DCube*)(base + oCubes);
    S.csgs = (const FRAY_RO DCsg*)(base + oCsgs);
    S.meshes = (const FRAY_RO DMesh*)(base + oMeshes);
    S.shaders = (const FRAY_RO DShader*)(base + oShaders);
    S.layers = (const FRAY_RO DLayer*)(base + oLayers);
    S.textures = (const FRAY_RO DTexture*)(base + oTex);
    S.lights = (const FRAY_RO DLight*)(base + oLights);
    S.env.present = d.environment.present;
    if (d.n_textures > 0 || (d.environment.present && d.environment.loaded)) sc->textured = true;
    S.env.loaded = d.environment.loaded;
    for (int f = 0; f < 6; f++) {
        S.env.width[f] = d.environment.width[f];
        S.env.height[f] = d.environment.height[f];
        S.env.face[f] = (const FRAY_RO float*)(base + oTexels) + d.environment.texel_offset[f];
    }
    S.nNodes = d.n_nodes;
    S.nLights = d.n_lights;
    S.probPickLight = d.n_lights > 0 ? 1.0f / (float)d.n_lights : 0.0f;
    sc->camera = d.camera;
    sc->settings = d.settings;
    // [0] everything but k_pt_shadow, [1] k_pt_shadow
    if (hipMalloc((void**)&sc->d_stats, kStatsBytes) != hipSuccess || hipMalloc((void**)&sc->d_qmeta, 3 * FRAY_PT_LANES * sizeof(QMeta)) != hipSuccess ||
        hipEventCreate(&sc->evA) != hipSuccess || hipEventCreate(&sc->evB) != hipSuccess || !create_lanes(sc)) {
        set_error("frayhip_scene_create: could not allocate the per-scene device state");
        frayhip_scene_destroy(sc);
        return FRAYHIP_E_NOMEM;
    }
    // profiling aids: the same knobs as frayhip_scene_set_option, preset from the environment
    if (const char* e = getenv("FRAYHIP_PT_LANES")) { long v = atol(e); if (v >= 1 && v <= FRAY_PT_LANES) sc->ptLanes = (int)v; }
    if (const char* e = getenv("FRAYHIP_SPECULATE_FANS")) sc->speculateFans = atol(e) != 0;
    if (const char* e = getenv("FRAYHIP_FP_CONTRACT")) sc->fpContract = atol(e) == 1;
    if (const char* e = getenv("FRAYHIP_FUSED_WHITTED_MAX")) { long v = atol(e); if (v >= 0 && v <= 1024) sc->fusedWhittedMax = (int)v; }
    if (const char* e = getenv("FRAYHIP_CSG_LANES")) { long v = atol(e); if (v >= 1 && v <= FRAY_PT_LANES) sc->csgLanes = (int)v; }
    if (const char* e = getenv("FRAYHIP_PT_BUDGET_MIB")) { long v = atol(e); if (v >= 1 && v <= (1 << 20)) { sc->ptBudgetBytes = (size_t)v << 20; sc->ptBudgetEff = 0; } }
    *out = sc;
    return FRAYHIP_OK;
}

int frayhip_scene_get_option(frayhip_scene* s, const char* name, int64_t* value)
{
    if (!s || !name || !value) { set_error("frayhip_scene_get_option: null argument"); return FRAYHIP_E_ARG; }
    const std::string n(name);
    if (n == "pt_lanes") *value = s->ptLanes;
    else if (n == "pt_budget_mib") *value = (int64_t)(s->ptBudgetBytes >> 20);
    else if (n == "speculate_fans") *value = s->speculateFans ? 1 : 0;
    else if (n == "fp_contract") *value = s->fpContract ? 1 : 0;
    else if (n == "contracted_launches") *value = s->lastContracted;
    else if (n == "whitted_path") *value = s->lastWhittedPath;
    else if (n == "fused_whitted_max") *value = s->fusedWhittedMax;
    else if (n == "pt_budget_effective_mib") *value = (int64_t)(frayhip_detail::work_budget(s) >> 20);
    else if (n == "fans_filed") *value = s->lastFans[0];
    else if (n == "fan_children") *value = s->lastFans[1];
    else if (n == "fan_children_looked_up") *value = s->lastFans[2];
    else if (n == "fans_given_up") *value = s->lastFans[3];
    else { set_error("frayhip_scene_get_option: unknown option " + n); return FRAYHIP_E_ARG; }
    return FRAYHIP_OK;
}

int frayhip_scene_set_option(frayhip_scene* s, const char* name, int64_t value)
{
    if (!s || !name) { set_error("frayhip_scene_set_option: null argument"); return FRAYHIP_E_ARG; }
    const std::string n(name);
    if (n == "pt_lanes") {
        if (value < 1 || value > FRAY_PT_LANES) { set_error("frayhip_scene_set_option: pt_lanes must be 1.." + std::to_string(FRAY_PT_LANES)); return FRAYHIP_E_ARG; }
        s->ptLanes = (int)value;
    } else if (n == "pt_budget_mib") {
        if (value < 1 || value > (1 << 20)) { set_error("frayhip_scene_set_option: pt_budget_mib must be 1..1048576"); return FRAYHIP_E_ARG; }
        s->ptBudgetBytes = (size_t)value << 20;
        s->ptBudgetEff = 0;                 // clamp again at the next frame
    } else if (n == "speculate_fans") {
        if (value != 0 && value != 1) { set_error("frayhip_scene_set_option: speculate_fans must be 0 or 1"); return FRAYHIP_E_ARG; }
        s->speculateFans = value != 0;
    } else if (n == "fused_whitted_max") {
        if (value < 0 || value > 1024) { set_error("frayhip_scene_set_option: fused_whitted_max must be 0..1024"); return FRAYHIP_E_ARG; }
        s->fusedWhittedMax = (int)value;
    } else if (n == "fp_contract") {
        if (value != 0 && value != 1) { set_error("frayhip_scene_set_option: fp_contract must be 0 or 1"); return FRAYHIP_E_ARG; }
        s->fpContract = value != 0;
    } else {
        set_error("frayhip_scene_set_option: unknown option " + n);
        return FRAYHIP_E_ARG;
    }
    return FRAYHIP_OK;
}

int frayhip_scene_set_view(frayhip_scene* s, const frayhip_camera* camera, const frayhip_settings* settings)
{
    if (!s) { set_error("frayhip_scene_set_view: null scene"); return FRAYHIP_E_ARG; }
    if (settings && (settings->frameWidth <= 0 || settings->frameHeight <= 0)) { set_error("frayhip_scene_set_view: bad frame size"); return FRAYHIP_E_ARG; }
    if (camera) s->camera = *camera;
    if (settings) s->settings = *settings;
    return FRAYHIP_OK;
}

void frayhip_scene_destroy(frayhip_scene* s)
{
    if (!s) return;
    if (s->d_arena) (void)hipFree(s->d_arena);
    if (s->d_work) (void)hipFree(s->d_work);
    if (s->d_stats) (void)hipFree(s->d_stats);
    if (s->d_qmeta) (void)hipFree(s->d_qmeta);
    if (s->evA) (void)hipEventDestroy(s->evA);
    if (s->evB) (void)hipEventDestroy(s->evB);
    if (s->evLaneStart) (void)hipEventDestroy(s->evLaneStart);
    for (int k = 0; k < FRAY_PT_LANES; k++) {
        if (s->evResolved[k]) (void)hipEventDestroy(s->evResolved[k]);
        if (s->laneStream[k]) (void)hipStreamDestroy(s->laneStream[k]);
    }
    for (auto e : s->evPool) (void)hipEventDestroy(e);
    for (auto e : s->evPoolShadow) (void)hipEventDestroy(e);
    delete s;
}


int frayhip_render_device(frayhip_scene* s, const frayhip_frame* f, float* d_rgb, int32_t* d_hit_id, double* d_hit_dist, void* hip_stream,
                          frayhip_stats* st)
{
    if (!s || !f) { set_error("frayhip_render_device: null argument"); return FRAYHIP_E_ARG; }
    hipStream_t stream = (hipStream_t)hip_stream;
    const bool stats = (f->flags & FRAYHIP_FRAME_STATS) != 0;
    if (s->extGeometry) return stats ? frayhip_detail::render_impl<3>(s, f, d_rgb, d_hit_id, d_hit_dist, stream, st) : frayhip_detail::render_impl<2>(s, f, d_rgb, d_hit_id, d_hit_dist, stream, st);
    if (s->kdMeshes) return stats ? frayhip_detail::render_impl<5>(s, f, d_rgb, d_hit_id, d_hit_dist, stream, st) : frayhip_detail::render_impl<4>(s, f, d_rgb, d_hit_id, d_hit_dist, stream, st);
    if (s->textured) return stats ? frayhip_detail::render_impl<9>(s, f, d_rgb, d_hit_id, d_hit_dist, stream, st) : frayhip_detail::render_impl<8>(s, f, d_rgb, d_hit_id, d_hit_dist, stream, st);
    return stats ? frayhip_detail::render_impl<1>(s, f, d_rgb, d_hit_id, d_hit_dist, stream, st) : frayhip_detail::render_impl<0>(s, f, d_rgb, d_hit_id, d_hit_dist, stream, st);
}

int frayhip_render(frayhip_scene* s, const frayhip_frame* f, float* rgb, int32_t* hit_id, double* hit_dist, frayhip_stats* st)
{
    if (!s || !f) { set_error("frayhip_render: null argument"); return FRAYHIP_E_ARG; }
    const size_t n = (size_t)s->settings.frameWidth * s->settings.frameHeight;
    float* d_rgb = nullptr; int32_t* d_id = nullptr; double* d_dist = nullptr;
    int rc = FRAYHIP_OK;
    auto cleanup = [&]() { if (d_rgb) (void)hipFree(d_rgb); if (d_id) (void)hipFree(d_id); if (d_dist) (void)hipFree(d_dist); };
    if (rgb && hipMalloc((void**)&d_rgb, n * 12) != hipSuccess) rc = FRAYHIP_E_NOMEM;
    if (!rc && hit_id && hipMalloc((void**)&d_id, n * 4) != hipSuccess) rc = FRAYHIP_E_NOMEM;
    if (!rc && hit_dist && hipMalloc((void**)&d_dist, n * 8) != hipSuccess) rc = FRAYHIP_E_NOMEM;
    if (rc) { set_error("frayhip_render: hipMalloc failed"); cleanup(); return rc; }
    // pixels outside this call's buckets keep what the caller had in the buffers: only a call that renders a SUBSET of the
    // buckets needs the caller's frame on the device first; a whole-frame call overwrites every pixel
    hipError_t ce = hipSuccess;
    const bool subset = f->bucket_stride > 1 || f->bucket_first != 0;
    if (subset) {
        if (d_rgb && ce == hipSuccess) ce = hipMemcpy(d_rgb, rgb, n * 12, hipMemcpyHostToDevice);
        if (d_id && ce == hipSuccess) ce = hipMemcpy(d_id, hit_id, n * 4, hipMemcpyHostToDevice);
        if (d_dist && ce == hipSuccess) ce = hipMemcpy(d_dist, hit_dist, n * 8, hipMemcpyHostToDevice);
    }
    if (ce != hipSuccess) { set_error(std::string("frayhip_render: host-to-device copy failed: ") + hipGetErrorString(ce)); cleanup(); return frayhip_detail::hip_error_code(ce); }
    rc = frayhip_render_device(s, f, d_rgb, d_id, d_dist, nullptr, st);
    if (!rc) {
        if (d_rgb && ce == hipSuccess) ce = hipMemcpy(rgb, d_rgb, n * 12, hipMemcpyDeviceToHost);
        if (d_id && ce == hipSuccess) ce = hipMemcpy(hit_id, d_id, n * 4, hipMemcpyDeviceToHost);
        if (d_dist && ce == hipSuccess) ce = hipMemcpy(hit_dist, d_dist, n * 8, hipMemcpyDeviceToHost);
        if (ce != hipSuccess) { set_error(std::string("frayhip_render: device-to-host copy failed: ") + hipGetErrorString(ce)); rc = frayhip_detail::hip_error_code(ce); }
    }
    cleanup();
    return rc;
}

__global__ void k_debug_rng(uint32_t seed, int n, float* f, double* d, int32_t* it, int hi, uint32_t* work)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    MtLong g[3];
    for (int k = 0; k < 3; k++) { g[k].st = work + k; g[k].stride = 3; g[k].reseed(seed); }
    for (int i = 0; i < n; i++) {
        f[i] = rng_float(g[0]);
        d[i] = rng_double(g[1]);
        it[i] = rng_int0(g[2], hi);
    }
}

int frayhip_debug_rng(uint32_t seed, int n, float* floats, double* doubles, int32_t* ints, int int_hi)
{
    if (n < 0 || n > 4096 || int_hi < 0) { set_error("frayhip_debug_rng: bad argument"); return FRAYHIP_E_ARG; }
    // one allocation: floats | doubles | ints | three 624-word generator states
    unsigned char* buf = nullptr;
    const size_t oF = 0, oD = 4096 * 4, oI = oD + 4096 * 8, oW = oI + 4096 * 4, total = oW + 3 * 624 * 4;
    HIP_TRY(hipMalloc((void**)&buf, total));
    hipLaunchKernelGGL(k_debug_rng, dim3(1), dim3(64), 0, nullptr, seed, n, (float*)(buf + oF), (double*)(buf + oD), (int32_t*)(buf + oI), int_hi, (uint32_t*)(buf + oW));
    hipError_t e = hipDeviceSynchronize();
    if (e == hipSuccess && floats) e = hipMemcpy(floats, buf + oF, (size_t)n * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess && doubles) e = hipMemcpy(doubles, buf + oD, (size_t)n * 8, hipMemcpyDeviceToHost);
    if (e == hipSuccess && ints) e = hipMemcpy(ints, buf + oI, (size_t)n * 4, hipMemcpyDeviceToHost);
    (void)hipFree(buf);
    if (e != hipSuccess) { set_error(std::string("frayhip_debug_rng: ") + hipGetErrorString(e)); return frayhip_detail::hip_error_code(e); }
    return FRAYHIP_OK;
}

__global__ void k_debug_libm(int n, const double* x, double* out)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        double sn, cs;
        fray_sincos(x[i], &sn, &cs);
        out[i] = sn;
        out[n + i] = cs;
        double a = x[i];
        a = a - 2.0 * floor(a * 0.5) - 1.0;                 // folded into [-1, 1) for acos
        out[2 * n + i] = fray_acos(a);
        out[3 * n + i] = a;
    }
}

int frayhip_debug_libm(int n, const double* x, double* sin_out, double* cos_out, double* acos_out, double* acos_arg)
{
    if (n <= 0 || n > (1 << 22) || !x) { set_error("frayhip_debug_libm: bad argument"); return FRAYHIP_E_ARG; }
    double *dx = nullptr, *dout = nullptr;
    HIP_TRY(hipMalloc((void**)&dx, (size_t)n * 8));
    if (hipMalloc((void**)&dout, (size_t)n * 32) != hipSuccess) { (void)hipFree(dx); set_error("frayhip_debug_libm: out of device memory"); return FRAYHIP_E_NOMEM; }
    int rc = FRAYHIP_OK;
    if (hipMemcpy(dx, x, (size_t)n * 8, hipMemcpyHostToDevice) != hipSuccess) rc = FRAYHIP_E_NODEVICE;
    if (rc == FRAYHIP_OK) {
        hipLaunchKernelGGL(k_debug_libm, dim3(grid_for(n)), dim3(256), 0, nullptr, n, dx, dout);
        if (hipDeviceSynchronize() != hipSuccess) rc = FRAYHIP_E_NODEVICE;
    }
    double* outs[4] = {sin_out, cos_out, acos_out, acos_arg};
    for (int k = 0; k < 4 && rc == FRAYHIP_OK; k++)
        if (outs[k] && hipMemcpy(outs[k], dout + (size_t)k * n, (size_t)n * 8, hipMemcpyDeviceToHost) != hipSuccess) rc = FRAYHIP_E_NODEVICE;
    (void)hipFree(dx); (void)hipFree(dout);
    if (rc != FRAYHIP_OK) set_error("frayhip_debug_libm: device error");
    return rc;
}

static int pack_impl(const float* d_frame, float* d_packed, int width, int height, int channels, int first, int stride, void* hip_stream, int unpack)
{
    if (!d_frame || !d_packed || channels < 1) { set_error("frayhip_pack_buckets_device: bad argument"); return FRAYHIP_E_ARG; }
    int nb = frayhip_bucket_count(width, height, first, stride);
    if (nb < 0) { set_error("frayhip_pack_buckets_device: bad bucket range"); return FRAYHIP_E_ARG; }
    DFrame F{};
    F.W = width; F.H = height; F.BW = (width - 1) / 48 + 1; F.BH = (height - 1) / 48 + 1;
    F.bucketFirst = first; F.bucketStride = stride; F.nBuckets = nb;
    int nItems = nb * 2304;
    if (nItems > 0) hipLaunchKernelGGL(k_pack, dim3(grid_for(nItems)), dim3(256), 0, (hipStream_t)hip_stream, F, nItems, channels, (float*)d_frame, d_packed, unpack);
    HIP_TRY(hipGetLastError());
    return FRAYHIP_OK;
}
int frayhip_pack_buckets_device(const float* d_frame, float* d_packed, int width, int height, int channels, int bucket_first, int bucket_stride, void* hip_stream)
{
    return pack_impl(d_frame, d_packed, width, height, channels, bucket_first, bucket_stride, hip_stream, 0);
}
int frayhip_unpack_buckets_device(const float* d_packed, float* d_frame, int width, int height, int channels, int bucket_first, int bucket_stride, void* hip_stream)
{
    return pack_impl(d_frame, (float*)d_packed, width, height, channels, bucket_first, bucket_stride, hip_stream, 1);
}

}  // extern "C"

