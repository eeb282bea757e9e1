// fray_oracle -- CPU restatement of fray's per-pixel ray-trace hot path.
//
// TEST INFRASTRUCTURE ONLY.  This library is the checker for the HIP renderer: only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  Nothing under fray_amd/
// links, imports or calls it, and the product has no CPU fallback.
//
// It restates the reference algorithm (anrieff/fray, src/*.cpp; every function cites the lines it
// follows) over the flattened frayhip_scene_desc, in scalar C++ with the reference's evaluation
// order and its FP64-geometry / FP32-colour mix, compiled with -ffp-contract=off.
// Third-party arithmetic the reference relies on is used directly, not restated: libstdc++
// <random> (std::mt19937 + its distributions; GCC 11 here) and glibc libm.
//
// Pinning: primary-ray hit records are pinned by the FNV-1a hashes the survey measured on the
// unmodified reference (tests/test_oracle_golden.py); geometry, lights, camera
// and shader evaluation are additionally cross-checked against the reference's own object code
// where it compiles here (oracle/_ref, see oracle/Makefile and DESIGN.md).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

#include "frayhip.h"

namespace {

const double PI = 3.141592653589793238;   // constants.h:31
const double INF = 1e99;                  // constants.h:32

// ---------------------------------------------------------------- vector.h / color.h
struct Vec {
    double x, y, z;
    double operator[](int i) const { return i == 0 ? x : i == 1 ? y : z; }
    double& at(int i) { return i == 0 ? x : i == 1 ? y : z; }
};
inline Vec vec(const double* p) { return Vec{p[0], p[1], p[2]}; }
inline Vec operator+(Vec a, Vec b) { return Vec{a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vec operator-(Vec a, Vec b) { return Vec{a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec operator-(Vec a) { return Vec{-a.x, -a.y, -a.z}; }
inline Vec operator*(Vec a, double m) { return Vec{a.x * m, a.y * m, a.z * m}; }
inline Vec operator*(double m, Vec a) { return Vec{a.x * m, a.y * m, a.z * m}; }
inline double dot(Vec a, Vec b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline Vec cross(Vec a, Vec b) { return Vec{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline double lengthSqr(Vec a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
inline double length(Vec a) { return sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }
inline Vec normalized(Vec a) { double m = 1.0 / length(a); return a * m; }          // vector.h:81-85
inline double dist(Vec a, Vec b) { return length(a - b); }                          // vector.h:162-166
inline Vec faceforward(Vec d, Vec n) { return dot(d, n) < 0 ? n : -n; }              // vector.h:169-175
inline Vec reflect(Vec i, Vec n) { return i + 2 * dot(-i, n) * n; }                  // vector.h:178-181
inline Vec refract(Vec i, Vec n, double ior)                                         // vector.h:184-191
{
    double NdotI = dot(i, n);
    double k = 1 - (ior * ior) * (1 - NdotI * NdotI);
    if (k < 0.0) return Vec{0, 0, 0};
    return normalized(ior * i - (ior * NdotI + sqrt(k)) * n);
}
inline void orthonormalSystem(Vec a, Vec& b, Vec& c)                                 // vector.h:197-213
{
    Vec t = Vec{1, 0, 0};
    if (fabs(dot(t, a)) > 0.9) t = Vec{0, 1, 0};
    b = normalized(cross(a, t));
    c = cross(a, b);
}

struct Col {
    float r, g, b;
    float intensity() const { return (r + g + b) / 3; }                              // color.h:79-82
};
inline Col col(const float* p) { return Col{p[0], p[1], p[2]}; }
inline Col operator+(Col a, Col b) { return Col{a.r + b.r, a.g + b.g, a.b + b.b}; }
inline Col operator-(Col a, Col b) { return Col{a.r - b.r, a.g - b.g, a.b - b.b}; }
inline Col operator*(Col a, Col b) { return Col{a.r * b.r, a.g * b.g, a.b * b.b}; }
inline Col operator*(Col a, float m) { return Col{a.r * m, a.g * m, a.b * m}; }
inline Col operator/(Col a, float d) { return Col{a.r / d, a.g / d, a.b / d}; }
const Col BLACK{0, 0, 0};

enum { RF_DIFFUSE = 2 };                                                             // vector.h:215-219
struct Ray { Vec start, dir; int depth = 0; unsigned flags = 0; };

// IntersectionInfo, geometry.h:33-39 (geom pointer -> geoms[] index)
struct Hit {
    double dist;
    Vec ip, norm, dNdx, dNdy;
    double u, v;
    int geom;
};

// ---------------------------------------------------------------- matrix.h:36-45, matrix.cpp:143-161
inline Vec mulM(Vec v, const double* m)
{
    return Vec{v.x * m[0] + v.y * m[3] + v.z * m[6], v.x * m[1] + v.y * m[4] + v.z * m[7],
               v.x * m[2] + v.y * m[5] + v.z * m[8]};
}
inline Vec transformPoint(const frayhip_transform& T, Vec p) { return mulM(p, T.m) + vec(T.offset); }
inline Vec untransformPoint(const frayhip_transform& T, Vec p) { return mulM(p - vec(T.offset), T.invM); }
inline Vec transformDir(const frayhip_transform& T, Vec d) { return normalized(mulM(d, T.m)); }
inline Vec untransformDir(const frayhip_transform& T, Vec d) { return normalized(mulM(d, T.invM)); }

// ---------------------------------------------------------------- random_generator.cpp:41-80
struct Rng {
    std::mt19937 gen;
    void seed(unsigned s) { gen.seed(s); }
    int randint(int a, int b) { std::uniform_int_distribution<int> d(a, b); return d(gen); }
    float randfloat() { std::uniform_real_distribution<float> d; return d(gen); }
    double randdouble() { std::uniform_real_distribution<double> d; return d(gen); }
    void unitDiscSample(double& x, double& y)
    {
        double angle = randdouble() * 2 * PI;
        double rad = sqrt(randdouble());
        x = sin(angle) * rad;
        y = cos(angle) * rad;
    }
};

// RNG contract: both generators the reference consults -- the worker's local copy
// `rnd` (main.cpp:333) and the per-thread table entry (getRandomGen()) -- are re-seeded to
// sample_seed(seed, pixel, sample) before each camera sample.
inline uint32_t fmix32(uint32_t h)
{
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return h;
}
inline uint32_t sample_seed(uint32_t seed, uint32_t pixel, uint32_t sample)
{
    uint32_t h = fmix32(seed ^ (pixel * 0x9e3779b1u));
    return fmix32(h ^ (sample * 0x85ebca77u) ^ 0x27d4eb2fu);
}

struct Stats {
    uint64_t closest = 0, shadow = 0, node = 0, kdInner = 0, leafRefs = 0, tri = 0, prim = 0, smooth = 0, samples = 0, tex = 0;
};

// Camera::beginFrame, camera.cpp:34-57
struct CameraFrame {
    Vec topLeft, topRight, bottomLeft, frontDir, upDir, rightDir, pos;
    double w, h, apertureSize, focalPlaneDist, stereoSeparation;
    bool dof;
};

void matmul3(const double* a, const double* b, double* c)   // matrix.cpp:64-73
{
    for (int i = 0; i < 9; i++) c[i] = 0.0;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            for (int k = 0; k < 3; k++) c[i * 3 + j] += a[i * 3 + k] * b[k * 3 + j];
}
CameraFrame cameraBeginFrame(const frayhip_camera& c, int W, int H)
{
    auto rad = [](double a) { return a / 180.0 * PI; };
    CameraFrame f;
    double aspect = c.aspectRatio;
    Vec BC = Vec{-aspect, 1, 1} - Vec{0, 0, 1};
    double m = tan(rad(c.fov / 2)) / length(BC);
    f.topLeft = Vec{-aspect * m, +m, 1};
    f.topRight = Vec{+aspect * m, +m, 1};
    f.bottomLeft = Vec{-aspect * m, -m, 1};
    f.w = W; f.h = H;
    double S, C;
    // sincos() by name: what g++ -O2 makes of the reference's `sin(angle)` / `cos(angle)` pairs (oracle/_ref imports only sincos); its sine differs
    // from sin()'s in the last place for one angle in 700
    sincos(rad(c.roll), &S, &C);
    double rz[9] = {C, -S, 0, S, C, 0, 0, 0, 1};     // rotationAroundZ, matrix.cpp:53-62
    sincos(rad(c.pitch), &S, &C);
    double rx[9] = {1, 0, 0, 0, C, -S, 0, S, C};     // rotationAroundX, matrix.cpp:29-38
    sincos(rad(c.yaw), &S, &C);
    double ry[9] = {C, 0, S, 0, 1, 0, -S, 0, C};     // rotationAroundY, matrix.cpp:41-50
    double t[9], rot[9];
    matmul3(rz, rx, t);
    matmul3(t, ry, rot);
    f.topLeft = mulM(f.topLeft, rot);
    f.topRight = mulM(f.topRight, rot);
    f.bottomLeft = mulM(f.bottomLeft, rot);
    f.frontDir = mulM(Vec{0, 0, 1}, rot);
    f.upDir = mulM(Vec{0, 1, 0}, rot);
    f.rightDir = mulM(Vec{1, 0, 0}, rot);
    f.apertureSize = 1.0 / c.fNumber;
    f.pos = vec(c.pos);
    f.focalPlaneDist = c.focalPlaneDist;
    f.stereoSeparation = c.stereoSeparation;
    f.dof = c.dof != 0;
    return f;
}

struct Tracer {
    const frayhip_scene_desc& S;
    CameraFrame cam;
    Rng rnd;      // the worker's local generator (main.cpp:333)
    Rng table;    // getRandomGen() (random_generator.cpp:128-131)
    Stats st;

    Tracer(const frayhip_scene_desc& s, const CameraFrame& c) : S(s), cam(c) {}

    // ---------------------------------------------------------------- bbox.h
    struct RRay { Vec start, dir, rdir; };
    static bool boxInside(const double* lo, const double* hi, Vec v)   // bbox.h:79-84
    {
        return lo[0] - 1e-6 <= v.x && v.x <= hi[0] + 1e-6 && lo[1] - 1e-6 <= v.y && v.y <= hi[1] + 1e-6 &&
               lo[2] - 1e-6 <= v.z && v.z <= hi[2] + 1e-6;
    }
    static bool boxTest(const double* lo, const double* hi, const RRay& ray)   // bbox.h:87-134
    {
        if (boxInside(lo, hi, ray.start)) return true;
        for (int dim = 0; dim < 3; dim++) {
            if ((ray.dir[dim] < 0 && ray.start[dim] < lo[dim]) || (ray.dir[dim] > 0 && ray.start[dim] > hi[dim])) return false;
            if (fabs(ray.dir[dim]) < 1e-9) continue;
            double mul = ray.rdir[dim];
            int u = (dim == 0) ? 1 : 0;
            int v = (dim == 2) ? 1 : 2;
            double d, x, y;
            d = (lo[dim] - ray.start[dim]) * mul;
            if (d < 0) continue;
            x = ray.start[u] + ray.dir[u] * d;
            if (lo[u] <= x && x <= hi[u]) {
                y = ray.start[v] + ray.dir[v] * d;
                if (lo[v] <= y && y <= hi[v]) return true;
            }
            d = (hi[dim] - ray.start[dim]) * mul;
            if (d < 0) continue;
            x = ray.start[u] + ray.dir[u] * d;
            if (lo[u] <= x && x <= hi[u]) {
                y = ray.start[v] + ray.dir[v] * d;
                if (lo[v] <= y && y <= hi[v]) return true;
            }
        }
        return false;
    }

    // ---------------------------------------------------------------- mesh.cpp:102-141, triangle.cpp:66-94
    bool meshTriangle(const frayhip_mesh& M, int geomId, const RRay& ray, const frayhip_triangle& T, Hit& info)
    {
        st.tri++;
        if (M.backfaceCulling && dot(ray.dir, vec(T.gnormal)) > 0) return false;
        Vec A = vec(M.vertices + 3 * (size_t)T.v[0]);
        Vec N = vec(T.ABcrossAC), AB = vec(T.AB), AC = vec(T.AC);
        Vec D = -ray.dir;
        double Dcr = dot(N, D);
        if (fabs(Dcr) < 1e-12) return false;
        double rDcr = 1 / Dcr;
        Vec H = ray.start - A;
        double gamma = dot(N, H) * rDcr;
        if (gamma < 0 || gamma > info.dist) return false;
        double l2 = dot(cross(H, AC), D) * rDcr;
        if (l2 < 0 || l2 > 1) return false;
        double l3 = dot(cross(AB, H), D) * rDcr;
        if (l3 < 0 || l3 > 1) return false;
        double l1 = 1 - (l2 + l3);
        if (l1 < 0) return false;
        info.dist = gamma;
        info.geom = geomId;
        info.ip = ray.start + ray.dir * info.dist;
        if (M.faceted || M.n_normals == 0) {
            info.norm = vec(T.gnormal);
        } else {
            Vec nA = vec(M.normals + 3 * (size_t)T.n[0]), nB = vec(M.normals + 3 * (size_t)T.n[1]), nC = vec(M.normals + 3 * (size_t)T.n[2]);
            info.norm = normalized(nA + (nB - nA) * l2 + (nC - nA) * l3);
        }
        if (M.n_uvs == 0) {
            info.u = info.v = 0;
        } else {
            Vec tA = vec(M.uvs + 3 * (size_t)T.t[0]), tB = vec(M.uvs + 3 * (size_t)T.t[1]), tC = vec(M.uvs + 3 * (size_t)T.t[2]);
            Vec tc = tA + (tB - tA) * l2 + (tC - tA) * l3;
            info.u = tc.x;
            info.v = tc.y;
        }
        info.dNdx = vec(T.dNdx);
        info.dNdy = vec(T.dNdy);
        return true;
    }

    // Mesh::intersectKD, mesh.cpp:357-394 (recursive, child boxes derived by BBox::split)
    bool meshKD(const frayhip_mesh& M, int geomId, const RRay& ray, Hit& info, int nodeIdx, const double* lo, const double* hi)
    {
        const frayhip_kdnode& node = M.kdnodes[nodeIdx];
        if (node.axis == 3) {
            bool found = false;
            for (int k = 0; k < node.tri_count; k++) {
                st.leafRefs++;
                int idx = M.trirefs[node.tri_begin + k];
                if (meshTriangle(M, geomId, ray, M.triangles[idx], info)) found = true;
            }
            return found && boxInside(lo, hi, info.ip);
        }
        st.kdInner++;
        double clo[2][3], chi[2][3];
        for (int c = 0; c < 2; c++)
            for (int k = 0; k < 3; k++) { clo[c][k] = lo[k]; chi[c][k] = hi[k]; }
        chi[0][node.axis] = node.split;
        clo[1][node.axis] = node.split;
        int order[2];
        if (ray.start[node.axis] < node.split) { order[0] = 0; order[1] = 1; }
        else { order[0] = 1; order[1] = 0; }
        for (int c : order) {
            if (boxTest(clo[c], chi[c], ray)) {
                if (meshKD(M, geomId, ray, info, node.child0 + c, clo[c], chi[c])) return true;
            }
        }
        return false;
    }

    // Mesh::intersect, mesh.cpp:144-165
    bool meshIntersect(const frayhip_mesh& M, int geomId, const Ray& r, Hit& info)
    {
        RRay ray{r.start, r.dir, Vec{0, 0, 0}};
        ray.rdir.x = fabs(ray.dir.x) > 1e-12 ? 1.0 / ray.dir.x : 1e12;   // bbox.h:49-54
        ray.rdir.y = fabs(ray.dir.y) > 1e-12 ? 1.0 / ray.dir.y : 1e12;
        ray.rdir.z = fabs(ray.dir.z) > 1e-12 ? 1.0 / ray.dir.z : 1e12;
        if (!boxTest(M.bbox_min, M.bbox_max, ray)) return false;
        info.dist = INF;
        bool found = false;
        if (M.has_kd) {
            found = meshKD(M, geomId, ray, info, 0, M.bbox_min, M.bbox_max);
        } else {
            for (int i = 0; i < M.n_triangles; i++)
                if (meshTriangle(M, geomId, ray, M.triangles[i], info)) found = true;
        }
        return found;
    }

    // ---------------------------------------------------------------- geometry.cpp
    bool planeIntersect(const frayhip_plane& P, int geomId, const Ray& ray, Hit& info)   // :30-50
    {
        st.prim++;
        if (ray.start.y > P.height && ray.dir.y >= 0) return false;
        if (ray.start.y < P.height && ray.dir.y <= 0) return false;
        double travelByY = fabs(ray.start.y - P.height);
        double unitTravel = fabs(ray.dir.y);
        double scaling = travelByY / unitTravel;
        Vec ip = ray.start + ray.dir * scaling;
        if (fabs(ip.x) > P.limit) return false;
        if (fabs(ip.z) > P.limit) return false;
        info.ip = ip;
        info.dist = dist(ray.start, info.ip);
        info.norm = Vec{0, 1, 0};
        info.u = info.ip.x;
        info.v = info.ip.z;
        info.geom = geomId;
        return true;
    }
    bool sphereIntersect(const frayhip_sphere& Sp, int geomId, const Ray& ray, Hit& info)   // :52-83
    {
        st.prim++;
        Vec O = vec(Sp.O);
        Vec H = ray.start - O;
        double A = 1;
        double B = 2 * dot(ray.dir, H);
        double C = lengthSqr(H) - Sp.R * Sp.R;
        double Disc = B * B - 4 * A * C;
        if (Disc < 0) return false;
        double sqrtDisc = sqrt(Disc);
        double p1 = (-B + sqrtDisc) / (2 * A);
        double p2 = (-B - sqrtDisc) / (2 * A);
        double smaller = std::min(p1, p2);
        double larger = std::max(p1, p2);
        if (larger < 0) return false;
        double d = (smaller >= 0) ? smaller : larger;
        info.ip = ray.start + ray.dir * d;
        info.dist = dist(ray.start, info.ip);
        info.norm = normalized(info.ip - O);
        info.u = ((atan2(info.norm.z, info.norm.x) / PI * 180.0) + 180.0) / 360.0;
        info.v = 1 - ((asin(info.norm.y) / PI * 180.0) + 90) / 180.0;
        info.geom = geomId;
        return true;
    }
    bool cubeIntersect(const frayhip_cube& Cb, int geomId, const Ray& ray, Hit& info)   // :85-137
    {
        st.prim++;
        info.dist = 1e99;
        Vec O = vec(Cb.O);
        double hs = Cb.halfSide;
        auto side = [&](double start, double dir, double target, Vec normal, int uvAxis) {
            if (fabs(dir) < 1e-9) return;
            double mult = (target - start) / dir;
            if (mult < 0) return;
            Vec ip = ray.start + ray.dir * mult;
            if (ip.x < O.x - hs - 1e-6 || ip.x > O.x + hs + 1e-6) return;
            if (ip.y < O.y - hs - 1e-6 || ip.y > O.y + hs + 1e-6) return;
            if (ip.z < O.z - hs - 1e-6 || ip.z > O.z + hs + 1e-6) return;
            double d = dist(ray.start, ip);
            if (d < info.dist) {
                info.dist = d;
                info.ip = ip;
                info.norm = normal;
                if (uvAxis == 0) { info.u = ip.y; info.v = ip.z; }
                else if (uvAxis == 1) { info.u = ip.x; info.v = ip.z; }
                else { info.u = ip.x; info.v = ip.y; }
            }
        };
        side(ray.start.x, ray.dir.x, O.x - hs, Vec{-1, 0, 0}, 0);
        side(ray.start.x, ray.dir.x, O.x + hs, Vec{+1, 0, 0}, 0);
        side(ray.start.y, ray.dir.y, O.y - hs, Vec{0, -1, 0}, 1);
        side(ray.start.y, ray.dir.y, O.y + hs, Vec{0, +1, 0}, 1);
        side(ray.start.z, ray.dir.z, O.z - hs, Vec{0, 0, -1}, 2);
        side(ray.start.z, ray.dir.z, O.z + hs, Vec{0, 0, +1}, 2);
        if (info.dist < 1e99) { info.geom = geomId; return true; }
        return false;
    }
    std::vector<Hit> allIntersections(const Ray& r, int g)   // findAllIntersections, :139-159
    {
        std::vector<Hit> result;
        Ray ray = r;
        int counter = 30;
        Vec origin = ray.start;
        Hit info;
        zeroHit(info);
        while (geomIntersect(g, ray, info) && counter-- > 0) {
            result.push_back(info);
            ray.start = info.ip + ray.dir * 1e-6;
        }
        for (size_t i = 1; i < result.size(); i++) result[i].dist = dist(result[i].ip, origin);
        return result;
    }
    bool csgIntersect(const frayhip_csg& C, int geomId, const Ray& ray, Hit& info)   // :161-194
    {
        std::vector<Hit> L = allIntersections(ray, C.left), R = allIntersections(ray, C.right), all;
        for (auto& h : L) all.push_back(h);
        for (auto& h : R) all.push_back(h);
        std::sort(all.begin(), all.end(), [](const Hit& a, const Hit& b) { return a.dist < b.dist; });
        bool inL = (L.size() % 2) == 1, inR = (R.size() % 2) == 1;
        auto op = [&](bool l, bool r) { return C.op == FRAYHIP_CSG_PLUS ? (l || r) : C.op == FRAYHIP_CSG_AND ? (l && r) : (l && !r); };
        bool cur = op(inL, inR);
        for (auto& h : all) {
            if (h.geom == C.left) inL = !inL; else inR = !inR;
            if (op(inL, inR) != cur) {
                info = h;
                info.geom = geomId;
                return true;
            }
        }
        return false;
    }
    static void zeroHit(Hit& h)
    {
        // The reference leaves IntersectionInfo uninitialised; fields a geometry does not write
        // (dNdx/dNdy outside meshes) are indeterminate there and defined as zero here.
        h.dist = 0; h.u = h.v = 0; h.geom = -1;
        h.ip = h.norm = h.dNdx = h.dNdy = Vec{0, 0, 0};
    }
    bool geomIntersect(int g, const Ray& ray, Hit& info)
    {
        const frayhip_geom_ref& ref = S.geoms[g];
        switch (ref.kind) {
            case FRAYHIP_GEOM_PLANE: return planeIntersect(S.planes[ref.index], g, ray, info);
            case FRAYHIP_GEOM_SPHERE: return sphereIntersect(S.spheres[ref.index], g, ray, info);
            case FRAYHIP_GEOM_CUBE: return cubeIntersect(S.cubes[ref.index], g, ray, info);
            case FRAYHIP_GEOM_MESH: return meshIntersect(S.meshes[ref.index], g, ray, info);
            case FRAYHIP_GEOM_CSG: return csgIntersect(S.csgs[ref.index], g, ray, info);
        }
        return false;
    }
    bool nodeIntersect(const frayhip_node& N, const Ray& ray, Hit& info)   // geometry.cpp:196-208
    {
        st.node++;
        Ray local = ray;
        local.start = untransformPoint(N.T, ray.start);
        local.dir = untransformDir(N.T, ray.dir);
        if (!geomIntersect(N.geom, local, info)) return false;
        info.ip = transformPoint(N.T, info.ip);
        info.norm = transformDir(N.T, info.norm);
        info.dist = dist(ray.start, info.ip);
        return true;
    }

    // ---------------------------------------------------------------- lights.cpp
    bool lightIntersect(const frayhip_light& L, const Ray& ray, Hit& info)   // :79-103; PointLight lights.h:68-71
    {
        if (L.kind == FRAYHIP_LIGHT_POINT) return false;
        st.prim++;
        Vec ls = untransformPoint(L.T, ray.start);
        Vec ld = untransformDir(L.T, ray.dir);
        if (ls.y >= 0) return false;
        if (ld.y <= 0) return false;
        double travelByY = fabs(ls.y);
        double unitTravel = fabs(ld.y);
        double scaling = travelByY / unitTravel;
        info.ip = ls + ld * scaling;
        if (fabs(info.ip.x) > 0.5 || fabs(info.ip.z) > 0.5) return false;
        info.norm = Vec{0, -1, 0};
        info.ip = transformPoint(L.T, info.ip);
        info.norm = transformDir(L.T, info.norm);
        info.dist = dist(ray.start, info.ip);
        return true;
    }
    int lightNumSamples(const frayhip_light& L) { return L.kind == FRAYHIP_LIGHT_POINT ? 1 : L.xSubd * L.ySubd; }
    Col lightColor(const frayhip_light& L) { return col(L.color) * L.power; }   // lights.h:45
    void lightNthSample(const frayhip_light& L, int idx, Vec shadePos, Vec& samplePos, Col& color)   // :31-35, :49-77
    {
        if (L.kind == FRAYHIP_LIGHT_POINT) {
            samplePos = vec(L.pos);
            color = col(L.color) * L.power;
            return;
        }
        int column = idx % L.xSubd;
        int row = idx / L.xSubd;
        double areaXsize = 1.0 / L.xSubd;
        double areaYsize = 1.0 / L.ySubd;
        double areaXstart = column * areaXsize;
        double areaYstart = row * areaYsize;
        double p_x = areaXstart + areaXsize * table.randfloat();
        double p_y = areaYstart + areaYsize * table.randfloat();
        Vec pointOnLight{p_x - 0.5, 0, p_y - 0.5};
        Vec sp = untransformPoint(L.T, shadePos);
        if (sp.y > 0) {
            color = BLACK;
        } else {
            float cosWeight = float(dot(Vec{0, -1, 0}, sp) / length(sp));
            color = col(L.color) * L.power * (float)L.area * cosWeight;
        }
        samplePos = transformPoint(L.T, pointOnLight);
    }
    double lightSolidAngle(const frayhip_light& L, const Hit& x)   // :105-108; base lights.h:47
    {
        if (L.kind == FRAYHIP_LIGHT_POINT) return 0;
        return L.area / std::max(1.0, lengthSqr(x.ip - vec(L.center)));
    }

    // ---------------------------------------------------------------- textures (shading.cpp)
    Col texel(const frayhip_texture& T, int x, int y)   // Bitmap::getPixel, bitmap.cpp:67-71
    {
        st.tex++;
        if (T.width <= 0 || x < 0 || x >= T.width || y < 0 || y >= T.height) return BLACK;
        return col(S.texels + T.texel_offset + 3 * ((int64_t)x + (int64_t)y * T.width));
    }
    void wrapTexel(const frayhip_texture& T, const Hit& info, int& ix, int& iy)   // shading.cpp:149-155, 404-410
    {
        ix = int(floor(info.u * T.scaling * T.width));
        iy = int(floor(info.v * T.scaling * T.height));
        ix %= T.width;
        iy %= T.height;
        if (ix < 0) ix += T.width;
        if (iy < 0) iy += T.height;
    }
    static float fresnel(Vec i, Vec n, float ior)   // shading.cpp:230-236
    {
        float f = (float)(((1.0f - ior) / (1.0f + ior)) * (double)((1.0f - ior) / (1.0f + ior)));   // sqr() takes and returns double
        float NdotI = (float)-dot(n, i);
        return f + (1.0f - f) * std::pow(1.0f - NdotI, 5.0f);   // pow(float,float) -> powf via <math.h> overloads
    }
    Col textureSample(int t, const Ray& ray, const Hit& info)
    {
        const frayhip_texture& T = S.textures[t];
        switch (T.kind) {
            case FRAYHIP_TEX_CHECKER: {   // shading.cpp:40-46
                int ix = int(floor(info.u * T.scaling) / 5.0);
                int iy = int(floor(info.v * T.scaling) / 5.0);
                return ((ix + iy) % 2 == 0) ? col(T.color1) : col(T.color2);
            }
            case FRAYHIP_TEX_BITMAP: {   // shading.cpp:147-158
                int ix, iy;
                wrapTexel(T, info, ix, iy);
                return texel(T, ix, iy);
            }
            case FRAYHIP_TEX_FRESNEL: {   // shading.cpp:369-385
                Vec n;
                double myIor;
                if (dot(ray.dir, info.norm) < 0) { n = info.norm; myIor = T.ior; }
                else { n = -info.norm; myIor = 1.0 / T.ior; }
                float f = fresnel(ray.dir, n, (float)myIor);
                return Col{f, f, f};
            }
            default: return BLACK;   // BumpTexture::sample, shading.cpp:392-395
        }
    }
    void applyBump(const frayhip_node& N, Hit& info)   // main.cpp:82-90, shading.cpp:397-418
    {
        if (N.bump_tex < 0) return;
        const frayhip_texture& T = S.textures[N.bump_tex];
        if (T.kind != FRAYHIP_TEX_BUMP) return;   // only BumpTexture implements BumpMapperInterface
        int ix, iy;
        wrapTexel(T, info, ix, iy);
        Col t = texel(T, ix, iy);
        float dx = (float)(t.r * T.bumpIntensity);
        float dy = (float)(t.g * T.bumpIntensity);
        info.norm = info.norm + (dx * info.dNdx + dy * info.dNdy) * T.bumpIntensity;
        info.norm = normalized(info.norm);
    }

    // ---------------------------------------------------------------- main.cpp:64-80
    bool visible(Vec a, Vec b)
    {
        st.shadow++;
        Ray ray;
        ray.dir = b - a;
        ray.start = a;
        double maxDist = dist(a, b);
        ray.dir = normalized(ray.dir);
        for (int i = 0; i < S.n_nodes; i++) {
            Hit info;
            zeroHit(info);
            if (nodeIntersect(S.nodes[i], ray, info) && info.dist < maxDist) return false;
        }
        return true;
    }

    // closest-hit loops shared by raytrace/pathtrace (main.cpp:178-199, 250-271)
    int closestHit(const Ray& ray, Hit& closest, int& lightIdx)
    {
        st.closest++;
        int closestNode = -1;
        closest.dist = 1e99;
        for (int i = 0; i < S.n_nodes; i++) {
            Hit info;
            zeroHit(info);
            if (nodeIntersect(S.nodes[i], ray, info) && info.dist < closest.dist) {
                closest = info;
                closestNode = i;
            }
        }
        if (closestNode >= 0) {   // byte model: the winner's 3 normals + 3 uvs
            const frayhip_geom_ref& g = S.geoms[S.nodes[closestNode].geom];
            if (g.kind == FRAYHIP_GEOM_MESH && !(S.meshes[g.index].faceted || S.meshes[g.index].n_normals == 0)) st.smooth++;
        }
        lightIdx = -1;
        for (int i = 0; i < S.n_lights; i++) {
            Hit info;
            zeroHit(info);
            if (lightIntersect(S.lights[i], ray, info) && info.dist < closest.dist) {
                closest = info;
                lightIdx = i;
            }
        }
        return closestNode;
    }

    Col environment(Vec dir)   // environment.cpp:64-98
    {
        const frayhip_environment& E = S.environment;
        if (!E.present || !E.loaded) return BLACK;
        double maxVal = fabs(dir.x);
        int dim = 0;
        if (fabs(dir.y) > maxVal) { dim = 1; maxVal = fabs(dir.y); }
        if (fabs(dir.z) > maxVal) dim = 2;
        bool positive = dir[dim] > 0;
        Vec on = dir * (1.0 / fabs(dir[dim]));
        int face = (positive ? 3 : 0) + dim;
        double sx, sy;
        switch (face) {
            case 0: sx = on.z; sy = -on.y; break;     // NEGX
            case 3: sx = -on.z; sy = -on.y; break;    // POSX
            case 1: sx = on.x; sy = -on.z; break;     // NEGY
            case 4: sx = on.x; sy = on.z; break;      // POSY
            case 2: sx = on.x; sy = on.y; break;      // NEGZ
            default: sx = on.x; sy = -on.y; break;    // POSZ
        }
        int W = E.width[face], H = E.height[face];
        int ix = (int)(((sx + 1) / 2) * W);
        int iy = (int)(((sy + 1) / 2) * H);
        st.tex++;
        if (ix < 0 || ix >= W || iy < 0 || iy >= H) return BLACK;
        return col(S.texels + E.texel_offset[face] + 3 * ((int64_t)ix + (int64_t)iy * W));
    }

    // ---------------------------------------------------------------- Whitted shade() family
    Col directLighting(const frayhip_shader& sh, const Ray& ray, const Hit& info, bool phong)   // shading.cpp:48-80, 101-144
    {
        Col diffuse = col(sh.color);
        if (sh.texture >= 0) diffuse = diffuse * textureSample(sh.texture, ray, info);
        Col result = diffuse * col(S.settings.ambientLight);
        for (int li = 0; li < S.n_lights; li++) {
            const frayhip_light& L = S.lights[li];
            int ns = lightNumSamples(L);
            Col sum = BLACK;
            for (int k = 0; k < ns; k++) {
                Col lc;
                Vec lp;
                lightNthSample(L, k, info.ip, lp, lc);
                double lightDistSqr = lengthSqr(info.ip - lp);
                Vec toLight = normalized(lp - info.ip);
                Vec n = faceforward(ray.dir, info.norm);
                float cosAngle = (float)dot(toLight, n);
                float lambertTerm = (float)(cosAngle / lightDistSqr);
                lambertTerm = std::max(0.0f, lambertTerm);
                if (visible(info.ip + n * 1e-6, lp)) {
                    Col r = diffuse * lc * lambertTerm;
                    if (phong) {
                        Vec fromLight = -toLight;
                        Vec rr = reflect(fromLight, n);
                        double cosCam = dot(-ray.dir, rr);
                        if (cosCam > 0)
                            r = r + lc / (float)lightDistSqr * col(sh.specularColor) * (float)pow(cosCam, sh.exponent) * (float)sh.specularMultiplier;
                    }
                    sum = sum + r;
                }
            }
            result = result + sum / (float)ns;
        }
        return result;
    }
    Col shade(int shaderIdx, const Ray& ray, const Hit& info)
    {
        const frayhip_shader& sh = S.shaders[shaderIdx];
        switch (sh.kind) {
            case FRAYHIP_SHADER_CONST: return col(sh.color);                       // shading.cpp:35-38
            case FRAYHIP_SHADER_LAMBERT: return directLighting(sh, ray, info, false);
            case FRAYHIP_SHADER_PHONG: return directLighting(sh, ray, info, true);
            case FRAYHIP_SHADER_REFL: {                                            // shading.cpp:160-207
                Vec n = faceforward(ray.dir, info.norm);
                if (sh.glossiness == 1.0) {
                    Ray nr = ray;
                    nr.start = info.ip + n * 1e-6;
                    nr.dir = reflect(ray.dir, n);
                    nr.depth = ray.depth + 1;
                    return raytrace(nr) * col(sh.mult);
                }
                Vec b, c;
                orthonormalSystem(n, b, c);
                Col sum = BLACK;
                int count = ray.depth == 0 ? sh.numSamples : 3;   // LOW_GLOSSY_SAMPLES, constants.h:36
                for (int i = 0; i < count; i++) {
                    double x, y;
                    Vec reflected;
                    while (1) {
                        table.unitDiscSample(x, y);
                        x *= sh.deflectionScaling;
                        y *= sh.deflectionScaling;
                        Vec nn = normalized(n + b * x + c * y);
                        reflected = reflect(ray.dir, nn);
                        if (dot(reflected, n) > 0) break;
                    }
                    Ray nr = ray;
                    nr.start = info.ip + n * 1e-6;
                    nr.dir = reflected;
                    nr.depth = ray.depth + 1;
                    sum = sum + raytrace(nr) * col(sh.mult);
                }
                return sum / (float)count;
            }
            case FRAYHIP_SHADER_REFR: {                                            // shading.cpp:238-263
                Vec n = faceforward(ray.dir, info.norm);
                double myIor = dot(n, info.norm) > 0 ? 1.0 / sh.ior : sh.ior / 1.0;
                Vec refr = refract(ray.dir, n, myIor);
                if (refr.x == 0 && refr.y == 0 && refr.z == 0) return BLACK;
                Ray nr = ray;
                nr.start = info.ip - n * 1e-6;
                nr.dir = refr;
                nr.depth = ray.depth + 1;
                return raytrace(nr) * col(sh.mult);
            }
            case FRAYHIP_SHADER_LAYERED: {                                         // shading.cpp:357-367
                Col result = BLACK;
                for (int i = 0; i < sh.layer_count; i++) {
                    const frayhip_layer& L = S.layers[sh.layer_begin + i];
                    Col opacity = L.texture >= 0 ? textureSample(L.texture, ray, info) : col(L.opacity);
                    result = shade(L.shader, ray, info) * opacity + (Col{1, 1, 1} - opacity) * result;
                }
                return result;
            }
        }
        return BLACK;
    }
    Col raytrace(const Ray& ray)   // main.cpp:246-285
    {
        if (ray.depth > S.settings.maxTraceDepth) return BLACK;
        Hit ci;
        int light;
        int node = closestHit(ray, ci, light);
        if (light >= 0) return lightColor(S.lights[light]);
        if (node < 0) return environment(ray.dir);
        applyBump(S.nodes[node], ci);
        return shade(S.nodes[node].shader, ray, ci);
    }

    // ---------------------------------------------------------------- BRDF eval / spawnRay
    Vec hemisphereSample(const Hit& info)   // main.cpp:92-116
    {
        double u = table.randdouble();
        double v = table.randdouble();
        double theta = 2 * PI * u;
        double phi = acos(2 * v - 1);
        Vec dir{sin(phi) * cos(theta), cos(phi), sin(phi) * sin(theta)};
        if (dot(dir, info.norm) > 0) return dir;
        return -dir;
    }
    Col brdfEval(const frayhip_shader& sh, const Hit& x, Vec w_out)
    {
        switch (sh.kind) {
            case FRAYHIP_SHADER_LAMBERT: {   // shading.cpp:82-86
                float cosTerm = (float)std::max(0.0, dot(x.norm, w_out));
                return col(sh.color) * (float)(cosTerm / PI);
            }
            case FRAYHIP_SHADER_REFL:        // shading.cpp:209-215
            case FRAYHIP_SHADER_REFR:        // shading.cpp:265-268
                return BLACK;
            default: return Col{1, 0, 0};    // Shader::eval default, shading.h:124-127
        }
    }
    void spawnRay(const frayhip_shader& sh, const Hit& x, const Ray& w_in, Ray& w_out, Col& brdf, float& pdf)
    {
        switch (sh.kind) {
            case FRAYHIP_SHADER_LAMBERT: {   // shading.cpp:88-99
                w_out = w_in;
                w_out.depth++;
                w_out.start = x.ip + x.norm * 1e-6;
                w_out.dir = hemisphereSample(x);
                w_out.flags |= RF_DIFFUSE;
                float cosTerm = (float)std::max(0.0, dot(x.norm, w_out.dir));
                brdf = col(sh.color) * (float)(cosTerm / PI);
                pdf = (float)(1 / (2 * PI));
                return;
            }
            case FRAYHIP_SHADER_REFL: {      // shading.cpp:217-227
                Vec n = faceforward(w_in.dir, x.norm);
                w_out = w_in;
                w_out.depth++;
                w_out.start = x.ip + n * 1e-6;
                w_out.dir = reflect(w_in.dir, x.norm);
                w_out.flags &= ~RF_DIFFUSE;
                brdf = col(sh.mult) * 1e9f;
                pdf = 1e9f;
                return;
            }
            case FRAYHIP_SHADER_REFR: {      // shading.cpp:270-299
                Vec n = faceforward(w_in.dir, x.norm);
                double myIor = dot(n, x.norm) > 0 ? 1.0 / sh.ior : sh.ior / 1.0;
                Vec refr = refract(w_in.dir, n, myIor);
                if (!(refr.x == 0 && refr.y == 0 && refr.z == 0)) {
                    w_out = w_in;
                    w_out.start = x.ip - n * 1e-6;
                    w_out.dir = refr;
                    w_out.depth = w_in.depth + 1;
                    w_out.flags &= ~RF_DIFFUSE;
                    brdf = col(sh.mult) * 1e9f;
                    pdf = 1e9f;
                } else {
                    brdf = BLACK;
                    pdf = 1.0;
                }
                return;
            }
            default:                          // Shader::spawnRay default, shading.h:128-134
                w_out = w_in;
                w_out.depth++;
                brdf = Col{1, 0, 0};
                pdf = 1;
        }
    }

    // ---------------------------------------------------------------- path tracer
    Col explicitLightSample(const Ray& ray, const Hit& info, Col pm, const frayhip_shader& sh)   // main.cpp:118-169
    {
        if (S.n_lights == 0) return BLACK;
        int lightIdx = rnd.randint(0, S.n_lights - 1);
        const frayhip_light& L = S.lights[lightIdx];
        Vec x = info.ip;
        double solidAngle = lightSolidAngle(L, info);
        if (solidAngle == 0) return BLACK;
        int randSample = rnd.randint(0, lightNumSamples(L) - 1);
        Vec pointOnLight;
        Col unused;
        lightNthSample(L, randSample, x, pointOnLight, unused);
        if (!visible(x + info.norm * 1e-6, pointOnLight)) return BLACK;
        Col Le = lightColor(L);
        Vec w_out = normalized(pointOnLight - x);
        Col brdfAtPoint = brdfEval(sh, info, w_out);
        if (brdfAtPoint.intensity() == 0) return BLACK;
        float probHitLightArea = (float)(1.0f / solidAngle);
        float probPickThisLight = 1.0f / S.n_lights;
        float chooseLightProb = probHitLightArea * probPickThisLight;
        return Le * pm * brdfAtPoint / chooseLightProb;
    }
    Col pathtrace(Ray ray, Col pm)   // main.cpp:171-244; the tail recursion is written as a loop that keeps the
    {                                 // pending `contribLight + (...)` sums and adds them innermost-first
        std::vector<Col> pending;
        Col result;
        while (true) {
            if (ray.depth > S.settings.maxTraceDepth || pm.intensity() < 0.01) { result = BLACK; break; }
            Hit ci;
            int light;
            int node = closestHit(ray, ci, light);
            if (light >= 0) {
                if (ray.flags & RF_DIFFUSE) result = BLACK;
                else result = lightColor(S.lights[light]) * pm;
                break;
            }
            if (node < 0) { result = environment(ray.dir) * pm; break; }
            const frayhip_node& N = S.nodes[node];
            const frayhip_shader& sh = S.shaders[N.shader];
            applyBump(N, ci);
            Ray newRay = ray;
            newRay.depth++;
            newRay.start = ci.ip + ci.norm * 1e-6;
            Col brdfColor;
            float rayPdf;
            spawnRay(sh, ci, ray, newRay, brdfColor, rayPdf);       // discarded, but consumes RNG (main.cpp:219-224)
            Col contribLight = explicitLightSample(ray, ci, pm, sh);
            Ray w_out = ray;
            w_out.depth++;
            Col brdf;
            float pdf;
            spawnRay(sh, ci, ray, w_out, brdf, pdf);
            if (pdf == -1) { result = Col{1, 0, 0}; break; }
            if (pdf == 0) { result = BLACK; break; }
            pending.push_back(contribLight);
            ray = w_out;
            pm = pm * brdf / pdf;
        }
        for (size_t k = pending.size(); k-- > 0;) result = pending[k] + result;
        return result;
    }

    // ---------------------------------------------------------------- camera.cpp:59-92, main.cpp:287-321
    Ray screenRay(double x, double y, int which)
    {
        Ray r;
        r.dir = cam.topLeft + (cam.topRight - cam.topLeft) * (x / cam.w) + (cam.bottomLeft - cam.topLeft) * (y / cam.h);
        r.dir = normalized(r.dir);
        r.start = cam.pos;
        if (which == 1) r.start = r.start + cam.rightDir * -cam.stereoSeparation;
        else if (which == 2) r.start = r.start + cam.rightDir * cam.stereoSeparation;
        return r;
    }
    Ray dofRay(double x, double y, int which)
    {
        Ray ray = screenRay(x, y, which);
        Vec d = ray.dir;
        double M = cam.focalPlaneDist / dot(cam.frontDir, d);
        Vec T = cam.pos + d * M;
        double u, v;
        table.unitDiscSample(u, v);
        u *= cam.apertureSize;
        v *= cam.apertureSize;
        ray.start = ray.start + (u * cam.rightDir + v * cam.upDir);
        ray.dir = normalized(T - ray.start);
        return ray;
    }
    Col trace(const Ray& ray)
    {
        st.samples++;
        if (S.settings.gi) return pathtrace(ray, Col{1, 1, 1});
        return raytrace(ray);
    }
    Col singlePixel(double x, double y)
    {
        auto getRay = [&](int which) { return cam.dof ? dofRay(x, y, which) : screenRay(x, y, which); };
        if (cam.stereoSeparation > 0) {
            Ray l = getRay(1), r = getRay(2);
            Col cl = trace(l), cr = trace(r);
            float sat = S.settings.saturation;
            if (sat != 1) {
                auto adj = [&](Col& c) { float mid = (c.r + c.g + c.b) / 3.0f; c.r = mid + (c.r - mid) * sat; c.g = mid + (c.g - mid) * sat; c.b = mid + (c.b - mid) * sat; };
                adj(cl); adj(cr);
            }
            return cl * col(S.camera.leftMask) + cr * col(S.camera.rightMask);
        }
        return trace(getRay(0));
    }
};

int samplesPerPixel(const frayhip_scene_desc& S)   // main.cpp:395-400
{
    int spp = S.settings.wantAA ? 5 : 1;
    if (S.camera.dof) spp = std::max(spp, S.camera.numDOFSamples);
    if (S.settings.gi) spp = std::max(spp, S.settings.numPaths);
    return spp;
}

const double kOffsets[5][2] = {{0, 0}, {0.6, 0}, {0.3, 0.3}, {0, 0.6}, {0.6, 0.6}};   // main.cpp:55-61

}  // namespace

extern "C" {

// Same contract as frayhip_render (include/frayhip.h), computed on the CPU with n_threads
// std::threads pulling buckets from an atomic cursor (main.cpp:331-370, sdl.cpp:243-262).
int fray_oracle_render(const frayhip_scene_desc* desc, const frayhip_frame* f, float* rgb, int32_t* hit_id,
                       double* hit_dist, frayhip_stats* out_stats, int n_threads)
{
    if (!desc || !f) return FRAYHIP_E_ARG;
    const frayhip_scene_desc& S = *desc;
    const int W = S.settings.frameWidth, H = S.settings.frameHeight;
    if (W <= 0 || H <= 0) return FRAYHIP_E_ARG;
    const int stride = f->bucket_stride > 0 ? f->bucket_stride : 1;
    const int first = f->bucket_first;
    const int BW = (W - 1) / 48 + 1, BH = (H - 1) / 48 + 1;
    std::vector<int> buckets;
    for (int b = 0; b < BW * BH; b++) if (b % stride == first) buckets.push_back(b);
    CameraFrame cam = cameraBeginFrame(S.camera, W, H);
    const int spp = samplesPerPixel(S);
    const bool jitter = S.camera.dof || S.settings.gi;
    if (n_threads < 1) n_threads = 1;
    std::atomic<int> cursor{0};
    std::vector<Stats> stats(n_threads);
    auto t0 = std::chrono::steady_clock::now();
    auto worker = [&](int tid) {
        Tracer T(S, cam);
        while (true) {
            int k = cursor++;
            if (k >= (int)buckets.size()) break;
            int by = buckets[k] / BW, bx = (buckets[k] % BW + FRAYHIP_BUCKET_SKEW * by) % BW;   // the bucket numbering of include/frayhip.h
            int x1 = std::min(W, (bx + 1) * 48), y1 = std::min(H, (by + 1) * 48);
            for (int y = by * 48; y < y1; y++)
                for (int x = bx * 48; x < x1; x++) {
                    size_t p = (size_t)y * W + x;
                    if (f->mode == FRAYHIP_MODE_PRIMARY_ID) {
                        Ray ray = T.screenRay(x, y, 0);
                        Hit ci;
                        int light;
                        int node = T.closestHit(ray, ci, light);
                        if (hit_id) hit_id[p] = light >= 0 ? -2 - light : node;
                        if (hit_dist) hit_dist[p] = ci.dist;
                        continue;
                    }
                    Col avg = BLACK;
                    for (int i = 0; i < spp; i++) {
                        uint32_t s = sample_seed(f->seed, (uint32_t)p, (uint32_t)i);
                        T.rnd.seed(s);
                        T.table.seed(s);
                        float ox, oy;
                        if (jitter) { ox = T.rnd.randfloat(); oy = T.rnd.randfloat(); }
                        else { ox = (float)kOffsets[i][0]; oy = (float)kOffsets[i][1]; }
                        avg = avg + T.singlePixel(x + ox, y + oy);   // int + float, as main.cpp:359
                    }
                    avg = avg / (float)spp;
                    if (rgb) { rgb[p * 3] = avg.r; rgb[p * 3 + 1] = avg.g; rgb[p * 3 + 2] = avg.b; }
                }
        }
        stats[tid] = T.st;
    };
    if (n_threads == 1) worker(0);
    else {
        std::vector<std::thread> th;
        for (int i = 0; i < n_threads; i++) th.emplace_back(worker, i);
        for (auto& t : th) t.join();
    }
    auto t1 = std::chrono::steady_clock::now();
    if (out_stats) {
        frayhip_stats o{};
        for (auto& s : stats) {
            o.closest_rays += s.closest; o.shadow_rays += s.shadow; o.node_tests += s.node; o.kd_inner_visits += s.kdInner;
            o.leaf_refs += s.leafRefs; o.tri_tests += s.tri; o.prim_tests += s.prim; o.smooth_hits += s.smooth;
            o.samples += s.samples; o.texture_fetches += s.tex;
        }
        o.ms_total = std::chrono::duration<double, std::milli>(t1 - t0).count();
        *out_stats = o;
    }
    return FRAYHIP_OK;
}

// FNV-1a-64 over a byte range (the known-answer hash of SURVEY.md 8c).
uint64_t fray_oracle_fnv1a64(const void* data, uint64_t n)
{
    const unsigned char* p = (const unsigned char*)data;
    uint64_t h = 14695981039346656037ull;
    for (uint64_t i = 0; i < n; i++) { h ^= p[i]; h *= 1099511628211ull; }
    return h;
}

uint32_t fray_oracle_sample_seed(uint32_t seed, uint32_t pixel, uint32_t sample) { return sample_seed(seed, pixel, sample); }

// First n raw words / canonical floats / doubles of a freshly seeded generator -- lets the tests
// check the device RNG restatement word for word.
void fray_oracle_rng_words(uint32_t seed, int n, uint32_t* out)
{
    std::mt19937 g(seed);
    for (int i = 0; i < n; i++) out[i] = (uint32_t)g();
}
void fray_oracle_rng_stream(uint32_t seed, int n, float* floats, double* doubles, int32_t* ints, int int_hi)
{
    Rng a, b, c;
    a.seed(seed); b.seed(seed); c.seed(seed);
    for (int i = 0; i < n; i++) {
        if (floats) floats[i] = a.randfloat();
        if (doubles) doubles[i] = b.randdouble();
        if (ints) ints[i] = c.randint(0, int_hi);
    }
}

// Single-ray probe: closest hit with the full intersection record (for cross-checks against
// oracle/_ref and for the strided golden fixtures).  out: [dist, ip3, norm3, u, v] = 9 doubles.
int fray_oracle_probe(const frayhip_scene_desc* desc, const double* start, const double* dir, double* out)
{
    CameraFrame cam = cameraBeginFrame(desc->camera, desc->settings.frameWidth, desc->settings.frameHeight);
    Tracer T(*desc, cam);
    Ray r;
    r.start = vec(start);
    r.dir = vec(dir);
    Hit ci;
    Tracer::zeroHit(ci);
    int light;
    int node = T.closestHit(r, ci, light);
    out[0] = ci.dist;
    out[1] = ci.ip.x; out[2] = ci.ip.y; out[3] = ci.ip.z;
    out[4] = ci.norm.x; out[5] = ci.norm.y; out[6] = ci.norm.z;
    out[7] = ci.u; out[8] = ci.v;
    return light >= 0 ? -2 - light : node;
}

void fray_oracle_camera_ray(const frayhip_scene_desc* desc, double x, double y, double* start, double* dir)
{
    CameraFrame cam = cameraBeginFrame(desc->camera, desc->settings.frameWidth, desc->settings.frameHeight);
    Tracer T(*desc, cam);
    Ray r = T.screenRay(x, y, 0);
    start[0] = r.start.x; start[1] = r.start.y; start[2] = r.start.z;
    dir[0] = r.dir.x; dir[1] = r.dir.y; dir[2] = r.dir.z;
}

}  // extern "C"
