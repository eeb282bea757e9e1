// Test harness (CPU): fray_amd/csrc/dev_nodecert.hpp against the reference's own arithmetic, restated below from bbox.h:79-134
// (BBox::inside / testIntersect), triangle.cpp:66-97 (Triangle::intersectFast), geometry.cpp:196-208 + matrix.cpp:137-161
// (Node::intersect's transforms and distance).
//   part A: flat boxes.  Whenever the reference's triangle test accepts a triangle of a flat mesh and flat_box_sure() says the box test is
//           proved, the reference's box test must say true.  Rays are aimed at the triangles' edges and corners (offsets 1e-17 .. 1e-3), at
//           barycentrics around delta, start on / next to the plane, graze it (|dir_f| 1e-12 .. 1e-3), start next to the slabs' faces moving
//           either way, come from 1e-9 .. 1e6 away; triangles include slivers and polygons whose corners sit on the box's edges.
//   part B: transform classes.  For hits at parameters t_a, t_b of one local ray: |t_a - t_b| > class_order_margin  =>  the world distances
//           order the same way (strictly); sHi t + E < maxDist  =>  dist < maxDist;  sLo t - E >= maxDist  =>  !(dist < maxDist).  Transforms:
//           identity, rotations, uniform and non-uniform scales, shears, with offsets up to 1e4; pairs of parameters from 1e-3 to 1e3 margins apart.
// Exit code 1 on any contradiction.   usage: nodecert_check [cases [deltaScale [marginScale]]]   (scales of 0 show that the harness sees
// contradictions once the margins are removed)
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define FRAY_CERT_FN static inline
#include "dev_nodecert.hpp"

struct V { double x, y, z; };
static V operator-(V a, V b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static V operator+(V a, V b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static V operator*(V a, double m) { return {a.x * m, a.y * m, a.z * m}; }
static V operator-(V a) { return {-a.x, -a.y, -a.z}; }
static double dot(V a, V b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static V cross(V a, V b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
static double det(V a, V b, V c) { return dot(cross(a, b), c); }
static double length(V a) { return sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }
static V normalized(V a) { double m = 1.0 / length(a); return a * m; }        // vector.h:81-85
static double& comp(V& a, int k) { return k == 0 ? a.x : (k == 1 ? a.y : a.z); }
static double compc(const V& a, int k) { return k == 0 ? a.x : (k == 1 ? a.y : a.z); }

struct Box { double lo[3], hi[3]; };
static bool ref_inside(const Box& b, const V& p)
{
    return b.lo[0] - 1e-6 <= p.x && p.x <= b.hi[0] + 1e-6 && b.lo[1] - 1e-6 <= p.y && p.y <= b.hi[1] + 1e-6 &&
           b.lo[2] - 1e-6 <= p.z && p.z <= b.hi[2] + 1e-6;
}
static bool ref_box(const Box& b, const V& s, const V& d)
{
    V r;      // RRay::prepareForTracing
    r.x = fabs(d.x) > 1e-12 ? 1.0 / d.x : 1e12; r.y = fabs(d.y) > 1e-12 ? 1.0 / d.y : 1e12; r.z = fabs(d.z) > 1e-12 ? 1.0 / d.z : 1e12;
    if (ref_inside(b, s)) return true;
    for (int dim = 0; dim < 3; dim++) {
        const double dd = compc(d, dim), sd = compc(s, dim);
        if ((dd < 0 && sd < b.lo[dim]) || (dd > 0 && sd > b.hi[dim])) return false;
        if (fabs(dd) < 1e-9) continue;
        const double mul = compc(r, dim);
        const int u = dim == 0 ? 1 : 0, v = dim == 2 ? 1 : 2;
        double dist = (b.lo[dim] - sd) * mul;
        if (dist < 0) continue;
        double x = compc(s, u) + compc(d, u) * dist;
        if (b.lo[u] <= x && x <= b.hi[u]) {
            double y = compc(s, v) + compc(d, v) * dist;
            if (b.lo[v] <= y && y <= b.hi[v]) return true;
        }
        dist = (b.hi[dim] - sd) * mul;
        if (dist < 0) continue;
        x = compc(s, u) + compc(d, u) * dist;
        if (b.lo[u] <= x && x <= b.hi[u]) {
            double y = compc(s, v) + compc(d, v) * dist;
            if (b.lo[v] <= y && y <= b.hi[v]) return true;
        }
    }
    return false;
}
static bool ref_intersect_fast(V start, V dir, V A, V AB, V AC, V N, double& minDist, double& l2, double& l3)
{
    V D = -dir;
    double Dcr = dot(N, D);
    if (fabs(Dcr) < 1e-12) return false;
    double rDcr = 1 / Dcr;
    V H = start - A;
    double gamma = dot(N, H) * rDcr;
    if (gamma < 0 || gamma > minDist) return false;
    double lambda2 = det(H, AC, D) * rDcr;
    if (lambda2 < 0 || lambda2 > 1) return false;
    double lambda3 = det(AB, H, D) * rDcr;
    if (lambda3 < 0 || lambda3 > 1) return false;
    double lambda1 = 1 - (lambda2 + lambda3);
    if (lambda1 < 0) return false;
    minDist = gamma; l2 = lambda2; l3 = lambda3;
    return true;
}

static uint64_t rs = 0xA0761D6478BD642FULL;
static uint64_t rnd() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return rs; }
static double u01() { return (rnd() >> 11) * (1.0 / 9007199254740992.0); }
static double sym() { return 2 * u01() - 1; }
static double tiny() { return sym() * pow(10.0, -3 - 14 * u01()); }      // +-1e-17 .. 1e-3
static int pick(int n) { return (int)(rnd() % (uint64_t)n); }

struct TriRec { double A[3], AB[3], AC[3], N[3]; };
static void put(double* p, V v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }
static V get(const double* p) { return {p[0], p[1], p[2]}; }

static long partA(long n, double deltaScale)
{
    long accepted = 0, sure = 0, boxFalse = 0, bad = 0, meshes = 0, flatMeshes = 0;
    for (long it = 0; it < n;) {
        // ---- a flat mesh: a fan-triangulated polygon (or a few loose triangles) in the plane x_f = c
        const int f = pick(3), u = f == 0 ? 1 : 0, v = f == 2 ? 1 : 2;
        const double size = pow(10.0, -3 + 7 * u01());
        const double c = pick(3) == 0 ? 0.0 : sym() * size * pow(10.0, 2 * u01());
        const double cu = pick(2) ? 0.0 : sym() * size * 10, cv = pick(2) ? 0.0 : sym() * size * 10;
        const int nv = 3 + pick(4);
        std::vector<V> P(nv);
        const bool sliver = pick(6) == 0;
        for (int k = 0; k < nv; k++) {
            const double ang = 2 * M_PI * (k + 0.3 * sym()) / nv, rad = size * (0.3 + 0.7 * u01());
            V p = {0, 0, 0};
            comp(p, f) = c; comp(p, u) = cu + rad * cos(ang); comp(p, v) = cv + rad * sin(ang) * (sliver ? pow(10.0, -1 - 5 * u01()) : 1.0);
            if (pick(3) == 0) { comp(p, u) = (double)(float)comp(p, u); comp(p, v) = (double)(float)comp(p, v); }
            P[k] = p;
        }
        std::vector<TriRec> T;
        for (int k = 1; k + 1 < nv; k++) {                    // mesh.cpp:214-231: (v0, vk, vk+1)
            TriRec r;
            const V AB = P[k] - P[0], AC = P[k + 1] - P[0];
            put(r.A, P[0]); put(r.AB, AB); put(r.AC, AC); put(r.N, cross(AB, AC));
            T.push_back(r);
        }
        Box b;
        for (int k = 0; k < 3; k++) { b.lo[k] = 1e300; b.hi[k] = -1e300; }
        for (auto& p : P) for (int k = 0; k < 3; k++) { b.lo[k] = fmin(b.lo[k], compc(p, k)); b.hi[k] = fmax(b.hi[k], compc(p, k)); }
        if (c == 0 && pick(2)) for (int k = 0; k < 3; k++) { b.lo[k] = fmin(b.lo[k], 0.0); b.hi[k] = fmax(b.hi[k], 0.0); }   // the OBJ loader's dummy vertex 0
        double boxMax = 0;
        for (int k = 0; k < 3; k++) boxMax = fmax(boxMax, fmax(fabs(b.lo[k]), fabs(b.hi[k])));
        double flatR;
        const int fd = flat_make(T.data(), (int)T.size(), b.lo, b.hi, boxMax, flatR);
        meshes++;
        if (fd < 0) { it += 16; continue; }
        flatMeshes++;
        if (deltaScale == 0) flatR = 1e300;
        // ---- rays at it
        for (int q = 0; q < 256; q++, it++) {
            const TriRec& r = T[pick((int)T.size())];
            const V A = get(r.A), AB = get(r.AB), AC = get(r.AC), N = get(r.N);
            double l2, l3;
            switch (pick(8)) {
                case 0: l2 = u01(); l3 = u01() * (1 - l2); break;                                     // interior
                case 1: l2 = tiny(); l3 = u01(); break;                                                // at an edge
                case 2: l2 = u01(); l3 = tiny(); break;
                case 3: l2 = u01(); l3 = 1 - l2 + tiny(); break;
                case 4: l2 = FRAY_FLAT_DELTA * (1 + tiny()); l3 = u01(); break;                        // around delta
                case 5: l2 = u01(); l3 = 1 - l2 - FRAY_FLAT_DELTA * (1 + tiny()); break;
                case 6: l2 = pick(2) ? tiny() : 1 + tiny(); l3 = pick(2) ? tiny() : 1 - l2 + tiny(); break;   // at a corner
                default: l2 = 2 * sym(); l3 = 2 * sym(); break;                                         // anywhere in the plane
            }
            V target = A + AB * l2 + AC * l3;
            V dir = {sym(), sym(), sym()};
            const int gz = pick(5);
            if (gz == 0) comp(dir, f) = sym() * pow(10.0, -12 + 9 * u01());                            // grazing the plane
            if (gz == 1) { comp(dir, u) = tiny(); }                                                     // nearly axis parallel
            dir = normalized(dir);
            double dist = pow(10.0, -9 + 15 * u01()) * size;
            if (pick(6) == 0) dist = fabs(tiny()) * size;                                               // starts on / next to the plane
            V s = target - dir * dist;
            if (pick(8) == 0) comp(s, f) = c + tiny() * size * 1e-3;                                    // start within a hair of the plane
            if (pick(10) == 0) { const int k = pick(2) ? u : v; comp(s, k) = (pick(2) ? b.lo[k] : b.hi[k]) + tiny() * size; }   // start next to a slab face
            if (pick(12) == 0) dir = -dir;                                                              // moving away
            double best = 1e99, a2 = 0, a3 = 0;
            if (!ref_intersect_fast(s, dir, A, AB, AC, N, best, a2, a3)) continue;
            accepted++;
            const double sAbsMax = fmax(fmax(fabs(s.x), fabs(s.y)), fabs(s.z));
            bool ok = flat_box_sure(a2, a3, best, sAbsMax, compc(dir, f), flatR);
            if (deltaScale == 0) ok = !(fabs(compc(dir, f)) < 1e-9);
            const bool box = ref_box(b, s, dir);
            if (!box) boxFalse++;
            if (ok) { sure++; if (!box) bad++; }
        }
    }
    printf("part A: %ld meshes (%ld flat), %ld accepted triangle hits, %ld proved, %ld with a FALSE box test among the accepted, %ld contradictions\n",
           meshes, flatMeshes, accepted, sure, boxFalse, bad);
    return bad;
}

static V mulM(V v, const double* m) { return {v.x * m[0] + v.y * m[3] + v.z * m[6], v.x * m[1] + v.y * m[4] + v.z * m[7], v.x * m[2] + v.y * m[5] + v.z * m[8]}; }
static void matmul(const double* a, const double* b, double* o)
{
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double s = 0; for (int k = 0; k < 3; k++) s += a[3 * i + k] * b[3 * k + j]; o[3 * i + j] = s; }
}
static bool inverse(const double* m, double* o)
{
    const double d = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
    if (fabs(d) < 1e-300) return false;
    const double r = 1 / d;
    o[0] = (m[4] * m[8] - m[5] * m[7]) * r; o[1] = (m[2] * m[7] - m[1] * m[8]) * r; o[2] = (m[1] * m[5] - m[2] * m[4]) * r;
    o[3] = (m[5] * m[6] - m[3] * m[8]) * r; o[4] = (m[0] * m[8] - m[2] * m[6]) * r; o[5] = (m[2] * m[3] - m[0] * m[5]) * r;
    o[6] = (m[3] * m[7] - m[4] * m[6]) * r; o[7] = (m[1] * m[6] - m[0] * m[7]) * r; o[8] = (m[0] * m[4] - m[1] * m[3]) * r;
    return true;
}
// Node::intersect's distance for a hit at parameter t of the local ray (geometry.cpp:196-208)
static double F(V o, V ls, V ld, double t, const double* m, const double* off)
{
    V ipl = ls + ld * t;
    V ipw = mulM(ipl, m) + V{off[0], off[1], off[2]};
    return length(o - ipw);
}

static long partB(long n, double marginScale)
{
    long bad = 0, decidedOrder = 0, undecidedOrder = 0, decidedVis = 0, undecidedVis = 0, classes = 0, okClasses = 0;
    double worst = 0;
    for (long it = 0; it < n;) {
        double m[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, inv[9], off[3] = {0, 0, 0};
        const int kind = pick(6);
        auto rot = [&](int ax, double a) {
            double r[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, o[9];
            const int p = (ax + 1) % 3, q = (ax + 2) % 3;
            r[3 * p + p] = cos(a); r[3 * p + q] = sin(a); r[3 * q + p] = -sin(a); r[3 * q + q] = cos(a);
            matmul(m, r, o); for (int k = 0; k < 9; k++) m[k] = o[k];
        };
        if (kind >= 1) { rot(2, sym() * M_PI); rot(0, sym() * M_PI); rot(1, sym() * M_PI); }
        if (kind == 2) { const double s = pow(10.0, 3 * sym()); for (int k = 0; k < 9; k++) m[k] *= s; }
        if (kind == 3) { double sc[9] = {pow(10.0, 2 * sym()), 0, 0, 0, pow(10.0, 2 * sym()), 0, 0, 0, pow(10.0, 2 * sym())}, o[9]; matmul(m, sc, o); for (int k = 0; k < 9; k++) m[k] = o[k]; }
        if (kind == 4) { m[1] += sym(); m[5] += sym() * 0.5; }
        if (kind == 5) for (int k = 0; k < 9; k++) m[k] = sym() * pow(10.0, sym());                   // anything, some nearly singular
        if (pick(3)) for (int k = 0; k < 3; k++) off[k] = sym() * pow(10.0, 4 * u01());
        if (!inverse(m, inv)) { it += 8; continue; }
        if (pick(20) == 0) inv[pick(9)] *= 1 + 1e-9 * sym();                                          // an inverse that is a little off
        DClass C;
        class_make(off, m, inv, C);
        classes++;
        if (!C.ok) { it += 8; continue; }
        okClasses++;
        if (marginScale == 0) { C.EA = 0; C.EB = 0; }
        for (int q = 0; q < 128; q++, it++) {
            const double scale = pow(10.0, -2 + 6 * u01());
            V o = {sym() * scale, sym() * scale, sym() * scale};
            V d = normalized(V{sym(), sym(), sym()});
            const V ls = mulM(o - V{off[0], off[1], off[2]}, inv);
            const V ld = normalized(mulM(d, inv));
            const double W = fabs(o.x) + fabs(o.y) + fabs(o.z);
            const double ta = pow(10.0, -6 + 10 * u01());
            const double mu0 = class_order_margin(C, W, ta, ta);
            double tb;
            switch (pick(4)) {
                case 0: tb = ta + (pick(2) ? 1 : -1) * mu0 * pow(10.0, 3 * sym()); break;                 // around the margin
                case 1: tb = ta * (1 + tiny()); break;
                case 2: tb = nextafter(ta, pick(2) ? 1e300 : 0.0); break;
                default: tb = pow(10.0, -6 + 10 * u01()); break;
            }
            if (!(tb >= 0)) tb = 0;
            const double fa = F(o, ls, ld, ta, m, off), fb = F(o, ls, ld, tb, m, off);
            const double mu = class_order_margin(C, W, ta, tb);
            if (ta <= 1e30 && tb <= 1e30 && (ta < tb - mu || ta > tb + mu)) {
                decidedOrder++;
                const bool byT = ta < tb;
                if (byT != (fa < fb) || fa == fb) bad++;
            } else undecidedOrder++;
            // how much of E the true error uses: F(t) against sLo t - E, sHi t + E
            {
                const double E = class_err(C, W, ta);
                if (marginScale != 0) {
                    if (fa > C.sHi * ta + E || fa < C.sLo * ta - E) bad++;
                    const V ldm = mulM(ld, m);
                    const double sc = length(ldm);
                    if (E > 0) worst = fmax(worst, fabs(fa - sc * ta) / E);
                }
                // visible(): maxDist around the hit's distance
                double maxDist;
                switch (pick(4)) {
                    case 0: maxDist = fa * (1 + tiny()); break;
                    case 1: maxDist = fa + (pick(2) ? 1 : -1) * E * pow(10.0, 2 * sym()); break;
                    case 2: maxDist = nextafter(fa, pick(2) ? 1e300 : 0.0); break;
                    default: maxDist = fa * pow(10.0, sym()); break;
                }
                const bool closer = C.sHi * ta + E < maxDist, notCloser = C.sLo * ta - E >= maxDist;
                if (closer || notCloser) {
                    decidedVis++;
                    if (closer && notCloser) bad++;
                    if (closer && !(fa < maxDist)) bad++;
                    if (notCloser && (fa < maxDist)) bad++;
                } else undecidedVis++;
            }
        }
    }
    printf("part B: %ld transforms (%ld covered), order: %ld decided / %ld computed, visible: %ld decided / %ld computed, worst |F - s t| / E = %.3g, %ld contradictions\n",
           classes, okClasses, decidedOrder, undecidedOrder, decidedVis, undecidedVis, worst, bad);
    return bad;
}

int main(int argc, char** argv)
{
    const long n = argc > 1 ? atol(argv[1]) : 4000000;
    const double deltaScale = argc > 2 ? atof(argv[2]) : 1.0, marginScale = argc > 3 ? atof(argv[3]) : 1.0;
    long bad = partA(n, deltaScale);
    bad += partB(n, marginScale);
    return bad ? 1 : 0;
}
