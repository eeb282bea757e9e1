// Test harness (CPU): fray_amd/csrc/dev_tricert.hpp -- the certified "surely misses" filter in front of Triangle::intersectFast --
// against the reference's own arithmetic (restated below from triangle.cpp:27-31, 66-97, scalar, with its early returns), over
// adversarial rays: aimed at edges and vertices with offsets from 1e-17 to 1e-3 of the triangle, grazing the triangle's plane,
// starting on the triangle, far away (up to 1e5 triangle sizes), at slivers, at triangles 1e-9 .. 1e6 in size far from the
// mesh's reference point.
// Exit code 1 if a triangle the filter calls SURELY REJECTED is accepted by the reference's arithmetic (with minDist = INF and
// without backface culling: the largest set of triangles the reference can accept).
// usage: tricert_check [cases [kappa]]      (kappa = 0 shows that the harness sees contradictions when the margins are removed)
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define FRAY_CERT_FN static inline
#include "dev_tricert.hpp"

struct V { double x, y, z; };
static V operator-(V a, V b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static V operator+(V a, V b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static V operator*(V a, double m) { return {a.x * m, a.y * m, a.z * m}; }
static V operator-(V a) { return {-a.x, -a.y, -a.z}; }
static double dot(V a, V b) { return a.x * b.x + a.y * b.y + a.z * b.z; }                                  // vector.h operator*
static V cross(V a, V b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }  // vector.h operator^
static double det(V a, V b, V c) { return dot(cross(a, b), c); }                                            // triangle.cpp:27-31

// Triangle::intersectFast, triangle.cpp:66-97
static bool ref_intersect_fast(V start, V dir, V A, V AB, V AC, V ABcrossAC, double& minDist, double& l2, double& l3)
{
    V D = -dir;
    double Dcr = dot(ABcrossAC, D);
    if (fabs(Dcr) < 1e-12) return false;
    double rDcr = 1 / Dcr;
    V H = start - A;
    double gamma = dot(ABcrossAC, H) * rDcr;
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

static uint64_t rs = 0xD1B54A32D192ED03ULL;
static uint64_t rnd() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return rs; }
static double u01() { return (rnd() >> 11) * (1.0 / 9007199254740992.0); }
static double sym() { return 2 * u01() - 1; }
static double tiny() { return sym() * pow(10.0, -3 - 14 * u01()); }      // +-1e-17 .. 1e-3
static int pick(int n) { return (int)(rnd() % (uint64_t)n); }
static V rv() { return {sym(), sym(), sym()}; }

int main(int argc, char** argv)
{
    const long n = argc > 1 ? atol(argv[1]) : 4000000;
    const float kappa = argc > 2 ? (float)atof(argv[2]) : FRAY_TRICERT_KAPPA;
    long sure = 0, cand = 0, hits = 0, bad = 0, unfiltered = 0;
    for (long it = 0; it < n; it++) {
        // ---- a triangle: size 1e-9 .. 1e6, at up to 1e3 sizes (sometimes 1e6) from the mesh's reference point
        const double size = pow(10.0, -9 + 15 * u01());
        const double away = pick(8) ? size * pow(10.0, 3 * u01()) : size * pow(10.0, 6 * u01());
        const V ref = rv() * (pick(2) ? 0.0 : 1000.0 * u01());
        const V A = ref + rv() * away;
        V B = A + rv() * size, C = A + rv() * size;
        if (pick(6) == 0) C = A + (B - A) * (2 * u01() - 0.5) + rv() * (size * fabs(tiny()));          // sliver
        if (pick(24) == 0) C = A + (B - A) * u01();                                                     // degenerate (up to rounding)
        const V AB = B - A, AC = C - A, N = cross(AB, AC);
        DTri32 rec;
        const double a3[3] = {A.x, A.y, A.z}, ab3[3] = {AB.x, AB.y, AB.z}, ac3[3] = {AC.x, AC.y, AC.z}, n3[3] = {N.x, N.y, N.z}, r3[3] = {ref.x, ref.y, ref.z};
        tricert_make(rec, a3, ab3, ac3, n3, r3, kappa);
        // ---- rays: from a start of some kind to a target of some kind
        for (int rr = 0; rr < 4; rr++) {
            // target: a point of the triangle's plane given by (l2, l3)
            double l2, l3;
            const int tk = pick(7);
            if (tk == 0) { l2 = u01(); l3 = u01() * (1 - l2); }                                          // inside
            else if (tk == 1) { l2 = u01(); l3 = 0; }                                                    // on edge AB
            else if (tk == 2) { l2 = 0; l3 = u01(); }                                                    // on edge AC
            else if (tk == 3) { l2 = u01(); l3 = 1 - l2; }                                               // on edge BC
            else if (tk == 4) { const int v = pick(3); l2 = v == 1; l3 = v == 2; }                       // a vertex
            else if (tk == 5) { l2 = 3 * sym(); l3 = 3 * sym(); }                                        // anywhere around
            else { l2 = u01() + 0.3 * sym() * u01(); l3 = (1 - l2) * u01() + 0.3 * sym() * u01(); }      // near
            if (pick(3)) { l2 += tiny(); l3 += tiny(); }
            const V tgt = A + AB * l2 + AC * l3;
            V s;
            const int sk = pick(7);
            const V nn = N * (1.0 / (sqrt(dot(N, N)) + 1e-300));
            if (sk == 0) s = tgt + rv() * (size * pow(10.0, 5 * u01()));                                 // far: up to 1e5 sizes
            else if (sk == 1) s = tgt + rv() * size;
            else if (sk == 2) s = tgt + rv() * (size * fabs(tiny()));                                    // right at the target
            else if (sk == 3) s = A + AB * (3 * sym()) + AC * (3 * sym()) + nn * (size * tiny());        // grazing: almost in the plane
            else if (sk == 4) s = A + AB * (3 * sym()) + AC * (3 * sym());                               // in the plane (up to rounding)
            else if (sk == 5) s = ref + rv() * (away * 2);
            else s = tgt + nn * (size * (pick(2) ? 1 : -1) * pow(10.0, 4 * u01() - 2)) + rv() * (size * fabs(tiny()));   // head-on
            V d = tgt - s;
            if (pick(24) == 0) d = rv();
            if (pick(32) == 0) d = -d;
            const double len = sqrt(dot(d, d));
            if (!(len > 0)) continue;
            d = d * (1.0 / len);
            const float sx = (float)(s.x - ref.x), sy = (float)(s.y - ref.y), sz = (float)(s.z - ref.z);
            const bool rayOk = fmaxf(fmaxf(fabsf(sx), fabsf(sy)), fabsf(sz)) <= 1e9f;
            if (!rayOk || std::isinf(rec.Lq)) unfiltered++;
            const bool miss = rayOk && tri_sure_miss(rec.A[0], rec.A[1], rec.A[2], rec.AB[0], rec.AB[1], rec.AB[2], rec.AC[0], rec.AC[1], rec.AC[2], rec.Lq, rec.Cq,
                                                     sx, sy, sz, (float)d.x, (float)d.y, (float)d.z);
            double md = 1e99, o2, o3;
            const bool ex = ref_intersect_fast(s, d, A, AB, AC, N, md, o2, o3);
            hits += ex;
            if (miss) sure++; else cand++;
            if (miss && ex) {
                if (bad < 10)
                    fprintf(stderr, "MISMATCH: filter says surely rejected, reference accepts (l2 %.17g l3 %.17g gamma %.17g)\n  A %.17g %.17g %.17g\n  AB %.17g %.17g %.17g\n  AC %.17g %.17g %.17g\n"
                            "  s %.17g %.17g %.17g\n  d %.17g %.17g %.17g\n  ref %.17g %.17g %.17g  Lq %.9g Cq %.9g\n", o2, o3, md, A.x, A.y, A.z, AB.x, AB.y, AB.z, AC.x, AC.y, AC.z,
                            s.x, s.y, s.z, d.x, d.y, d.z, ref.x, ref.y, ref.z, (double)rec.Lq, (double)rec.Cq);
                bad++;
            }
        }
    }
    printf("rays %ld  surely rejected %ld  candidates %ld  of which the reference accepts %ld  unfiltered %ld  mismatches %ld\n", sure + cand, sure, cand, hits, unfiltered, bad);
    return bad ? 1 : 0;
}
