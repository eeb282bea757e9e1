// Test harness (CPU): fray_amd/csrc/dev_boxcert.hpp -- the certified shortcut of BBox::testIntersect -- against the reference's
// own arithmetic (restated below from bbox.h:79-134, scalar, with its early returns), over adversarial rays: aimed at faces,
// edges and corners with offsets from 1e-17 to 1e-3, starting inside / on / within inside()'s 1e-6 shell of the box, nearly
// axis-parallel, and down chains of midpoint splits the way Mesh::buildKD (mesh.cpp:316-345) makes them, with the child
// intervals derived incrementally (tstate_child) exactly as the device walk does.
// Exit code 1 if a box classified SURELY TRUE / SURELY FALSE is decided the other way by the reference's arithmetic.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define FRAY_CERT_FN static inline
#include "dev_boxcert.hpp"

struct V { double v[3]; };
struct Box { double lo[3], hi[3]; };

static bool ref_inside(const Box& b, const V& p)
{
    return b.lo[0] - 1e-6 <= p.v[0] && p.v[0] <= b.hi[0] + 1e-6 && b.lo[1] - 1e-6 <= p.v[1] && p.v[1] <= b.hi[1] + 1e-6 &&
           b.lo[2] - 1e-6 <= p.v[2] && p.v[2] <= b.hi[2] + 1e-6;
}
static bool ref_test(const Box& b, const V& s, const V& d, const V& r)
{
    if (ref_inside(b, s)) return true;
    for (int dim = 0; dim < 3; dim++) {
        if ((d.v[dim] < 0 && s.v[dim] < b.lo[dim]) || (d.v[dim] > 0 && s.v[dim] > b.hi[dim])) return false;
        if (fabs(d.v[dim]) < 1e-9) continue;
        const double mul = r.v[dim];
        const int u = dim == 0 ? 1 : 0, v = dim == 2 ? 1 : 2;
        double dist = (b.lo[dim] - s.v[dim]) * mul;
        if (dist < 0) continue;
        double x = s.v[u] + d.v[u] * dist;
        if (b.lo[u] <= x && x <= b.hi[u]) {
            double y = s.v[v] + d.v[v] * dist;
            if (b.lo[v] <= y && y <= b.hi[v]) return true;
        }
        dist = (b.hi[dim] - s.v[dim]) * mul;
        if (dist < 0) continue;
        x = s.v[u] + d.v[u] * dist;
        if (b.lo[u] <= x && x <= b.hi[u]) {
            double y = s.v[v] + d.v[v] * dist;
            if (b.lo[v] <= y && y <= b.hi[v]) return true;
        }
    }
    return false;
}

static uint64_t rs = 0x9E3779B97F4A7C15ULL;
static uint64_t rnd() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return rs; }
static double u01() { return (rnd() >> 11) * (1.0 / 9007199254740992.0); }
static double sym() { return 2 * u01() - 1; }
static double tiny() { return sym() * pow(10.0, -3 - 14 * u01()); }      // +-1e-17 .. 1e-3
static int pick(int n) { return (int)(rnd() % (uint64_t)n); }

struct Tally { long n = 0, yes = 0, no = 0, unc = 0, bad = 0; };

static void prep(const V& s, const V& d, V& r, double boxMax, CertRay& c)
{
    for (int k = 0; k < 3; k++) r.v[k] = fabs(d.v[k]) > 1e-12 ? 1.0 / d.v[k] : 1e12;      // RRay::prepareForTracing
    const double rmax = fmax(fmax(fabs(r.v[0]), fabs(r.v[1])), fabs(r.v[2]));
    const double sMax = fmax(fmax(fabs(s.v[0]), fabs(s.v[1])), fabs(s.v[2]));
    const double dMin = fmin(fmin(fabs(d.v[0]), fabs(d.v[1])), fabs(d.v[2]));
    c = cert_ray(rmax, sMax, dMin >= 1e-6, boxMax);
}
static void check(const Box& b, const V& s, const V& d, const V& r, TState t, const CertRay& c, Tally& T, const char* what)
{
    const int cls = cert_classify(t, c);
    const bool ex = ref_test(b, s, d, r);
    T.n++;
    if (cls < 0) { T.unc++; return; }
    if (cls) T.yes++; else T.no++;
    if ((cls != 0) != ex) {
        if (T.bad < 10)
            fprintf(stderr, "MISMATCH (%s): classified %d, reference %d\n  box [%.17g %.17g %.17g] [%.17g %.17g %.17g]\n  s %.17g %.17g %.17g\n  d %.17g %.17g %.17g\n  t0 %.17g n2 %.17g t1 %.17g mu %.3g A %.3g\n",
                    what, cls, (int)ex, b.lo[0], b.lo[1], b.lo[2], b.hi[0], b.hi[1], b.hi[2], s.v[0], s.v[1], s.v[2], d.v[0], d.v[1], d.v[2], t.t0, t.n2, t.t1, c.mu, c.A);
        T.bad++;
    }
}
static void normalise(V& d)
{
    const double m = 1.0 / sqrt(d.v[0] * d.v[0] + d.v[1] * d.v[1] + d.v[2] * d.v[2]);
    for (int k = 0; k < 3; k++) d.v[k] *= m;
}
static double box_max(const Box& b)
{
    double m = 0;
    for (int k = 0; k < 3; k++) m = fmax(m, fmax(fabs(b.lo[k]), fabs(b.hi[k])));
    return m;
}
static TState state_of(const Box& b, const V& s, const V& r)
{
    return tstate_box(b.lo[0], b.lo[1], b.lo[2], b.hi[0], b.hi[1], b.hi[2], s.v[0], s.v[1], s.v[2], r.v[0], r.v[1], r.v[2]);
}

// a point on (or `off` away from) a random face / edge / corner / interior point of the box
static V target(const Box& b, int kind, double off)
{
    V p;
    for (int k = 0; k < 3; k++) p.v[k] = b.lo[k] + (b.hi[k] - b.lo[k]) * u01();
    int fixed = kind == 0 ? 0 : kind;                 // 0 interior, 1 face, 2 edge, 3 corner
    int perm[3] = {0, 1, 2};
    for (int i = 2; i > 0; i--) { int j = pick(i + 1); int t = perm[i]; perm[i] = perm[j]; perm[j] = t; }
    for (int i = 0; i < fixed; i++) {
        const int k = perm[i];
        p.v[k] = (pick(2) ? b.hi[k] : b.lo[k]) + (pick(3) ? tiny() : 0) * (pick(2) ? 1.0 : off);
    }
    return p;
}

int main(int argc, char** argv)
{
    const long n = argc > 1 ? atol(argv[1]) : 4000000;
    Tally single, chain;
    for (long it = 0; it < n; it++) {
        // ---- a box at a random scale and position
        const double scale = pow(10.0, -2 + 5 * u01()), centre = pick(4) ? scale * 3 * sym() : scale * 1000 * sym();
        Box b;
        for (int k = 0; k < 3; k++) {
            const double a = centre + scale * sym(), e = scale * pow(10.0, -3 * u01()) * u01();
            b.lo[k] = a; b.hi[k] = a + e;
        }
        // ---- a ray: from a random start (far, near, on the surface, inside the 1e-6 shell, inside) through a target
        V s, d, r;
        const int sk = pick(6);
        if (sk == 0) for (int k = 0; k < 3; k++) s.v[k] = centre + scale * 30 * sym();
        else if (sk == 1) for (int k = 0; k < 3; k++) s.v[k] = b.lo[k] + (b.hi[k] - b.lo[k]) * (1.5 * u01() - 0.25);
        else if (sk == 2) s = target(b, 1 + pick(3), 1);                                       // on a face / edge / corner, +- tiny
        else if (sk == 3) { s = target(b, 1 + pick(3), 1); for (int k = 0; k < 3; k++) s.v[k] += 2e-6 * sym() * (pick(2) ? 1 : u01()); }   // around the shell
        else if (sk == 4) for (int k = 0; k < 3; k++) s.v[k] = b.lo[k] + (b.hi[k] - b.lo[k]) * u01();
        else for (int k = 0; k < 3; k++) s.v[k] = centre + scale * 3 * sym();
        const V tg = target(b, pick(4), 1);
        for (int k = 0; k < 3; k++) d.v[k] = tg.v[k] - s.v[k];
        if (pick(8) == 0) d.v[pick(3)] = tiny() * (pick(2) ? 1e-6 : 1);                         // nearly axis-parallel
        if (pick(16) == 0) for (int k = 0; k < 3; k++) d.v[k] = sym();                          // anything
        if (pick(32) == 0) for (int k = 0; k < 3; k++) d.v[k] = -d.v[k];                        // pointing away
        if (d.v[0] == 0 && d.v[1] == 0 && d.v[2] == 0) d.v[0] = 1;
        normalise(d);
        CertRay c;
        prep(s, d, r, box_max(b), c);
        check(b, s, d, r, state_of(b, s, r), c, single, "single box");

        // ---- the same ray down a chain of midpoint splits, child intervals derived incrementally
        Box cur = b;
        TState st = state_of(cur, s, r);
        const int depth0 = pick(3), levels = 1 + pick(40);
        for (int lv = 0; lv < levels; lv++) {
            const int axis = (depth0 + lv) % 3;
            const double split = (cur.lo[axis] + cur.hi[axis]) * 0.5;                          // findOptimalSplitPlane, mesh.cpp:316-319
            const double ts = (split - s.v[axis]) * r.v[axis];
            Box ch[2] = {cur, cur};
            ch[0].hi[axis] = split; ch[1].lo[axis] = split;
            TState cs[2];
            for (int q = 0; q < 2; q++) {
                cs[q] = tstate_child(st, ts, (q == 0) == (r.v[axis] > 0));
                check(ch[q], s, d, r, cs[q], c, chain, "split chain");
            }
            // follow the child the target lies in (mostly), so that the chain stays near the ray
            int go = tg.v[axis] < split ? 0 : 1;
            if (pick(8) == 0) go ^= 1;
            cur = ch[go];
            st = cs[go];
            if (cur.hi[axis] - cur.lo[axis] <= 0) break;
        }
    }
    printf("single boxes : %ld  surely true %ld  surely false %ld  uncertain %ld (%.3f %%)  mismatches %ld\n", single.n, single.yes, single.no, single.unc,
           100.0 * single.unc / single.n, single.bad);
    printf("split chains : %ld  surely true %ld  surely false %ld  uncertain %ld (%.3f %%)  mismatches %ld\n", chain.n, chain.yes, chain.no, chain.unc,
           100.0 * chain.unc / chain.n, chain.bad);
    return single.bad || chain.bad ? 1 : 0;
}
