// Test harness (CPU): fray_amd/csrc/dev_misscert.hpp -- "this ray surely misses this box, so nothing inside it reports an intersection" -- against the
// reference's own intersection routines restated below (Sphere / Cube::intersect, geometry.cpp:52-137; BBox::testIntersect, bbox.h:79-134, which guards
// every mesh), each given its exact bounding box: random rays, rays aimed to graze faces, edges and corners at offsets around the certificate's margin,
// starts far away and starts within the margin of the box, axis-parallel directions with exact zeros.  Exit code 1 if a certified ray is reported hit.
// Built a second time with -DFRAY_MISSCERT_SCALE=0 -DNO_HOST_MARGIN the same harness must find contradictions (it then tests a certificate without margins).
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define FRAY_CERT_FN static inline
#include "dev_misscert.hpp"

struct V { double x, y, z; };
static V operator+(V a, V b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static V operator-(V a, V b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static V operator*(V a, double m) { return {a.x * m, a.y * m, a.z * m}; }
static double dot(V a, V b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static double comp(V a, int k) { return k == 0 ? a.x : k == 1 ? a.y : a.z; }

static bool ref_sphere(V O, double R, V s, V d)
{
    V H = s - O;
    double A = 1, B = 2 * dot(d, H), C = dot(H, H) - R * R;
    double Disc = B * B - 4 * A * C;
    if (Disc < 0) return false;
    double sq = sqrt(Disc), p1 = (-B + sq) / (2 * A), p2 = (-B - sq) / (2 * A);
    double larger = p1 < p2 ? p2 : p1;
    return !(larger < 0);
}
static bool ref_cube(V O, double hs, V s, V d)
{
    bool hit = false;
    for (int side = 0; side < 6; side++) {
        const int ax = side >> 1;
        const double st = comp(s, ax), dr = comp(d, ax), target = (side & 1) ? comp(O, ax) + hs : comp(O, ax) - hs;
        if (fabs(dr) < 1e-9) continue;
        double mult = (target - st) / dr;
        if (mult < 0) continue;
        V ip = s + d * mult;
        if (ip.x < O.x - hs - 1e-6 || ip.x > O.x + hs + 1e-6) continue;
        if (ip.y < O.y - hs - 1e-6 || ip.y > O.y + hs + 1e-6) continue;
        if (ip.z < O.z - hs - 1e-6 || ip.z > O.z + hs + 1e-6) continue;
        hit = true;
    }
    return hit;
}
static bool ref_box(const double lo[3], const double hi[3], V s, V d)
{
    const double sv[3] = {s.x, s.y, s.z}, dv[3] = {d.x, d.y, d.z};
    double r[3];
    for (int k = 0; k < 3; k++) r[k] = fabs(dv[k]) > 1e-12 ? 1.0 / dv[k] : 1e12;      // RRay::prepareForTracing
    bool in = true;
    for (int k = 0; k < 3; k++) in = in && lo[k] - 1e-6 <= sv[k] && sv[k] <= hi[k] + 1e-6;
    if (in) return true;
    for (int dim = 0; dim < 3; dim++) {
        if ((dv[dim] < 0 && sv[dim] < lo[dim]) || (dv[dim] > 0 && sv[dim] > hi[dim])) return false;
        if (fabs(dv[dim]) < 1e-9) continue;
        const int u = dim == 0 ? 1 : 0, v = dim == 2 ? 1 : 2;
        double dist = (lo[dim] - sv[dim]) * r[dim];
        if (dist < 0) continue;
        double x = sv[u] + dv[u] * dist;
        if (lo[u] <= x && x <= hi[u]) { double y = sv[v] + dv[v] * dist; if (lo[v] <= y && y <= hi[v]) return true; }
        dist = (hi[dim] - sv[dim]) * r[dim];
        if (dist < 0) continue;
        x = sv[u] + dv[u] * dist;
        if (lo[u] <= x && x <= hi[u]) { double y = sv[v] + dv[v] * dist; if (lo[v] <= y && y <= hi[v]) return true; }
    }
    return false;
}

static uint64_t rs = 0x9E3779B97F4A7C15ULL;
static uint64_t rnd() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return rs; }
static double u01() { return (rnd() >> 11) * (1.0 / 9007199254740992.0); }
static double sym() { return 2 * u01() - 1; }
static double mag(double lo, double hi) { return pow(10.0, lo + (hi - lo) * u01()); }
static int pick(int n) { return (int)(rnd() % (uint64_t)n); }
static V unit()
{
    for (;;) { V d = {sym(), sym(), sym()}; double l = sqrt(dot(d, d)); if (l > 1e-3 && l <= 1) return d * (1.0 / l); }
}

int main(int argc, char** argv)
{
    const long N = argc > 1 ? atol(argv[1]) : 2000000;
#ifdef NO_HOST_MARGIN
    const double hostMargin = 0;
#else
    const double hostMargin = 1e-5;
#endif
    long certified = 0, certified32 = 0, realMiss = 0, bad = 0;
    for (long it = 0; it < N; it++) {
        // the box (and the geometry it is the exact bound of)
        const V c = {sym() * mag(-2, 3), sym() * mag(-2, 3), sym() * mag(-2, 3)};
        const int what = pick(3);                      // 0 sphere, 1 cube, 2 mesh box
        double h[3];
        if (what == 2) for (int k = 0; k < 3; k++) h[k] = pick(6) == 0 ? 0.0 : mag(-3, 2);       // flat boxes too (a wall)
        else h[0] = h[1] = h[2] = mag(-3, 2);
        const double lo[3] = {c.x - h[0], c.y - h[1], c.z - h[2]}, hi[3] = {c.x + h[0], c.y + h[1], c.z + h[2]};
        // the ray
        V s, d;
        const int mode = pick(5);
        const double far = mag(-3, 4);
        if (mode == 0) { s = c + unit() * far; d = unit(); }
        else {
            // aim at a point on / near the box's surface, edges and corners; offsets from 1e-9 to 1e-2 around the margins
            V t = c;
            double* tp[3] = {&t.x, &t.y, &t.z};
            for (int k = 0; k < 3; k++) {
                const int w = pick(4);
                double off = (pick(2) ? 1 : -1) * mag(-9, -2);
                if (w == 0) *tp[k] += sym() * h[k];
                else if (w == 1) *tp[k] += h[k] + off;
                else if (w == 2) *tp[k] -= h[k] + off;
                else *tp[k] += (pick(2) ? h[k] : -h[k]);
            }
            s = mode == 4 ? t + unit() * mag(-7, -3) : c + unit() * far;          // mode 4: the start itself sits next to the surface
            V to = t - s;
            double l = sqrt(dot(to, to));
            if (!(l > 1e-12)) continue;
            d = to * (1.0 / l);
            if (mode == 2) d = d * -1.0;                                           // the box behind the ray
            if (mode == 3) {                                                      // axis-parallel: exact zeros in the direction
                const int k = pick(3);
                d = {k == 0 ? (pick(2) ? 1.0 : -1.0) : 0.0, k == 1 ? (pick(2) ? 1.0 : -1.0) : 0.0, k == 2 ? (pick(2) ? 1.0 : -1.0) : 0.0};
            }
        }
        double M = 0;
        for (int k = 0; k < 3; k++) M = fmax(M, fabs(comp(c, k)) + h[k] + hostMargin);
        bool cert = ray_surely_misses_box(c.x, c.y, c.z, h[0] + hostMargin, h[1] + hostMargin, h[2] + hostMargin, M, s.x, s.y, s.z, d.x, d.y, d.z);
        {   // the FP32 form, fed the way the host (capi.hip: float centre, half extents rounded up over the centre's rounding) and the device (ray_gate_class) feed it;
            // every other case with a direction that is not a unit vector (a shadow segment)
            const float cf[3] = {(float)c.x, (float)c.y, (float)c.z};
            float hf[3], Mf = 0;
            for (int k = 0; k < 3; k++) {
                hf[k] = nextafterf((float)(h[k] + hostMargin + fabs(comp(c, k) - (double)cf[k])), INFINITY);
                Mf = fmaxf(Mf, nextafterf(fabsf(cf[k]) + hf[k], INFINITY));
            }
            const double stretch = (it & 1) ? mag(-2, 3) : 1.0;
            const float ox = (float)s.x, oy = (float)s.y, oz = (float)s.z, dx = (float)(d.x * stretch), dy = (float)(d.y * stretch), dz = (float)(d.z * stretch);
            const bool cert32 = ray_surely_misses_box_f32(cf[0], cf[1], cf[2], hf[0], hf[1], hf[2], Mf, ox, oy, oz, dx, dy, dz,
                                                          fmaxf(fmaxf(fabsf(ox), fabsf(oy)), fabsf(oz)), fabsf(dx) + fabsf(dy) + fabsf(dz));
            if (cert32) certified32++;
            cert = cert || cert32;
        }
        const bool hit = what == 0 ? ref_sphere(c, h[0], s, d) : what == 1 ? ref_cube(c, h[0], s, d) : ref_box(lo, hi, s, d);
        if (!hit) realMiss++;
        if (cert) {
            certified++;
            if (hit) {
                if (bad < 10) fprintf(stderr, "CONTRADICTION (%s, mode %d): certified miss, the reference's routine reports a hit\n  c %.17g %.17g %.17g  h %.17g %.17g %.17g\n  s %.17g %.17g %.17g\n  d %.17g %.17g %.17g\n",
                                      what == 0 ? "sphere" : what == 1 ? "cube" : "box", mode, c.x, c.y, c.z, h[0], h[1], h[2], s.x, s.y, s.z, d.x, d.y, d.z);
                bad++;
            }
        }
    }
    printf("cases %ld, reference misses %ld, certified %ld (%.1f %% of the misses; by the FP32 form %ld), contradictions %ld\n", N, realMiss, certified, 100.0 * certified / (realMiss ? realMiss : 1), certified32, bad);
    return bad ? 1 : 0;
}
