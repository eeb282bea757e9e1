// FP64 vector / FP32 colour helpers for the device code.  Operation order inside each helper is
// the reference's (src/vector.h, src/color.h, src/matrix.h); the translation unit is compiled with
// -ffp-contract=off so no a*b+c is fused, which is what keeps hit records bit-identical to the
// CPU reference build.
#pragma once
#include <hip/hip_runtime.h>

#define FD __device__ __forceinline__
#include "dev_scene.hpp"

// Truth values of the hottest predicated code as wave-wide lane masks in scalar registers (dev_trace.hpp box_dim).
// lanes(cmp): one v_cmp writing the mask; only active lanes ever have their bit set.  lane_of(mask): this lane's bit as a
// bool again (no instruction: the mask is used as the condition).
#ifdef __HIP_DEVICE_COMPILE__
typedef unsigned long long lanes_t;
FD lanes_t lanes(bool p) { return __builtin_amdgcn_ballot_w64(p); }
FD bool lane_of(lanes_t m) { return __builtin_amdgcn_inverse_ballot_w64(m); }
#else
typedef unsigned long long lanes_t;      // host pass: declarations only, never called
__device__ lanes_t lanes(bool p);
__device__ bool lane_of(lanes_t m);
#endif

// Diagnostic build only (-DFRAY_STAMPS, never shipped): s_memtime stamps that attribute a wave's cycles to
// sections of the trace kernels.  STAMP(k) charges the cycles since the wave's previous stamp to section k.  The
// sums live in LDS (every active lane writes the same value, so any exec mask works) and leave the kernel
// through DStats.stamp, which nothing else reads.
#ifdef FRAY_STAMPS
__shared__ unsigned long long g_stampT0[4];
__shared__ unsigned long long g_stampAcc[4][24];
__shared__ unsigned long long g_stampLanes[4][24];     // the same cycles weighted by the lanes active at the stamp: lanes / (64 x cycles) = how full the section ran
FD void STAMP(int k)
{
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    const int w = threadIdx.x >> 6;
    g_stampAcc[w][k] += t - g_stampT0[w];
    g_stampLanes[w][k] += (t - g_stampT0[w]) * (unsigned long long)__builtin_popcountll(__builtin_amdgcn_ballot_w64(true));
    g_stampT0[w] = t;
    __builtin_amdgcn_sched_barrier(0);
}
FD void stamp_begin()
{
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    const int w = threadIdx.x >> 6;
    for (int k = 0; k < 24; k++) { g_stampAcc[w][k] = 0; g_stampLanes[w][k] = 0; }
    g_stampT0[w] = t;
}
#else
#define STAMP(k) do { } while (0)
#endif

// Division, reciprocal, square root.  FRAY_ARITH == 0 (every translation unit but render_contract.hip): IEEE, the reference's.  FRAY_ARITH == 1 (the
// path tracer's kernels for rays after a sample's first closest hit, option "fp_contract"): the hardware's reciprocal / reciprocal square root with the two
// refinement steps the IEEE expansions start with, but without their scaling, fix-up and final correction -- results within an ulp or two, 5 and 9
// instructions instead of 13 and 17.  (Operands here are lengths, determinants and direction components: never subnormal, never huge.)
#ifndef FRAY_ARITH
#define FRAY_ARITH 0
#endif
#if FRAY_ARITH && defined(__HIP_DEVICE_COMPILE__)
FD double fray_rcp(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-x, r, 1.0);
    return __builtin_fma(r, e, r);
}
FD double fray_div(double a, double b) { return a * fray_rcp(b); }
FD double fray_rsqrt(double x)
{
    double y = __builtin_amdgcn_rsq(x);
    double e = __builtin_fma(-x * y, y, 1.0);          // 1 - x y^2
    y = __builtin_fma(0.5 * y, e, y);
    e = __builtin_fma(-x * y, y, 1.0);
    return __builtin_fma(0.5 * y, e, y);
}
FD double fray_sqrt(double x)
{
    if (!(x > 0.0)) return x < 0.0 ? __builtin_nan("") : x;          // +-0 (a degenerate segment) stays what sqrt makes of it
    const double y = fray_rsqrt(x);
    const double g = x * y;
    return __builtin_fma(__builtin_fma(-g, g, x), 0.5 * y, g);       // one correction of x y towards sqrt(x)
}
#else
FD double fray_rcp(double x) { return 1.0 / x; }
FD double fray_div(double a, double b) { return a / b; }
FD double fray_sqrt(double x) { return sqrt(x); }
#endif

struct V3 { double x, y, z; };
struct C3 { float r, g, b; };

FD V3 v3(double x, double y, double z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
FD V3 ld3(const double* p) { return v3(p[0], p[1], p[2]); }
#ifdef __HIP_DEVICE_COMPILE__
FD V3 ld3(const FRAY_RO double* p) { return v3(p[0], p[1], p[2]); }
#endif
FD V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
FD V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
FD V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
FD V3 operator*(V3 a, double m) { return v3(a.x * m, a.y * m, a.z * m); }
FD V3 operator*(double m, V3 a) { return v3(a.x * m, a.y * m, a.z * m); }
FD double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
FD V3 cross(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
FD double lengthSqr(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
FD double length(V3 a) { return fray_sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }
#if FRAY_ARITH && defined(__HIP_DEVICE_COMPILE__)
FD V3 normalized(V3 a) { return a * fray_rsqrt(a.x * a.x + a.y * a.y + a.z * a.z); }
#else
FD V3 normalized(V3 a) { double m = 1.0 / length(a); return a * m; }   // vector.h:81-85
#endif
FD double comp(V3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }
FD V3 faceforward(V3 d, V3 n) { return dot(d, n) < 0 ? n : -n; }        // vector.h:169-175
FD V3 reflect(V3 i, V3 n) { return i + 2 * dot(-i, n) * n; }            // vector.h:178-181
FD V3 refract(V3 i, V3 n, double ior)                                   // vector.h:184-191
{
    double NdotI = dot(i, n);
    double k = 1 - (ior * ior) * (1 - NdotI * NdotI);
    if (k < 0.0) return v3(0, 0, 0);
    return normalized(ior * i - (ior * NdotI + fray_sqrt(k)) * n);
}
FD void orthonormalSystem(V3 a, V3& b, V3& c)                           // vector.h:197-213
{
    V3 t = v3(1, 0, 0);
    if (fabs(dot(t, a)) > 0.9) t = v3(0, 1, 0);
    b = normalized(cross(a, t));
    c = cross(a, b);
}
// v * M, row-vector convention (matrix.h:36-45); m is row-major 3x3.
template <class P>
FD V3 mulM(V3 v, P m)
{
    return v3(v.x * m[0] + v.y * m[3] + v.z * m[6], v.x * m[1] + v.y * m[4] + v.z * m[7],
              v.x * m[2] + v.y * m[5] + v.z * m[8]);
}

FD C3 c3(float r, float g, float b) { C3 c; c.r = r; c.g = g; c.b = b; return c; }
FD C3 ldc(const float* p) { return c3(p[0], p[1], p[2]); }
#ifdef __HIP_DEVICE_COMPILE__
FD C3 ldc(const FRAY_RO float* p) { return c3(p[0], p[1], p[2]); }
#endif
FD C3 operator+(C3 a, C3 b) { return c3(a.r + b.r, a.g + b.g, a.b + b.b); }
FD C3 operator-(C3 a, C3 b) { return c3(a.r - b.r, a.g - b.g, a.b - b.b); }
FD C3 operator*(C3 a, C3 b) { return c3(a.r * b.r, a.g * b.g, a.b * b.b); }
FD C3 operator*(C3 a, float m) { return c3(a.r * m, a.g * m, a.b * m); }
FD C3 operator/(C3 a, float d) { return c3(a.r / d, a.g / d, a.b / d); }     // true divide per channel (color.h:165-168)
FD float intensity(C3 a) { return (a.r + a.g + a.b) / 3; }                  // color.h:79-82

#define FRAY_PI 3.141592653589793238   // constants.h:31
