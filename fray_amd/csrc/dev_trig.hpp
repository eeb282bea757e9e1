// sin / cos / acos for the path tracer's direction sampling (hemisphereSample, main.cpp:92-116; unitDiscSample,
// random_generator.cpp:71-80), CORRECTLY ROUNDED in all but about one call in 10^4.
//
// Why not the device library's functions.  The reference takes sin / cos / acos from glibc; ROCm's (ocml) differ from glibc
// in the last place in 3 % (sin, cos) and 7 % (acos) of calls, so one spawned direction in six had a different last bit than
// the CPU's, and a path that grazes an edge could take the other branch.  glibc's own results are the
// correctly rounded ones in 99.85 % (sin, cos) and 99.93 % (acos) of calls (measured against 113-bit arithmetic,
// tests/native/trig_check.cpp), so functions that round correctly agree with glibc in all but those.
//
// How.  x = m * pi/128 + t with |t| <= pi/256 (three-part pi/128, t as a double-double); sin / cos of m * pi/128 come from a
// 65-entry double-double table (quadrant symmetry for the rest); sin t and cos t - 1 are short polynomials whose leading
// terms are kept as double-doubles; the products with the table values are exact (fma) and summed with error-free
// transformations.  The result's error before the final rounding is below 2^-68.  acos(v) is one Newton step on that
// cosine from the device library's acos.  No operation here relies on contraction: every fused multiply-add is written out.
//
// FRAY_TRIG_FN / FRAY_TRIG_TABLE let a host test compile this header with g++ (tests/native/trig_check.cpp).
#pragma once
#ifndef FRAY_TRIG_FN
#define FRAY_TRIG_FN __device__ __forceinline__
#define FRAY_TRIG_TABLE static __device__ const
#define FRAY_TRIG_LIBM_SINCOS(x, s, c) sincos(x, s, c)
#define FRAY_TRIG_LIBM_ACOS(x) acos(x)
#endif
#include "dev_trig_table.hpp"

struct DD { double h, l; };

FRAY_TRIG_FN DD dd_two_sum(double a, double b)
{
    const double s = a + b, bb = s - a;
    return DD{s, (a - (s - bb)) + (b - bb)};
}

// sin(x) and cos(x) as unevaluated sums h + l, for 0 <= x <= 6.2918 (2 pi and a little)
FRAY_TRIG_FN void fray_sincos_dd(double x, DD& sn, DD& cs)
{
    const double mf = __builtin_rint(x * FRAY_128OPI);
    const int m = (int)mf;
    // t = x - m * pi/128 as th + tl
    const double a = __builtin_fma(-mf, FRAY_PIO128_1, x);            // exact: m * P1 has at most 42 significant bits
    const double b = mf * FRAY_PIO128_2;
    const double be = __builtin_fma(mf, FRAY_PIO128_2, -b);           // b + be = m * P2 exactly
    DD t = dd_two_sum(a, -b);
    t.l -= be + mf * FRAY_PIO128_3;
    const double th = t.h, tl = t.l;
    // sin t = t + p_s,  cos t - 1 = (c1 + c1l) + q_c
    const double u = th * th;
    const double ul = __builtin_fma(th, th, -u) + 2.0 * th * tl;
    const double ps = th * u * (-0x1.5555555555555p-3 + u * (0x1.1111111111111p-7 + u * -0x1.a01a01a01a01ap-13));      // -1/6, 1/120, -1/5040
    const double sh = th, sl = tl + ps;
    const double c1 = -0.5 * u, c1l = -0.5 * ul;
    const double qc = u * u * (0x1.5555555555555p-5 + u * (-0x1.6c16c16c16c17p-10 + u * 0x1.a01a01a01a01ap-16));          // 1/24, -1/720, 1/40320
    const double ch = c1, cl = c1l + qc;
    // table: angle (m mod 64) * pi/128 in quadrant m / 64
    const int q = (m >> 6) & 3, j = m & 63;
    double Sh = kTrigTable[j][0], Sl = kTrigTable[j][1], Ch = kTrigTable[j][2], Cl = kTrigTable[j][3];
    if (q & 1) { const double a0 = Sh, a1 = Sl; Sh = Ch; Sl = Cl; Ch = -a0; Cl = -a1; }      // sin(pi/2 + p) = cos p, cos(pi/2 + p) = -sin p
    if (q & 2) { Sh = -Sh; Sl = -Sl; Ch = -Ch; Cl = -Cl; }
    // sin x = S (1 + (cos t - 1)) + C sin t ;  cos x = C (1 + (cos t - 1)) - S sin t
    {
        const double pa = Sh * ch, pal = __builtin_fma(Sh, ch, -pa) + Sh * cl;
        const double pb = Ch * sh, pbl = __builtin_fma(Ch, sh, -pb) + (Ch * sl + Cl * sh);
        const DD x1 = dd_two_sum(pb, pa);
        const DD r = dd_two_sum(Sh, x1.h);
        sn.h = r.h;
        sn.l = r.l + (x1.l + (Sl + (pal + pbl)));
    }
    {
        const double pa = Ch * ch, pal = __builtin_fma(Ch, ch, -pa) + Ch * cl;
        const double pb = -Sh * sh, pbl = __builtin_fma(-Sh, sh, -pb) - (Sh * sl + Sl * sh);
        const DD x1 = dd_two_sum(pb, pa);
        const DD r = dd_two_sum(Ch, x1.h);
        cs.h = r.h;
        cs.l = r.l + (x1.l + (Cl + (pal + pbl)));
    }
}

#if defined(FRAY_ARITH) && FRAY_ARITH && defined(__HIP_DEVICE_COMPILE__)
// The kernels for rays after a sample's first closest hit under option "fp_contract" (render_contract.hip): their directions need no correct rounding --
// a last place of a sampled direction is 1e-16 of a colour -- so the same reduction and table run in plain double (an ulp or two), a sixth of the instructions.
FRAY_TRIG_FN void fray_sincos(double x, double* s, double* c)
{
    const double mf = __builtin_rint(x * FRAY_128OPI);
    const int m = (int)mf;
    const double t = __builtin_fma(-mf, FRAY_PIO128_2, __builtin_fma(-mf, FRAY_PIO128_1, x));
    const double u = t * t;
    const double st = __builtin_fma(t * u, -0x1.5555555555555p-3 + u * (0x1.1111111111111p-7 + u * -0x1.a01a01a01a01ap-13), t);      // sin t
    const double cm = u * (-0.5 + u * (0x1.5555555555555p-5 + u * -0x1.6c16c16c16c17p-10));                                         // cos t - 1
    const int q = (m >> 6) & 3, j = m & 63;
    double S = kTrigTable[j][0], C = kTrigTable[j][2];
    if (q & 1) { const double a0 = S; S = C; C = -a0; }
    if (q & 2) { S = -S; C = -C; }
    *s = S + __builtin_fma(S, cm, C * st);
    *c = C + __builtin_fma(C, cm, -S * st);
}
// sin and cos of acos(v): v itself and sqrt(1 - v^2)
FRAY_TRIG_FN void fray_acos_sincos(double v, double* s, double* c)
{
    *c = v;
    *s = fray_sqrt(__builtin_fma(-v, v, 1.0));
}
#define FRAY_TRIG_RELAXED 1
#endif

#ifndef FRAY_TRIG_RELAXED
FRAY_TRIG_FN void fray_sincos(double x, double* s, double* c)
{
    if (!(x >= 0.0 && x <= 6.2918)) { FRAY_TRIG_LIBM_SINCOS(x, s, c); return; }      // outside the sampler's range (never in a render): the library's
    DD sn, cs;
    fray_sincos_dd(x, sn, cs);
    *s = sn.h + sn.l;
    *c = cs.h + cs.l;
}

FRAY_TRIG_FN double fray_acos(double v)
{
    const double y0 = FRAY_TRIG_LIBM_ACOS(v);                     // within an ulp; NaN outside [-1, 1]
    if (!(y0 > 0x1p-20 && y0 < 3.1415916)) return y0;           // at the ends cos is flat: the step below would lose its accuracy (|v| > 1 - 5e-13)
    DD sn, cs;
    fray_sincos_dd(y0, sn, cs);
    // cos(y0 + d) = v  =>  d = (cos y0 - v) / sin y0
    const double d = ((cs.h - v) + cs.l) / sn.h;
    return y0 + d;
}

// sin and cos of acos(v) -- what fray_sincos(fray_acos(v), ...) returns, without evaluating the table twice: the sine and
// cosine of the Newton step's starting point are carried over the step (a last place at most) to first order.
FRAY_TRIG_FN void fray_acos_sincos(double v, double* s, double* c)
{
    const double y0 = FRAY_TRIG_LIBM_ACOS(v);
    if (!(y0 > 0x1p-20 && y0 < 3.1415916)) { fray_sincos(y0, s, c); return; }
    DD sn, cs;
    fray_sincos_dd(y0, sn, cs);
    const double y = y0 + ((cs.h - v) + cs.l) / sn.h;         // = fray_acos(v)
    const double dy = y - y0;                                  // exact: zero or a unit in the last place
    *s = sn.h + (sn.l + cs.h * dy);                            // sin(y0 + dy) = sin y0 + cos y0 dy  (dy^2 ~ 2^-104)
    *c = cs.h + (cs.l - sn.h * dy);
}
#endif
