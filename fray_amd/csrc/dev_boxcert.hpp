// Certified shortcut for BBox::testIntersect (bbox.h:87-134).
//
// The reference decides "does the ray touch this box" with six face tests (~100 FP64 instructions on a lane), and
// Mesh::intersectKD (mesh.cpp:357-394) asks it twice per inner node.  The boolean has to be the reference's, bit for
// bit (a `false` prunes triangles that a slab test would have reached, and vice versa), but it does not have to be
// COMPUTED the reference's way: here a box is first classified from the ray's parameter interval against it,
//
//     n_k / f_k  = near / far plane parameter of dimension k,  t0 = max n_k,  n2 = second largest n_k,  t1 = min f_k
//
// as SURELY TRUE, SURELY FALSE or UNCERTAIN, with margins that are orders of magnitude wider than every rounding
// error involved; only UNCERTAIN lanes run the reference's arithmetic (box_test, dev_trace.hpp).  Down a KD-tree the
// interval of a child follows from its parent's with one subtraction and one multiplication (BBox::split changes one
// plane), so a walking lane carries {t0, n2, t1} instead of a six-coordinate box.
//
// Why the classification is safe (u = 2^-53; M bounds |start| and every box coordinate; delta = 2^-36 M):
//  * every parameter computed here or by the reference is within 4 u M rmax of its real value (rmax = max |1/dir_k|), and
//    every face hit coordinate the reference compares is within 9 u M of its real value whenever that value lies within
//    M of the box; farther away the comparison is decided by orders of magnitude.  mu = 4 delta rmax exceeds, in
//    parameter units, delta in coordinate units for every dimension, and delta > 10^4 x 9 u M.
//  * SURELY TRUE, clean entry: t0 >= mu (the entry plane lies ahead), n2 + mu <= t0 and t0 + mu <= t1 (the entry point
//    lies inside the entry face's rectangle by more than delta in both coordinates).  The ray really crosses the box, so
//    no "moving away from the slab" test of bbox.h:91 fires (those compare inputs, without rounding); the entry face's
//    dimension starts outside its slab on the near side, so neither `continue` of bbox.h:92,98 skips it (dist to vmin is
//    positive for either sign of dir), and its hit point passes both interval tests.  The reference returns true there or
//    earlier.
//  * SURELY TRUE, inside: t0 <= -mu and t1 >= mu: the start lies strictly inside, bbox.h:89 returns true.
//  * SURELY FALSE: (a) t1 < -A with A = (1e-6 + 4 delta) rmax: one slab lies wholly behind the start by more than inside()'s
//    tolerance: not inside, and bbox.h:91 returns false (or no face passes).  (b) t0 > t1 + 3 mu and t0 > A: not inside (the
//    entry dimension starts more than 1e-6 before its slab); every face plane's hit point lies outside its rectangle by more
//    than delta in one coordinate (for the plane of dimension k at parameter T: T >= f_q + mu for the dimension q of t1, or
//    T <= n_p - mu for the dimension p of t0), so no face test passes and nothing else returns true.
//  * Rays with a direction component below 1e-6 in magnitude (RRay::prepareForTracing's 1e12 stand-in, the |dir| < 1e-9
//    `continue`) or M above 1e9 are never classified.
// tests/native/boxcert_check.cpp runs these functions on the host against the reference's arithmetic over adversarial
// rays (edges, corners, faces, starts inside inside()'s tolerance shell, midpoint-split chains).
#pragma once
#ifndef FRAY_CERT_FN
#define FRAY_CERT_FN __device__ __forceinline__
#endif

struct CertRay {      // per (local ray, box family): the margins
    double mu, A;
    bool ok;
};
struct TState { double t0, n2, t1; };

// rmax = max |rdir_k|, sMax = max |start_k| of the local ray, dirOk: min |dir_k| >= 1e-6; boxMax = max |coordinate| of the family's outermost box
FRAY_CERT_FN CertRay cert_ray(double rmax, double sMax, bool dirOk, double boxMax)
{
    CertRay c;
    const double M = __builtin_fmax(sMax, boxMax);
    const double delta = M * 0x1p-36;
    c.mu = 4.0 * delta * rmax;
    c.A = 1e-6 * rmax + c.mu;
    c.ok = dirOk && M <= 1e9;
    return c;
}

// the ray's interval against a box given by its six planes.  One dimension after the other (FRAY_CERT_SEQ keeps the device compiler from
// computing all six parameters first and holding them in thirty registers: the kernels' occupancy is decided by such peaks).
#ifndef FRAY_CERT_SEQ
#define FRAY_CERT_SEQ() do { } while (0)
#endif
FRAY_CERT_FN TState tstate_box(double lox, double loy, double loz, double hix, double hiy, double hiz, double sx, double sy, double sz, double rx, double ry, double rz)
{
    TState t;
    {
        const double a = (lox - sx) * rx, b = (hix - sx) * rx;
        t.t0 = __builtin_fmin(a, b);
        t.t1 = __builtin_fmax(a, b);
    }
    FRAY_CERT_SEQ();
    {
        const double a = (loy - sy) * ry, b = (hiy - sy) * ry;
        const double n = __builtin_fmin(a, b), f = __builtin_fmax(a, b);
        t.n2 = __builtin_fmin(t.t0, n);
        t.t0 = __builtin_fmax(t.t0, n);
        t.t1 = __builtin_fmin(t.t1, f);
    }
    FRAY_CERT_SEQ();
    {
        const double a = (loz - sz) * rz, b = (hiz - sz) * rz;
        const double n = __builtin_fmin(a, b), f = __builtin_fmax(a, b);
        t.n2 = __builtin_fmax(t.n2, __builtin_fmin(t.t0, n));
        t.t0 = __builtin_fmax(t.t0, n);
        t.t1 = __builtin_fmin(t.t1, f);
    }
    return t;
}

// BBox::split (bbox.h:205-211): the child keeps five planes; `ts` is the parameter of the split plane, `farPlane` says the
// split plane is the child's FAR plane along the axis (lower child of a ray going up the axis, upper child of one going down).
// n2 is kept conservatively (never below the true second-largest near parameter), which can only turn TRUE into UNCERTAIN.
FRAY_CERT_FN TState tstate_child(TState p, double ts, bool farPlane)
{
    TState c;
    const bool ahead = ts >= p.t0;
    const double n2n = ahead ? p.t0 : __builtin_fmax(p.n2, ts);
    c.t0 = farPlane ? p.t0 : __builtin_fmax(p.t0, ts);
    c.n2 = farPlane ? p.n2 : n2n;
    c.t1 = farPlane ? __builtin_fmin(p.t1, ts) : p.t1;
    return c;
}

// yes: surely true; unc: neither surely true nor surely false
FRAY_CERT_FN void cert_decide(TState t, CertRay c, bool& yes, bool& unc)
{
    const bool y = (t.t0 >= c.mu && t.n2 + c.mu <= t.t0 && t.t0 + c.mu <= t.t1) || (t.t0 <= -c.mu && t.t1 >= c.mu);
    const bool n = (t.t0 > t.t1 + 3.0 * c.mu && t.t0 > c.A) || (t.t1 < -c.A);
    yes = y && c.ok;
    unc = !c.ok || !(y || n);
}
// +1 surely true, 0 surely false, -1 uncertain
FRAY_CERT_FN int cert_classify(TState t, CertRay c)
{
    bool yes, unc;
    cert_decide(t, c, yes, unc);
    return unc ? -1 : (yes ? 1 : 0);
}
