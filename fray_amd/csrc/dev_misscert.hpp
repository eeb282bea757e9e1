// "This ray surely misses this box": a certificate under which NO geometry inside the box reports an intersection -- not in real arithmetic
// only, but as the reference's own code computes it (Sphere / Cube::intersect, geometry.cpp:52-137; BBox::testIntersect, bbox.h:87-134,
// which guards every mesh, mesh.cpp:148) -- and therefore no CsgOp over such geometries does (CsgOp::intersect, geometry.cpp:139-194: an operand
// without intersections never changes its in / out state).  The reference's CsgOp has no bounding volume: it runs findAllIntersections on both
// operand trees for every ray of the scene.  With the certificate a ray that passes an object by skips its whole tree, and the answer is the
// reference's: "no intersection".
//
// The box is given by centre c and half extents H; the caller has widened H beyond the geometry's true extents by
//     m = 1e-5 + 2^-22 (|start|_inf + M),      M = max_k (|c_k| + H_k)
// (the constant part on the host, the ray's part here).  The certificate holds when
//   (1) the box lies behind: for some k,  p_k > H_k and d_k >= 0,  or  p_k < -H_k and d_k <= 0      (p = start - c), or
//   (2) the ray's LINE misses the box: for some k, with (u, v) the other two axes,
//           |d_u p_v - d_v p_u|  >  (|d_v| H_u + |d_u| H_v) (1 + 2^-40) + 2^-40 |p|_inf                (separating axis e_k x d).
// Why that is enough (u = 2^-53, |d| <= 1 + 4u):
//   * (2): the left side is |line-to-centre distance| x |(d_u, d_v)| in the projection along e_k, the right side the box's support in that
//     direction x the same factor; the rounding of both sides stays below 6u |p|_inf + 3u R, which the 2^-40 terms cover 1000 times.  So the real
//     line misses the widened box, i.e. passes the true box at a distance of more than m.
//   * Every point the reference computes on the ray -- start + dir * t with t from a division or from (plane - start) * (1 / dir) -- lies within a
//     RELATIVE 8u of the real line's point at that plane (a subtraction, a reciprocal or a division, two multiplications, an addition).  If that real
//     point is within 2 (|start| + M) of the origin the computed one is within 16u (|start| + M) << m of it, hence outside the true box's face by more
//     than m / 2; if it is farther away, no relative 8u brings it back.  So no face test of Cube::intersect (tolerance 1e-6 < m / 2), of
//     BBox::testIntersect (exact <=) passes, and the start is not inside() (outside by > m > 1e-6).  (A tree with a Plane operand gets no box:
//     Plane::intersect reports a hit at NaN for a horizontal ray that starts at the plane's height, geometry.cpp:35-41, wherever the ray is.)
//   * (1): every forward point has coordinate k = start_k + d_k t >= start_k (1 - 2u) beyond the box by m / 2: the same tests fail; BBox::testIntersect
//     returns false at bbox.h:91 for dimension k itself (or skips it when |d_k| < 1e-9) and no earlier dimension's face test passes.
//   * Sphere::intersect decides by Disc = B^2 - 4C, computed within 22u |H|^2 of its real value (H = start - centre, |H| <= |start| + M); a line
//     passing the sphere's box at distance > m has Disc_real <= -4 m^2 < -22u |H|^2 because m >= 2^-22 |H| (2^-44 x 4 = 2.3e-13 > 22u = 2.4e-15);
//     a sphere behind the start has both roots negative by more than m / 2 (their rounding: 4u of magnitudes below 2 |H|).
// tests/native/misscert_check.cpp runs the function on the host against those three routines restated from the reference, over random and adversarial
// rays (grazing faces, edges and corners at distances around m, far starts, axis-parallel directions): no certified ray is ever reported hit; with
// the margin removed the same harness finds contradictions.
#pragma once
#ifndef FRAY_CERT_FN
#define FRAY_CERT_FN __device__ __forceinline__
#endif
#ifndef FRAY_MISSCERT_SCALE
#define FRAY_MISSCERT_SCALE 1.0     // the harness builds a second time with 0 (and without the host's 1e-5) to show that it sees contradictions then
#endif

FRAY_CERT_FN bool ray_surely_misses_box(double cx, double cy, double cz, double hx, double hy, double hz, double M,
                                        double sx, double sy, double sz, double dx, double dy, double dz)
{
    const double px = sx - cx, py = sy - cy, pz = sz - cz;
    const double sm = __builtin_fmax(__builtin_fmax(__builtin_fabs(sx), __builtin_fabs(sy)), __builtin_fabs(sz));
    const double eta = FRAY_MISSCERT_SCALE * 0x1p-22 * (sm + M);
    const double Hx = hx + eta, Hy = hy + eta, Hz = hz + eta;
    bool miss = (px > Hx && dx >= 0) || (px < -Hx && dx <= 0) || (py > Hy && dy >= 0) || (py < -Hy && dy <= 0) || (pz > Hz && dz >= 0) || (pz < -Hz && dz <= 0);
    const double pm = FRAY_MISSCERT_SCALE * 0x1p-40 * __builtin_fmax(__builtin_fmax(__builtin_fabs(px), __builtin_fabs(py)), __builtin_fabs(pz));
    const double g = 1.0 + FRAY_MISSCERT_SCALE * 0x1p-40;
    const double ax = __builtin_fabs(dx), ay = __builtin_fabs(dy), az = __builtin_fabs(dz);
    miss = miss || __builtin_fabs(dy * pz - dz * py) > (az * Hy + ay * Hz) * g + pm;      // axis x: (u, v) = (y, z)
    miss = miss || __builtin_fabs(dx * pz - dz * px) > (az * Hx + ax * Hz) * g + pm;      // axis y: (x, z)
    miss = miss || __builtin_fabs(dx * py - dy * px) > (ay * Hx + ax * Hy) * g + pm;      // axis z: (x, y)
    return miss;
}

// The same certificate evaluated in FP32 -- what the path tracer's PRODUCERS run on every ray they emit, for every gate of the scene (kernels.hpp
// ray_gate_class), so that a consumer may skip a gate's node for the rays filed as gate-free (a front entry of a queue segment) without testing anything.
// The ray is the queue's own (FP64 start and direction, converted here); the box is given in FP32: centre cf = fl32(c) and half extents hf >= half +
// 1e-5 + |c - cf|, rounded up (the host's doing).  e = 2^-24:
//   p~ = fl32(fl32(o) - cf) is within e |o| + e |p~| of o - cf;  d~ = fl32(d) within e |d|;  so
//   |L~ - L| <= 12 e |d|_inf (|o|_inf + |p|_inf)  and the right side's rounding stays below 4 e R:  the slack 2^-19 (omax + pmax) per unit of |d|_1 and the
//   factor 1 + 2^-20 cover them 2.6 and 4 times; the same slack, added to H, covers test (1).  eta = 2^-21 (omax + M) >= the FP64 form's 2^-22 (|o| + M)
//   with room for its own rounding.  `dsum` = |d~x| + |d~y| + |d~z| >= |d|_inf, so a direction that is not a unit vector (a shadow segment b - a) scales
//   both sides alike.  A certified ray is certified by the FP64 form's conditions with margins to spare, hence misses everything in the box.
// The harness runs both forms.
FRAY_CERT_FN bool ray_surely_misses_box_f32(float cx, float cy, float cz, float hx, float hy, float hz, float M,
                                            float ox, float oy, float oz, float dx, float dy, float dz, float omax, float dsum)
{
    const float px = ox - cx, py = oy - cy, pz = oz - cz;
    const float pmax = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(px), __builtin_fabsf(py)), __builtin_fabsf(pz));
    const float slack = (float)FRAY_MISSCERT_SCALE * 0x1p-19f * (omax + pmax);
    const float eta = (float)FRAY_MISSCERT_SCALE * 0x1p-21f * (omax + M) + slack;
    const float Hx = hx + eta, Hy = hy + eta, Hz = hz + eta;
    bool miss = (px > Hx && dx >= 0) || (px < -Hx && dx <= 0) || (py > Hy && dy >= 0) || (py < -Hy && dy <= 0) || (pz > Hz && dz >= 0) || (pz < -Hz && dz <= 0);
    const float g = 1.0f + (float)FRAY_MISSCERT_SCALE * 0x1p-20f, sl = slack * dsum;
    const float ax = __builtin_fabsf(dx), ay = __builtin_fabsf(dy), az = __builtin_fabsf(dz);
    miss = miss || __builtin_fabsf(dy * pz - dz * py) > (az * Hy + ay * Hz) * g + sl;
    miss = miss || __builtin_fabsf(dx * pz - dz * px) > (az * Hx + ax * Hz) * g + sl;
    miss = miss || __builtin_fabsf(dx * py - dy * px) > (ay * Hx + ax * Hy) * g + sl;
    return miss;
}
