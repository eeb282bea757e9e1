// Certified "surely misses" filter in front of Triangle::intersectFast (triangle.cpp:66-97, called by Mesh::intersectTriangle,
// mesh.cpp:102-141, for every triangle of a KD leaf, mesh.cpp:360-367).
//
// The reference's test costs ~75 FP64 instructions and a division per triangle, and a ray that walks a KD-tree meets twenty
// triangles per leaf of which it hits one or none.  A triangle the reference REJECTS leaves no trace: `info` is only written on
// acceptance.  So the walk may skip every triangle it can PROVE the reference rejects, and run the reference's arithmetic
// (tri_test, dev_trace.hpp) on the others in their order: the accepted triangles, their order and every number written are the
// reference's.  The proof is an FP32 evaluation of the two barycentric numerators and the determinant with an error bound carried
// along (40 FP32 instructions on a 48-byte record, no division):
//
//     P = d x AC,  det = AB . P,  h = s - A,  u = h . P,  Q = h x AB,  v = d . Q          (lambda2 = u / det, lambda3 = v / det)
//     E = kappa * ((|h|_inf + R) * L + L^2),   L = max(|AB|_inf, |AC|_inf),  R = |A - ref|_inf,  kappa = 2^-14
//
// SURELY REJECTED  <=>  |det| > E  and, with sigma = sign(det),
//     sigma u < -E   or   sigma v < -E   or   sigma u > |det| + E   or   sigma (u + v) > |det| + E.
//
// Why that is safe.  Inputs of the FP32 evaluation are roundings of the reference's own operands: d, AB, AC to FP32 (relative
// 2^-24 each), s - ref and A - ref in FP64 then to FP32, so h is H = s - A with an absolute error <= 2^-22.9 (|H|_inf + R).
// With eps = 2^-24: every component of P carries <= 4 eps of its terms' magnitudes, the dot products add 3 eps, so
//     |u_c - U|, |v_c - V|  <=  eps (42.6 |H| L + 12.9 (|H| + R) L)  <=  2^-18.2 (|H|_inf + R) L,      |det_c - W|  <=  2^-18.6 L^2
// (U = [H, AC, D], V = [AB, H, D], W = [AB, AC, D] in real arithmetic on the FP64 operands, D = -d; the sum u + v adds
// 2^-24 * 12 |H| L).  All of it together is below 2^-17 ((|H|_inf + R) L + L^2) = E / 8.  The reference's own FP64 evaluation
// (det() and dot() of triangle.cpp:27-31, ABcrossAC rounded once when the mesh was loaded) differs from U, V, W by less than
// 2^-49 of the same magnitudes.  Hence, when |det_c| > E: W has the sign of det_c and |W| > 7/8 E, far above the FP64 error of Dcr,
// so rDcr has that sign; sigma u_c < -E gives sigma U < -7/8 E, the reference's numerator has the sign of U, lambda2 < 0: rejected
// at triangle.cpp:84 at the latest (an earlier `return false` -- |Dcr| < 1e-12, gamma -- is as good).  sigma u_c > |det_c| + E gives
// U / W > 1 + 3/4 E / |W|, a relative margin >= 2^-17.4 against FP64 errors of 2^-47: lambda2 > 1.  The same for lambda3, and for
// lambda2 + lambda3 > 1, i.e. lambda1 < 0 (triangle.cpp:91) unless an earlier test already returned false.
// A record whose operands are outside the range where FP32 products neither overflow nor lose their bits to underflow
// (|coordinate - ref| > 2^30, L < 2^-40), or whose ABcrossAC is not the cross product of its AB and AC, carries E = +inf: never
// rejected here.  Products that underflow add at most 20 * 2^-126 (flushed or not) where E >= 2^-94.  A ray is only filtered when |s - ref| <= 1e9 per component.
// PRECONDITION |d|_inf <= 1 (the bound's terms |H| L and L^2 are what products with a direction component of magnitude <= 1 can reach): every
// caller passes a local ray whose direction Node::intersect has normalised (Transform::untransformDir, matrix.cpp:158-161), and the harness does too.  Backface culling (mesh.cpp:106) is left to the reference's
// arithmetic: a culled triangle that is not rejected here costs time, never a wrong answer.
// tests/native/tricert_check.cpp runs this header on the host against the reference's arithmetic over adversarial rays
// (edges, vertices, grazing rays, slivers, far origins, huge and tiny triangles).
#pragma once
#ifndef FRAY_CERT_FN
#define FRAY_CERT_FN __device__ __forceinline__
#endif
#include <stdint.h>

struct alignas(16) DTri32 {   // 48 B, one per leaf reference, beside DMesh::ltris
    float A[3];        // A - ref
    float AB[3], AC[3];
    float Lq;          // kappa * L
    float Cq;          // kappa * (R * L + L^2)
    float pad;
};

#ifndef FRAY_TRICERT_KAPPA
#define FRAY_TRICERT_KAPPA 0x1p-14f
#endif

// host side (frayhip_scene_create, the harness): the record of one triangle.  N = the mesh's stored ABcrossAC.
static inline void tricert_make(DTri32& o, const double* A, const double* AB, const double* AC, const double* N, const double* ref, float kappa = FRAY_TRICERT_KAPPA)
{
    double L = 0, R = 0, big = 0;
    bool ok = true;
    for (int k = 0; k < 3; k++) {
        const double a = A[k] - ref[k];
        o.A[k] = (float)a; o.AB[k] = (float)AB[k]; o.AC[k] = (float)AC[k];
        const double ab = AB[k] < 0 ? -AB[k] : AB[k], ac = AC[k] < 0 ? -AC[k] : AC[k], aa = a < 0 ? -a : a;
        L = L < ab ? ab : L; L = L < ac ? ac : L;
        R = R < aa ? aa : R;
        big = big < aa ? aa : big; big = big < ab ? ab : big; big = big < ac ? ac : big;
        ok = ok && a == a && AB[k] == AB[k] && AC[k] == AC[k];
    }
    // ABcrossAC must be what the reference's loader computes from AB and AC (triangle.h: ABcrossAC = AB ^ AC), up to 2^-40 L^2
    const double cx = AB[1] * AC[2] - AB[2] * AC[1], cy = AB[2] * AC[0] - AB[0] * AC[2], cz = AB[0] * AC[1] - AB[1] * AC[0];
    const double tol = 0x1p-40 * L * L;
    const double ex = cx - N[0], ey = cy - N[1], ez = cz - N[2];
    ok = ok && !(ex > tol || ex < -tol) && !(ey > tol || ey < -tol) && !(ez > tol || ez < -tol) && ex == ex && ey == ey && ez == ez;
    ok = ok && big <= 0x1p30 && L >= 0x1p-40;
    o.pad = 0;
    if (!ok) { o.Lq = __builtin_inff(); o.Cq = __builtin_inff(); return; }
    // rounded UP to FP32 (one ulp of slack each: the bound's slack is a factor 8)
    o.Lq = (float)(kappa * L * (1 + 0x1p-20));
    o.Cq = (float)(kappa * (R * L + L * L) * (1 + 0x1p-20));
}

// true: Triangle::intersectFast returns false for this ray and this triangle, whatever minDist is
FRAY_CERT_FN bool tri_sure_miss(const float ax, const float ay, const float az, const float abx, const float aby, const float abz,
                                const float acx, const float acy, const float acz, const float Lq, const float Cq,
                                const float sx, const float sy, const float sz, const float dx, const float dy, const float dz)
{
    // P = d x AC
    const float px = __builtin_fmaf(dy, acz, -(dz * acy));
    const float py = __builtin_fmaf(dz, acx, -(dx * acz));
    const float pz = __builtin_fmaf(dx, acy, -(dy * acx));
    const float det = __builtin_fmaf(abz, pz, __builtin_fmaf(aby, py, abx * px));
    const float hx = sx - ax, hy = sy - ay, hz = sz - az;
    const float u = __builtin_fmaf(hz, pz, __builtin_fmaf(hy, py, hx * px));
    // Q = h x AB
    const float qx = __builtin_fmaf(hy, abz, -(hz * aby));
    const float qy = __builtin_fmaf(hz, abx, -(hx * abz));
    const float qz = __builtin_fmaf(hx, aby, -(hy * abx));
    const float v = __builtin_fmaf(dz, qz, __builtin_fmaf(dy, qy, dx * qx));
    const float hn = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(hx), __builtin_fabsf(hy)), __builtin_fabsf(hz));
    const float E = __builtin_fmaf(hn, Lq, Cq);
    const float a = __builtin_fabsf(det);
    const float su = __builtin_copysignf(1.0f, det) * u, sv = __builtin_copysignf(1.0f, det) * v;
    const float top = a + E;
    return (a > E) & ((su < -E) | (sv < -E) | (su > top) | (su + sv > top));
}
