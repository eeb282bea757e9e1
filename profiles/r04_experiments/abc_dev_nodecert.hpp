// Certified shortcuts around Mesh::intersect's root box test and Node::intersect's world distance (mesh.cpp:144-165,
// geometry.cpp:196-208, main.cpp:64-80, 178-199).  As with dev_boxcert.hpp / dev_tricert.hpp the OUTCOME is the reference's, bit for bit;
// what changes is that a boolean the reference computes is decided from facts already at hand wherever a proof with wide margins
// allows it, and the reference's own arithmetic runs for the lanes (normally none of a wave) the proof does not cover.
// tests/native/nodecert_check.cpp runs these functions on the host against the reference's arithmetic over adversarial inputs.
//
// u = 2^-53 throughout.
//
// ---- (A) meshes without a KD-tree whose bounding box is FLAT (the Cornell box's floor and right wall) ----------------------------
//
// Mesh::intersect asks bbox.testIntersect(ray) (bbox.h:87-134, ~100-200 FP64 instructions for a box the ray does not start in) before
// its triangle loop and returns false when the box says no.  Both are pure functions of the local ray and a mesh's triangle loop
// starts from info.dist = INF (mesh.cpp:151), so
//        Mesh::intersect = testIntersect && (some triangle accepted)  =  (some triangle accepted) && testIntersect
// and the box test only matters for lanes that found a triangle.  For a box with vmin[f] == vmax[f] (every vertex of the mesh has that
// coordinate, the mesh lies in the plane x_f = c) an accepted triangle PROVES the box test true under the conditions of flat_box_sure():
//   (C1) the accepted triangle's computed barycentrics l1 = 1 - (l2 + l3), l2, l3 are all >= delta = 2^-16;
//   (C2) not |dir_f| < 1e-9 (the reference's own `continue`, bbox.h:92, same operands);
//   (C3) G = max|start_k| + gamma + boxMax <= flatR, where flatR = 2^23 min(hmin, wmin) is made at upload (flat_make): hmin = the
//        smallest altitude of any triangle of the mesh, wmin = the smallest extent of any triangle along either in-plane axis.
//        flat_make also checks what the proof assumes about the records: A_f == c, AB_f == AC_f == 0 exactly, the stored normal within
//        4 ulp of cross(AB, AC), all three corners inside the box, wmin >= 2^-30 boxMax; a mesh that fails gets flatDim = -1.
// Proof.  N = AB x AC has N_u = N_v = +-0 exactly, so Dcr = -N_f d_f (1 + u), and gamma = fl(s_f - c) N_f / Dcr (1 + 3u)^+-1 =
// t* (1 + th), |th| <= 5u, with t* = (c - s_f) / d_f the exact parameter of the plane: no cancellation, the sign of gamma is the sign
// of t* (an accepted triangle has gamma >= 0).  The numerators of l2, l3 are sums of products bounded by |AC| (|H_f| + |H| |d_f|), so
// |l_i - l_i*| <= 6u (t* + |H|) / h <= 6 sqrt(3) u G / hmin <= 2^-27 by (C3): the exact hit point P* = s + t* d has exact barycentrics
// >= delta / 2, hence lies inside the box's rectangle by at least (delta / 2) wmin in both in-plane coordinates (each coordinate of
// P* - lo is a sum of non-negative terms, one of which is l_i* times the triangle's extent).  testIntersect: (1) if inside(start) it
// returns true.  (2) Otherwise no `return false` of bbox.h:91 can fire: for k != f it would put P*_k = s_k + t* d_k, t* >= 0, outside
// the slab, for k = f it would make t* < 0.  (3) When the loop reaches dim f it does not `continue` at :92 by (C2), nor at :98:
// fl(c - s_f) and rdir_f have the same sign (or the difference is +0), so dist >= 0; the face hit x = fl(s_u + fl(d_u dist)) is within
// 6u (|s_u| + t*) <= 12u G of P*_u, and 12u G <= 12u 2^23 wmin < (delta / 2) wmin: both interval tests pass, `return true`.  (4) An
// earlier dimension can only `continue` or `return true`.  Underflow (|s_f - c| < 1e-290): then |t* d| < 1e-280 and s itself is within
// the box by the margins of P*, case (1).  qed.  The margins: delta / 2 = 2^-17 against errors below 2^-27.
//
// ---- (B) the world distance of a node's hit --------------------------------------------------------------------------------------------
//
// Node::intersect returns info.dist = distance(ray.start, T.transformPoint(local ip)) (geometry.cpp:205-206), ~45 FP64 instructions per
// node hit, and the callers only COMPARE it: with the best hit so far (main.cpp:184, strict <, the first node wins ties) or with the
// light sample's distance (main.cpp:74).  For nodes of one transform class (bitwise equal offset, m, invM: the same local ray) the local
// ip is ls + ld t with t the geometry's own parameter (plane: scaling, sphere: dist, triangle: gamma), and
//        F(t) := fl(distance(o, (ls + ld t) m + offset)) = s_c t + e,   s_c = |ld m|,   |e| <= E(t) = EA (|o|_1 + |offset|_1) + EB |t|
// where EA covers |o - offset - ls m| (the residual of invM, I - invM m, measured at upload in long double, plus the rounding of
// ls) and EA, EB the ~16 roundings of F, with a safety factor 2^8 (class_make).  Hence
//   * two hits of one class are ordered by t whenever |t_a - t_b| > 2 E(max t) / sLo, sLo <= s_c (class_order_margin); nearer than that
//     both distances are computed and compared as the reference does;
//   * a hit is closer than maxDist when sHi t + E < maxDist and not closer when sLo t - E >= maxDist (sLo, sHi: the extreme singular
//     values of m from a Jacobi sweep in long double, checked by residual and widened by 1e-9; a rigid transform has sHi / sLo - 1 = 2e-9),
//     else the distance is computed.
// Classes whose m is singular or whose invM is not its inverse to 1e-6, and parameters above 1e30, are never decided (ok = 0).
#pragma once
#ifndef FRAY_CERT_FN
#define FRAY_CERT_FN __device__ __forceinline__
#endif
#include <stdint.h>

#define FRAY_FLAT_DELTA 0x1p-16

// per transform class (DScene::classes, indexed by DNode::xfClass)
struct DClass {
    double EA, EB;      // E(t) = EA (|o|_1 + off1) + EB |t|, in world units
    double off1;        // |offset|_1
    double sLo, sHi;    // bounds of |ld m| for a unit local direction
    double r2;          // 2 / sLo: E -> margin between two parameters
    int32_t ok, pad;
};

// (A): sAbsMax = max |start_k| of the local ray, df = its direction along the flat axis, {l2, l3, gamma} of the accepted triangle
FRAY_CERT_FN bool flat_box_sure(double l2, double l3, double gamma, double sAbsMax, double df, double flatR)
{
    const double l1 = 1 - (l2 + l3);
    return l2 >= FRAY_FLAT_DELTA && l3 >= FRAY_FLAT_DELTA && l1 >= FRAY_FLAT_DELTA && !(__builtin_fabs(df) < 1e-9) && sAbsMax + gamma <= flatR;
}

// (B)
FRAY_CERT_FN double class_err(const DClass& C, double W, double t) { return C.EA * (W + C.off1) + C.EB * __builtin_fabs(t); }
// parameters farther apart than this are ordered like their world distances (valid only if C.ok and both are <= 1e30)
FRAY_CERT_FN double class_order_margin(const DClass& C, double W, double ta, double tb)
{
    return class_err(C, W, __builtin_fmax(__builtin_fabs(ta), __builtin_fabs(tb))) * C.r2;
}

#include <cmath>
#include <cstring>
// ---- upload-time constants (host) --------------------------------------------------------------------------------------------------------
// Tri records as the device holds them: A, AB, AC, N = stored AB x AC (dev_scene.hpp DTri).  Returns the flat axis (or -1) and flatR.
template <class TRI>
static inline int flat_make(const TRI* tris, int nTris, const double bmin[3], const double bmax[3], double boxMax, double& flatR)
{
    flatR = -1;
    int f = -1;
    for (int k = 0; k < 3; k++)
        if (bmin[k] == bmax[k]) { if (f >= 0) return -1; f = k; }       // exactly one flat dimension
    if (f < 0 || nTris <= 0 || !(boxMax <= 1e9)) return -1;
    const int u = f == 0 ? 1 : 0, v = f == 2 ? 1 : 2;
    const double c = bmin[f];
    double hmin = 1e300, wmin = 1e300;
    for (int i = 0; i < nTris; i++) {
        const double* A = tris[i].A; const double* AB = tris[i].AB; const double* AC = tris[i].AC; const double* N = tris[i].N;
        if (A[f] != c || AB[f] != 0 || AC[f] != 0) return -1;
        if (N[u] != 0 || N[v] != 0) return -1;
        const double nf = f == 0 ? AB[1] * AC[2] - AB[2] * AC[1] : (f == 1 ? AB[2] * AC[0] - AB[0] * AC[2] : AB[0] * AC[1] - AB[1] * AC[0]);
        if (!(std::fabs(N[f] - nf) <= 4 * 0x1p-52 * std::fabs(nf)) || nf == 0) return -1;
        const double B[2] = {A[u] + AB[u], A[v] + AB[v]}, C[2] = {A[u] + AC[u], A[v] + AC[v]}, A2[2] = {A[u], A[v]};
        const double lo2[2] = {bmin[u], bmin[v]}, hi2[2] = {bmax[u], bmax[v]};
        for (int q = 0; q < 2; q++) {
            const double tol = 4 * 0x1p-52 * boxMax;
            const double mn = std::fmin(A2[q], std::fmin(B[q], C[q])), mx = std::fmax(A2[q], std::fmax(B[q], C[q]));
            if (mn < lo2[q] - tol || mx > hi2[q] + tol) return -1;
            wmin = std::fmin(wmin, mx - mn);
        }
        const double BC[2] = {C[0] - B[0], C[1] - B[1]};
        const double lab = std::hypot(AB[u], AB[v]), lac = std::hypot(AC[u], AC[v]), lbc = std::hypot(BC[0], BC[1]);
        const double longest = std::fmax(lab, std::fmax(lac, lbc));
        if (!(longest > 0)) return -1;
        hmin = std::fmin(hmin, std::fabs(nf) / longest);
    }
    if (!(wmin >= 0x1p-30 * boxMax) || !(hmin > 0) || !(wmin > 0)) return -1;
    flatR = std::fmin(hmin, wmin) * 0x1p23 - boxMax;
    if (!(flatR > 0)) { flatR = -1; return -1; }
    return f;
}

static inline void class_make(const double off[3], const double m[9], const double inv[9], DClass& C)
{
    typedef long double L;
    std::memset(&C, 0, sizeof C);
    C.off1 = std::fabs(off[0]) + std::fabs(off[1]) + std::fabs(off[2]);
    C.EA = C.EB = 1e300; C.sLo = 0; C.sHi = 1e300; C.r2 = 1e300; C.ok = 0;
    L nm = 0, ni = 0;
    for (int i = 0; i < 9; i++) {
        if (!std::isfinite(m[i]) || !std::isfinite(inv[i])) return;
        nm = std::fmax(nm, (L)std::fabs(m[i])); ni = std::fmax(ni, (L)std::fabs(inv[i]));
    }
    if (!std::isfinite(C.off1) || C.off1 > 1e9 || nm == 0 || ni == 0 || nm > 1e9 || ni > 1e9) return;
    nm *= 3; ni *= 3;                                   // row-sum bounds
    // rho = max row sum of |I - inv m| (row-vector convention: v inv m should be v)
    L rho = 0;
    for (int i = 0; i < 3; i++) {
        L row = 0;
        for (int j = 0; j < 3; j++) {
            L s = 0;
            for (int k = 0; k < 3; k++) s += (L)inv[3 * i + k] * (L)m[3 * k + j];
            row += fabsl(s - (i == j ? 1 : 0));
        }
        rho = fmaxl(rho, row);
    }
    if (!(rho <= 1e-6L)) return;
    // singular values of m: Jacobi on G = m^T m
    L G[3][3], Vv[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { L s = 0; for (int k = 0; k < 3; k++) s += (L)m[3 * k + i] * (L)m[3 * k + j]; G[i][j] = s; }
    L G0[3][3]; std::memcpy(G0, G, sizeof G);
    for (int sweep = 0; sweep < 60; sweep++) {
        L offd = fabsl(G[0][1]) + fabsl(G[0][2]) + fabsl(G[1][2]);
        if (offd <= 1e-30L * (fabsl(G[0][0]) + fabsl(G[1][1]) + fabsl(G[2][2]))) break;
        for (int p = 0; p < 2; p++) for (int q = p + 1; q < 3; q++) {
            if (G[p][q] == 0) continue;
            const L th = (G[q][q] - G[p][p]) / (2 * G[p][q]);
            const L t = (th >= 0 ? 1 : -1) / (fabsl(th) + sqrtl(th * th + 1));
            const L cs = 1 / sqrtl(t * t + 1), sn = t * cs;
            for (int k = 0; k < 3; k++) { const L a = G[k][p], b = G[k][q]; G[k][p] = cs * a - sn * b; G[k][q] = sn * a + cs * b; }
            for (int k = 0; k < 3; k++) { const L a = G[p][k], b = G[q][k]; G[p][k] = cs * a - sn * b; G[q][k] = sn * a + cs * b; }
            for (int k = 0; k < 3; k++) { const L a = Vv[k][p], b = Vv[k][q]; Vv[k][p] = cs * a - sn * b; Vv[k][q] = sn * a + cs * b; }
        }
    }
    L lmin = 1e300L, lmax = 0, tr = fabsl(G0[0][0]) + fabsl(G0[1][1]) + fabsl(G0[2][2]);
    for (int e = 0; e < 3; e++) {
        const L lam = G[e][e];
        // residual |G0 v - lam v|: the computed pair is within it of a true eigenvalue
        L res = 0, nv = 0;
        for (int i = 0; i < 3; i++) { L s = 0; for (int k = 0; k < 3; k++) s += G0[i][k] * Vv[k][e]; res += fabsl(s - lam * Vv[i][e]); nv += fabsl(Vv[i][e]); }
        if (!(res <= 1e-12L * tr) || !(nv > 0.5L)) return;
        lmin = fminl(lmin, lam); lmax = fmaxl(lmax, lam);
    }
    if (!(lmin > 1e-18L * lmax) || !(lmin > 0)) return;
    const L sLo = sqrtl(lmin) * (1 - 1e-9L), sHi = sqrtl(lmax) * (1 + 1e-9L);
    const L u = 0x1p-53L;
    // |o - offset - ls m| <= (|o|_1 + |offset|_1) (rho + 16 u ni nm); F's own roundings: 16 u nm (|ls|_1 + 2 |t|) + 4 u |offset|_1 + 4 u dist,
    // |ls|_1 <= 3 ni (|o|_1 + |offset|_1)
    const L ea = rho + 16 * u * ni * nm + 48 * u * nm * ni + 4 * u + 4 * u * sHi * 3 * ni;
    const L eb = 32 * u * nm + 8 * u * sHi;
    C.EA = (double)(256 * ea); C.EB = (double)(256 * eb);
    C.sLo = (double)sLo; C.sHi = (double)sHi;
    C.r2 = (double)(2 / sLo * (1 + 1e-9L));
    C.ok = 1;
}
