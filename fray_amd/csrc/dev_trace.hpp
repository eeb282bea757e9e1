// Ray / scene intersection on the device: the closest-hit loop over nodes and lights
// (main.cpp:178-199, 250-271), visible() (main.cpp:64-80), Node::intersect (geometry.cpp:196-208),
// Plane / Sphere (geometry.cpp:30-83), RectLight::intersect (lights.cpp:79-103) and
// Mesh::intersect with its KD-tree (mesh.cpp:144-165, 357-394).
//
// The KD walk (mesh_intersect) is a while-while loop over an explicit per-lane stack of pending far children (KdStack: the 16 most recent
// entries in LDS, older ones in scratch); the reference recurses (depth reaches 65 on teapot_hires) and carries the child boxes on its call
// stack.  Here a lane carries the ray's parameter interval against the current box ({t0, n2, t1}, dev_boxcert.hpp) instead of the box: a
// child's interval follows from its parent's with one multiplication, and it decides BBox::testIntersect for the child outright except
// within margins of an edge, where the lane reads the node's box and runs the reference's arithmetic.  Both children are tested when a node
// is entered (the tests are pure functions of the ray); "which child first" is ray.start[axis] < split, so the visiting order, the outcome
// of every box test, the triangles tested and the first accepted leaf are exactly the reference's.
#pragma once
#ifndef FRAY_IDENTITY_SKIP
#define FRAY_IDENTITY_SKIP 1       // node_intersect: untransformed nodes skip v * I (round 5)
#endif
#ifndef FRAY_FILTER_STRAIGHT
#define FRAY_FILTER_STRAIGHT 1     // see mesh_intersect's leaf loop
#endif
#include "dev_math.hpp"
#include "dev_scene.hpp"
#include "dev_sort.hpp"
#define FRAY_CERT_SEQ() __builtin_amdgcn_sched_barrier(0)
#include "dev_boxcert.hpp"
#include "dev_misscert.hpp"

#ifdef FRAY_LEAFSTAT
static __device__ unsigned long long g_leafStat[4];      // diagnostic build only, read back by render_impl (FRAY_LEAFSTAT)
#endif
// Per-lane work counters (frayhip_stats); only the <true> instantiations touch them.
struct Cnt {
    unsigned long long closest, shadow, node, kdInner, leafRefs, tri, prim, smooth, samples, tex;
    unsigned envelope;   // always maintained: a CSG operand produced more hits than FRAY_CSG_MAX
};
// Template flag word of the trace / shade code: bit 0 = maintain the work counters, bit 1 = the scene
// has Cube / CSG geometry (kept out of the common kernels: its hit lists live in scratch memory).
template <int ST> FD void bump(unsigned long long& c, unsigned long long n = 1) { if (ST & 1) c += n; }
// Kernel flag word: bit 0 = maintain the work counters, bit 1 = the scene has Cube / CSG geometry, bit 2 = it has meshes with a KD-tree,
// bit 3 = it has textures or an environment map.
// The KD walk (and its 40-odd registers) is compiled only into the variants that need it: bit 2, or bit 1 (a CSG operand may be a KD mesh).
constexpr bool kd_variant(int st) { return (st & 6) != 0; }
// bit 3 = the scene has textures (bitmap, checker, Fresnel, bump) or a loaded environment map; the variants without it (and without bits 1 / 2,
// which imply it) are compiled without texture sampling, bump mapping, the cubemap lookup and the sphere's uv (atan2 / asin)
constexpr bool tex_variant(int st) { return (st & 14) != 0; }
// the variants whose node_intersect takes the shortcut for untransformed nodes (flag words 0 and 1)
constexpr bool identity_variant(int st) { return FRAY_IDENTITY_SKIP && (st & 14) == 0; }

// Closest-hit record.  Shading attributes (ip, normal, uv, dNdx/dNdy) are re-derived from it for
// the winning node only (finalize_hit in dev_shade.hpp) -- same arithmetic, so same bits.
struct HitRec {
    int node;       // node index; -1 miss; -2-i light i
    int tri;        // winning triangle (meshes)
    double dist;    // world distance (Node::intersect's recomputed dist)
    double t;       // local ray parameter: plane `scaling`, sphere `dist`, triangle gamma
    double l2, l3;  // barycentrics (meshes)
};
// The Cube / CSG kernel variants (<ST & 2>) carry more: the winning intersection as its geometry reported it -- the local hit point (not
// ls + ld t for these two) and, for a CsgOp, the plain geometry at the bottom of the tree that produced it -- so that finalize_hit does not
// have to run CsgOp::intersect a second time to learn them.  The other variants keep the six-field record (two more fields in it measured
// +5 % on the wavefront's shade kernel: 6 more spilled VGPRs).
struct HitRecX : HitRec {
    V3 ipl;
    int leafKind, leafIndex;
};
template <int ST> struct HitOf { typedef HitRec type; };
template <> struct HitOf<2> { typedef HitRecX type; };
template <> struct HitOf<3> { typedef HitRecX type; };
template <int ST> using HitT = typename HitOf<ST>::type;

struct Box6 { double lox, loy, loz, hix, hiy, hiz; };

// Local ray of a node (Transform::untransformPoint / untransformDir, matrix.cpp:148-161), cached per
// transform class: nodes whose {offset, invM} are bitwise identical yield the same local ray.
// rmax / sMax / dirOk: what the certified box test's margins are made of (cert_ray, dev_boxcert.hpp), functions of the local ray too.
struct LocalRay { V3 s, d, rd; int cls; bool haveRd, dirOk; double rmax, sMax; };


FD bool box_inside(const Box6& b, V3 v)   // BBox::inside, bbox.h:79-84
{
    return b.lox - 1e-6 <= v.x && v.x <= b.hix + 1e-6 && b.loy - 1e-6 <= v.y && v.y <= b.hiy + 1e-6 &&
           b.loz - 1e-6 <= v.z && v.z <= b.hiz + 1e-6;
}

// BBox::testIntersect (bbox.h:87-134), written as straight-line predicated code: on a 64-lane wave
// every early `return` / `continue` of the reference is a divergent branch (exec-mask save /
// restore + scalar branch) that costs more than the dozen FP64 operations it skips, and the wave
// executes the union of its lanes' paths anyway.  Each face test below evaluates exactly the
// expressions the reference evaluates (same operands, same order); only the control flow around
// them is replaced by boolean algebra with the same truth table:
//   rej  : the dimension's "moving away from the slab" test  -> the reference returns false
//   skip : |dir| < 1e-9                                      -> `continue`
//   near : dist to the vmin face not < 0 (else `continue`: the far face is skipped too)
//   hitN / hitF : the face's hit point lies inside the other two extents (<= on both sides)
struct BoxDim { bool rej, hit; };
FD BoxDim box_dim(double sd, double dd, double rd, double lo, double hi, double su, double du, double lou, double hiu,
                  double sv, double dv, double lov, double hiv)
{
    BoxDim r;
    r.rej = (dd < 0 && sd < lo) | (dd > 0 && sd > hi);
    const bool skip = fabs(dd) < 1e-9;
    const double d1 = (lo - sd) * rd;
    const bool near = !(d1 < 0);
    const double x1 = su + du * d1, y1 = sv + dv * d1;
    const bool hitN = (lou <= x1) & (x1 <= hiu) & (lov <= y1) & (y1 <= hiv);
    const double d2 = (hi - sd) * rd;
    const bool far = !(d2 < 0);
    const double x2 = su + du * d2, y2 = sv + dv * d2;
    const bool hitF = far & (lou <= x2) & (x2 <= hiu) & (lov <= y2) & (y2 <= hiv);
    r.hit = !r.rej & !skip & near & (hitN | hitF);
    return r;
}

// `be`: the box widened by inside()'s tolerance, i.e. {lo - 1e-6, hi + 1e-6} as bbox.h:81-83 computes them -- the same six FP64 operations
// whoever performs them, so a caller that has them precomputed (the tree-less meshes' node records) passes them in
// The form the scenes WITHOUT KD meshes use for their few boxes (cornell_box: seven per ray, a quarter of its any-hit kernel): it leaves as soon as every
// lane is known to start inside (a room-sized box: the walls' meshes), headline -1.4 %.  The KD variants keep box_test below untouched: any
// change to their code measured +1..3 % on boxed / forest / dragon.
FD bool box_test_pre(const Box6& b, const Box6& be, V3 s, V3 d, V3 rd)
{
    bool res = be.lox <= s.x && s.x <= be.hix && be.loy <= s.y && s.y <= be.hiy && be.loz <= s.z && s.z <= be.hiz;      // BBox::inside
    bool alive = !res;
    if (!__any(alive)) return res;           // wave-uniform
    // dim 0: u = 1 (y), v = 2 (z);  dim 1: u = 0 (x), v = 2 (z);  dim 2: u = 0 (x), v = 1 (y)
    BoxDim a = box_dim(s.x, d.x, rd.x, b.lox, b.hix, s.y, d.y, b.loy, b.hiy, s.z, d.z, b.loz, b.hiz);
    res |= alive & a.hit;
    alive &= !a.rej & !a.hit;
    if (!__any(alive)) return res;          // wave-uniform: every lane is decided
    a = box_dim(s.y, d.y, rd.y, b.loy, b.hiy, s.x, d.x, b.lox, b.hix, s.z, d.z, b.loz, b.hiz);
    res |= alive & a.hit;
    alive &= !a.rej & !a.hit;
    if (!__any(alive)) return res;
    a = box_dim(s.z, d.z, rd.z, b.loz, b.hiz, s.x, d.x, b.lox, b.hix, s.y, d.y, b.loy, b.hiy);
    res |= alive & a.hit;
    return res;
}

FD bool box_test(const Box6& b, V3 s, V3 d, V3 rd)
{
    bool res = box_inside(b, s);
    bool alive = !res;
    // dim 0: u = 1 (y), v = 2 (z);  dim 1: u = 0 (x), v = 2 (z);  dim 2: u = 0 (x), v = 1 (y)
    BoxDim a = box_dim(s.x, d.x, rd.x, b.lox, b.hix, s.y, d.y, b.loy, b.hiy, s.z, d.z, b.loz, b.hiz);
    res |= alive & a.hit;
    alive &= !a.rej & !a.hit;
    if (!__any(alive)) return res;          // wave-uniform: every lane is decided
    a = box_dim(s.y, d.y, rd.y, b.loy, b.hiy, s.x, d.x, b.lox, b.hix, s.z, d.z, b.loz, b.hiz);
    res |= alive & a.hit;
    alive &= !a.rej & !a.hit;
    if (!__any(alive)) return res;
    a = box_dim(s.z, d.z, rd.z, b.loz, b.hiz, s.x, d.x, b.lox, b.hix, s.y, d.y, b.loy, b.hiy);
    res |= alive & a.hit;
    return res;
}

// pure selects: a store through a variable axis would push the whole box into scratch memory
FD void box_set_hi(Box6& b, int axis, double v) { b.hix = axis == 0 ? v : b.hix; b.hiy = axis == 1 ? v : b.hiy; b.hiz = axis == 2 ? v : b.hiz; }
FD void box_set_lo(Box6& b, int axis, double v) { b.lox = axis == 0 ? v : b.lox; b.loy = axis == 1 ? v : b.loy; b.loz = axis == 2 ? v : b.loz; }

// Mesh::intersectTriangle + Triangle::intersectFast (mesh.cpp:102-141, triangle.cpp:66-94),
// test part only.  `best` is info.dist: accepted when gamma <= best, so the LAST equal-distance
// triangle in visiting order wins, as in the reference.
template <int ST>
FD bool tri_test(const FRAY_RO DTri* T, int culling, V3 s, V3 d, double& best, double& l2o, double& l3o, Cnt& c)
{
    bump<ST>(c.tri);
    // the whole 120-byte record is fetched up front: one memory round trip per triangle instead
    // of one per early-out stage (the walk is latency-bound, not bandwidth-bound)
    const V3 g = ld3(T->g), N = ld3(T->N), A = ld3(T->A), AC = ld3(T->AC), AB = ld3(T->AB);
    // straight-line predicated form (see box_dim): every value is the reference's, the early returns are folded into `ok`.
    // `ok` is kept as a 64-bit LANE MASK in scalar registers (lanes(): one v_cmp per comparison, combined with s_and), so
    // "does any lane still need the next stage" is a scalar compare with zero and the lane's own answer is read back for
    // free (lane_of); written on a bool, every such vote goes through a 0/1 vector register and a vector compare.  A mask
    // only ever has bits of active lanes.  (The box test stays on bools: the same rewrite measured 2 % slower there.)
    lanes_t ok = lanes(!(dot(d, g) > 0));
    if (!culling) ok = lanes(true);
    V3 D = -d;
    double Dcr = dot(N, D);
    ok &= lanes(!(fabs(Dcr) < 1e-12));
    if (!ok) return false;                  // wave-uniform: every lane culled this triangle
    double rDcr = fray_rcp(Dcr);
    V3 H = s - A;
    double gamma = dot(N, H) * rDcr;
    ok &= lanes(!(gamma < 0)) & lanes(!(gamma > best));
    if (!ok) return false;                  // wave-uniform: nobody needs the barycentrics
    double l2 = dot(cross(H, AC), D) * rDcr;
    ok &= lanes(!(l2 < 0)) & lanes(!(l2 > 1));
    double l3 = dot(cross(AB, H), D) * rDcr;
    ok &= lanes(!(l3 < 0)) & lanes(!(l3 > 1));
    double l1 = 1 - (l2 + l3);
    ok &= lanes(!(l1 < 0));
    const bool hit = lane_of(ok);
    if (hit) {
        best = gamma;
        l2o = l2;
        l3o = l3;
    }
    return hit;
}

// Mesh::intersectTriangle's barycentrics for a triangle already known to be the winner (the hit queue carries
// {node, triangle, gamma}; l2 / l3 are only read by smooth or textured meshes): tri_test's expressions again.
FD void tri_bary(const FRAY_RO DTri* T, V3 s, V3 d, double& l2, double& l3)
{
    const V3 N = ld3(T->N), A = ld3(T->A), AC = ld3(T->AC), AB = ld3(T->AB);
    const V3 D = -d;
    const double rDcr = fray_rcp(dot(N, D));
    const V3 H = s - A;
    l2 = dot(cross(H, AC), D) * rDcr;
    l3 = dot(cross(AB, H), D) * rDcr;
}

// RRay::prepareForTracing (bbox.h:49-54) of a local ray -- a function of the local ray only, computed once per transform class --
// and the ray's part of the certified box test's margins.
FD V3 ray_rdir(V3 d)
{
    V3 rd;
    rd.x = fabs(d.x) > 1e-12 ? fray_rcp(d.x) : 1e12;
    rd.y = fabs(d.y) > 1e-12 ? fray_rcp(d.y) : 1e12;
    rd.z = fabs(d.z) > 1e-12 ? fray_rcp(d.z) : 1e12;
    return rd;
}
template <int ST>
FD void prepare_ray(LocalRay& lr)
{
    if (lr.haveRd) return;
    lr.rd = ray_rdir(lr.d);
    lr.haveRd = true;
    if (!kd_variant(ST)) return;          // scenes without KD meshes test their few boxes with the reference's arithmetic (geom_intersect)
    lr.rmax = __builtin_fmax(__builtin_fmax(fabs(lr.rd.x), fabs(lr.rd.y)), fabs(lr.rd.z));
    lr.sMax = __builtin_fmax(__builtin_fmax(fabs(lr.s.x), fabs(lr.s.y)), fabs(lr.s.z));
    lr.dirOk = __builtin_fmin(__builtin_fmin(fabs(lr.d.x), fabs(lr.d.y)), fabs(lr.d.z)) >= 1e-6;
}

// BBox::testIntersect, certified (dev_boxcert.hpp): decided from the ray's parameter interval against the box; the reference's
// arithmetic (box_test) runs only for the lanes the interval cannot decide.  `st` is the interval, for the walk below a root.
FD bool box_test_cert(const Box6& b, const LocalRay& lr, const CertRay& cr, TState& st)
{
    st = tstate_box(b.lox, b.loy, b.loz, b.hix, b.hiy, b.hiz, lr.s.x, lr.s.y, lr.s.z, lr.rd.x, lr.rd.y, lr.rd.z);
    bool res, unc;
    cert_decide(st, cr, res, unc);
    if (unc) res = box_test(b, lr.s, lr.d, lr.rd);
    return res;
}

FD Box6 kd_box(const FRAY_RO DKdBox* n)
{
    Box6 b;
    b.lox = n->lo[0]; b.loy = n->lo[1]; b.loz = n->lo[2];
    b.hix = n->hi[0]; b.hiy = n->hi[1]; b.hiz = n->hi[2];
    return b;
}

// The walk's stack of pending far children (node index; bit 31 = the node is a leaf).  The reference's recursion returns to a
// parent to try its second child (mesh.cpp:386-391); both children's box tests are pure functions of the ray, so they are evaluated
// together when the parent is entered and a far child that passed waits here until the near subtree is done without an accepted
// hit.  The 16 most recent entries of a lane live in LDS (stride 256 words: a lane always hits bank lane % 32, no conflicts),
// older ones in the lane's scratch memory: a tree is at most FRAY_KD_MAX_DEPTH levels deep (frayhip_scene_create checks; the
// reference's builder stops at 65, constants.h:39), so at most that many children are ever pending.
struct KdStack {
    int sp, lo;                                      // entries [lo, sp) are in LDS, [0, lo) in scratch
    int spill[FRAY_KD_MAX_DEPTH - FRAY_KD_LDS_STACK];
};
FD int* kd_stack_slot(int i)
{
    __shared__ int slots[FRAY_KD_LDS_STACK][256];
    return &slots[i & (FRAY_KD_LDS_STACK - 1)][threadIdx.x];
}
FD void kd_push(KdStack& k, int e)
{
    if (k.sp - k.lo == FRAY_KD_LDS_STACK) {          // the oldest LDS entry makes room
        k.spill[k.lo] = *kd_stack_slot(k.lo);
        k.lo++;
    }
    *kd_stack_slot(k.sp) = e;
    k.sp++;
}
FD int kd_pop(KdStack& k)                            // caller checked sp > 0
{
    k.sp--;
    if (k.sp < k.lo) { k.lo = k.sp; return k.spill[k.sp]; }
    return *kd_stack_slot(k.sp);
}

// Mesh::intersect (mesh.cpp:144-165).  On true: gamma / tri / l2 / l3 describe info.
template <int ST>
FD bool mesh_intersect(const FRAY_RO DMesh& M, LocalRay& lr, double& gamma, int& tri, double& l2, double& l3, Cnt& c)
{
    prepare_ray<ST>(lr);
    const V3 s = lr.s, d = lr.d, rd = lr.rd;
    const CertRay cr = cert_ray(lr.rmax, lr.sMax, lr.dirOk, M.boxMax);
    Box6 box;
    box.lox = M.bmin[0]; box.loy = M.bmin[1]; box.loz = M.bmin[2];
    box.hix = M.bmax[0]; box.hiy = M.bmax[1]; box.hiz = M.bmax[2];
    TState st;                                       // the ray's interval against the current node's box
    const bool rootHit = box_test_cert(box, lr, cr, st);
    STAMP(2);
    if (!rootHit) return false;
    gamma = 1e99;
    const int culling = M.culling;
    if (!M.hasKd) {
        bool found = false;
        const int n = M.nTris;
        for (int i = 0; i < n; i++)
            if (tri_test<ST>(M.tris + i, culling, s, d, gamma, l2, l3, c)) { found = true; tri = i; }
        STAMP(3);
        return found;
    }
    // ---- KD walk (Mesh::intersectKD, mesh.cpp:357-394) ----
    // while-while: an inner loop whose every iteration handles ONE inner node for every lane still walking (both child tests, then
    // down into the near child, or the far one, or on to the most recent pending far child), until the lane stands in a leaf or has
    // nothing pending; then the leaf's triangles; then back to walking.  Lanes of a wave therefore run child tests together and
    // triangle tests together.  Per lane the outcomes of the box tests, the order of inner nodes, leaves and triangles, and the first
    // accepted leaf are the reference's recursion, step for step.
    const FRAY_RO DKd* kd = M.kd;
    const FRAY_RO DKdBox* kdb = M.kdBox;
    KdStack stk;
    stk.sp = 0; stk.lo = 0;
    int P = 0;                                       // the inner node being entered
    int leaf = -1;                                   // >= 0: the lane stands in this leaf
    bool alive = true;
    for (;;) {
        while (alive && leaf < 0) {
            bump<ST>(c.kdInner);
            const double split = kd[P].split;
            const int child0 = kd[P].child0, meta = kd[P].meta;
            const int axis = meta & 3;
            // the split plane's parameter (the reference's own expression for the plane, bbox.h:97) and the children's intervals:
            // one child takes the plane as its far plane, the other as its near plane
            const double diff = split - comp(s, axis);
            const double ra = comp(rd, axis);
            const double ts = diff * ra;
            const int nearCh = diff > 0 ? 0 : 1;                    // ray.start[axis] < splitPos, mesh.cpp:381: visited first
            const bool nearTakesFar = (nearCh == 0) == (ra > 0);    // the first child has the split plane as its FAR plane
            const TState sa = tstate_child(st, ts, true), sb = tstate_child(st, ts, false);
            bool yA, uA, yB, uB;
            cert_decide(sa, cr, yA, uA);
            cert_decide(sb, cr, yB, uB);
            bool hitNear = nearTakesFar ? yA : yB, hitFar = nearTakesFar ? yB : yA;
            const bool uncNear = nearTakesFar ? uA : uB, uncFar = nearTakesFar ? uB : uA;
            if (uncNear || uncFar) {                                // within margins of an edge: BBox::split + testIntersect as the reference computes them
                const Box6 pb = kd_box(kdb + P);
                if (uncNear) {
                    Box6 cb = pb;
                    if (nearCh == 0) box_set_hi(cb, axis, split); else box_set_lo(cb, axis, split);
                    hitNear = box_test(cb, s, d, rd);
                }
                if (uncFar) {
                    Box6 cb = pb;
                    if (nearCh == 1) box_set_hi(cb, axis, split); else box_set_lo(cb, axis, split);
                    hitFar = box_test(cb, s, d, rd);
                }
            }
            const int farNode = child0 + 1 - nearCh, farLeaf = (meta >> (3 - nearCh)) & 1;
            int next = -1, nextLeaf = 0;
            if (hitNear) {
                if (hitFar) kd_push(stk, farNode | (farLeaf << 31));
                next = child0 + nearCh; nextLeaf = (meta >> (2 + nearCh)) & 1;
                st = nearTakesFar ? sa : sb;
            } else if (hitFar) {
                next = farNode; nextLeaf = farLeaf;
                st = nearTakesFar ? sb : sa;
            }
            STAMP(4);
            if (next < 0) {                                         // neither child: on to the most recent pending far child
                if (stk.sp == 0) { alive = false; break; }
                const int e = kd_pop(stk);
                next = e & 0x7fffffff; nextLeaf = (unsigned)e >> 31;
                if (!nextLeaf) {
                    const Box6 b = kd_box(kdb + next);
                    st = tstate_box(b.lox, b.loy, b.loz, b.hix, b.hiy, b.hiz, s.x, s.y, s.z, rd.x, rd.y, rd.z);
                }
                STAMP(9);
            }
            if (nextLeaf) leaf = next; else P = next;
        }
        if (!alive) return false;
        // ---- leaf: test every triangle, accept iff found && inside(leaf box, ip)
        {
#ifdef FRAY_LEAFSTAT
            {   // diagnostic build: how coherent are the lanes of a wave when they reach their leaves?  [0] leaf phases of a wave, [1] phases in which every
                // active lane stands in the SAME leaf, [2] active lanes summed, [3] distinct leaves summed (counted by peeling off the first lane's leaf)
                const unsigned long long act = __ballot(true);
                unsigned long long rest = act;
                int distinct = 0;
                while (rest) {
                    const int l0 = __builtin_amdgcn_readlane(leaf, (int)__builtin_ctzll(rest));
                    rest &= ~__ballot(leaf == l0);
                    distinct++;
                }
                if ((threadIdx.x & 63) == (unsigned)__builtin_ctzll(act)) {
                    atomicAdd(&g_leafStat[0], 1ull); if (distinct == 1) atomicAdd(&g_leafStat[1], 1ull);
                    atomicAdd(&g_leafStat[2], (unsigned long long)__popcll(act)); atomicAdd(&g_leafStat[3], (unsigned long long)distinct);
                }
            }
#endif
            const int beg = kd[leaf].triBegin, cnt = kd[leaf].triCount;
            bool found = false;
            const FRAY_RO DTri* lt = M.ltris + beg;
            const FRAY_RO DTri32* lf = M.ltris32 + beg;
            // the ray as the certified triangle filter reads it (dev_tricert.hpp): FP32, relative to the mesh's reference point.  Made again in
            // every leaf (nine instructions) rather than kept in seven registers through the walk
            // (the empty asm keeps the optimiser from hoisting the conversions out of the walk's loop, where they would hold those registers)
            double hx = s.x, hy = s.y, hz = s.z, hdx = d.x, hdy = d.y, hdz = d.z;
            asm volatile("" : "+v"(hx), "+v"(hy), "+v"(hz), "+v"(hdx), "+v"(hdy), "+v"(hdz));
            const float s32x = (float)(hx - M.ref[0]), s32y = (float)(hy - M.ref[1]), s32z = (float)(hz - M.ref[2]);
            const float d32x = (float)hdx, d32y = (float)hdy, d32z = (float)hdz;
            const bool rayOk32 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(s32x), __builtin_fabsf(s32y)), __builtin_fabsf(s32z)) <= 1e9f;
            // two passes over (at most 32 at a time of) the leaf's triangles: the certified FP32 filter marks the ones the reference's test may
            // accept (dev_tricert.hpp: the others it surely rejects, and a rejected triangle leaves no trace), then the reference's arithmetic
            // runs on the marked ones in their order.  A wave spends its FP64 tests on max-over-lanes CANDIDATES instead of triangles.
            for (int base = 0; base < cnt; base += 32) {
                const int m = cnt - base < 32 ? cnt - base : 32;
                unsigned cand = 0;
                for (int t = 0; t < m; t++) {
                    bump<ST>(c.leafRefs);
                    const FRAY_RO DTri32* r = lf + base + t;
                    // Two forms of the same test.  As straight-line code (`sure_miss(...) & rayOk32`, the mask updated by a select) the compiler unrolls the loop by
                    // two and packs the pair's FP32 operations: 3-7 % off the KD workloads.  In round 3 the Cube / CSG kernel variants had to keep the short circuit
                    // and the branch: with the straight-line form k_whitted<2> and k_pt_shadow<2> rendered wrong pictures on fuzz scenes while their instrumented
                    // twins were right (profiles/r03_experiments/README.md E).  That was the compiler's greedy allocation of the whole-wave VGPRs holding spilled
                    // SGPRs (the library is built with -wwm-regalloc=basic now): with it every variant takes the straight-line form and
                    // passes the fuzz tests and 80 more Cube / CSG seeds (bokeh.fray 16.9 -> 16.0 ms).
                    if constexpr (!FRAY_FILTER_STRAIGHT) {
                        const bool miss = rayOk32 && tri_sure_miss(r->A[0], r->A[1], r->A[2], r->AB[0], r->AB[1], r->AB[2], r->AC[0], r->AC[1], r->AC[2], r->Lq, r->Cq,
                                                                   s32x, s32y, s32z, d32x, d32y, d32z);
                        if (miss) bump<ST>(c.tri);           // the reference ran (and failed) its test on this one too
                        else cand |= 1u << t;
                    } else {
                        const bool miss = tri_sure_miss(r->A[0], r->A[1], r->A[2], r->AB[0], r->AB[1], r->AB[2], r->AC[0], r->AC[1], r->AC[2], r->Lq, r->Cq,
                                                        s32x, s32y, s32z, d32x, d32y, d32z) & rayOk32;
                        bump<ST>(c.tri, miss ? 1ull : 0ull);
                        cand |= (miss ? 0u : 1u) << t;
                    }
                }
                STAMP(15);
                while (cand) {
                    const int t = base + __builtin_ctz(cand);
                    cand &= cand - 1;
                    if (tri_test<ST>(lt + t, culling, s, d, gamma, l2, l3, c)) { found = true; tri = lt[t].index; }
                }
            }
            STAMP(11);
            if (found && box_inside(kd_box(kdb + leaf), s + d * gamma)) return true;
        }
        // the leaf is done: on to the most recent pending far child (a pending leaf waits for the next round of triangle tests)
        if (stk.sp == 0) return false;
        {
            const int e = kd_pop(stk);
            const int next = e & 0x7fffffff;
            if ((unsigned)e >> 31) leaf = next;
            else {
                leaf = -1;
                P = next;
                const Box6 b = kd_box(kdb + next);
                st = tstate_box(b.lox, b.loy, b.loz, b.hix, b.hiy, b.hiz, s.x, s.y, s.z, rd.x, rd.y, rd.z);
            }
            STAMP(9);
        }
    }
}

// ---- Cube and CSG (geometry.cpp:85-194) -- only in the <ST & 2> kernel variants ----------------------
// One intersection as <Geometry>::intersect reports it, reduced to what later stages rebuild the
// full IntersectionInfo from: dist (the geometry's own info.dist), the local hit point, and a code
// (cube side / triangle index) with the triangle's barycentrics.
struct GHit { double dist; V3 ip; int code; double l2, l3; int leafKind, leafIndex; };   // leaf: the non-CSG geometry that produced it

FD bool cube_intersect(const FRAY_RO DCube& Cb, V3 s, V3 d, GHit& h)   // Cube::intersect, geometry.cpp:85-137
{
    const V3 O = ld3(Cb.O);
    const double hs = Cb.halfSide;
    double best = 1e99;
    for (int side = 0; side < 6; side++) {
        const int ax = side >> 1;
        const double st = comp(s, ax), dr = comp(d, ax);
        const double target = (side & 1) ? comp(O, ax) + hs : comp(O, ax) - hs;
        if (fabs(dr) < 1e-9) continue;
        double mult = fray_div(target - st, dr);
        if (mult < 0) continue;
        V3 ip = s + d * mult;
        if (ip.x < O.x - hs - 1e-6 || ip.x > O.x + hs + 1e-6) continue;
        if (ip.y < O.y - hs - 1e-6 || ip.y > O.y + hs + 1e-6) continue;
        if (ip.z < O.z - hs - 1e-6 || ip.z > O.z + hs + 1e-6) continue;
        double dist = length(s - ip);
        if (dist < best) { best = dist; h.dist = dist; h.ip = ip; h.code = side; }
    }
    return best < 1e99;
}

// <Geometry>::intersect of a non-CSG geometry, by kind.
template <int ST>
FD bool prim_intersect(const DScene& S, int kind, int index, V3 s, V3 d, V3 rd, GHit& h, Cnt& c)
{
    h.code = -1; h.l2 = 0; h.l3 = 0;
    if (kind == 0) {
        bump<ST>(c.prim);
        const FRAY_RO DPlane& P = S.planes[index];
        if (s.y > P.height && d.y >= 0) return false;
        if (s.y < P.height && d.y <= 0) return false;
        double scaling = fray_div(fabs(s.y - P.height), fabs(d.y));
        V3 ip = s + d * scaling;
        if (fabs(ip.x) > P.limit) return false;
        if (fabs(ip.z) > P.limit) return false;
        h.ip = ip;
        h.dist = length(s - ip);
        return true;
    }
    if (kind == 1) {
        bump<ST>(c.prim);
        const FRAY_RO DSphere& Sp = S.spheres[index];
        V3 H = s - ld3(Sp.O);
        double B = 2 * dot(d, H);
        double C = lengthSqr(H) - Sp.R * Sp.R;
        double Disc = B * B - 4 * 1 * C;
        if (Disc < 0) return false;
        double sq = fray_sqrt(Disc);
        double p1 = (-B + sq) / (2 * 1.0), p2 = (-B - sq) / (2 * 1.0);
        double smaller = p2 < p1 ? p2 : p1, larger = p1 < p2 ? p2 : p1;
        if (larger < 0) return false;
        double dd = (smaller >= 0) ? smaller : larger;
        h.ip = s + d * dd;
        h.dist = length(s - h.ip);
        return true;
    }
    if (kind == 2) {
        bump<ST>(c.prim);
        return cube_intersect(S.cubes[index], s, d, h);
    }
    // mesh
    LocalRay lr;
    lr.s = s; lr.d = d; lr.cls = 0; lr.haveRd = false;
    double gamma;
    int tri = -1;
    double l2 = 0, l3 = 0;
    if (!mesh_intersect<ST>(S.meshes[index], lr, gamma, tri, l2, l3, c)) return false;
    h.dist = gamma; h.ip = s + d * gamma; h.code = tri; h.l2 = l2; h.l3 = l3;
    return true;
}


// CsgOp::intersect (geometry.cpp:139-194), nested CsgOps included, as ONE loop over an explicit stack of activations -- the reference
// recurses through the Geometry virtual (geometry.cpp:146) without bound; here the nesting bound is the depth of the stack
// (FRAY_CSG_DEPTH; frayhip_scene_create rejects deeper trees), and there is one copy of the geometry code (prim_intersect) and no call.
//
// One activation = one CsgOp asked about one ray start.  findAllIntersections keeps up to 30 intersections per operand
// (geometry.cpp:144-152); per activation only their distances are kept, ordered exactly as libstdc++'s std::sort orders them
// (dev_sort.hpp: the sort is not stable, and coincident faces of two operands are equally distant).  The winning intersection's full
// record is then derived again by a second pass over that operand's chain up to it -- the same calls on the same rays, so the same
// bits -- which keeps an activation at ~600 bytes of scratch.  The machine:
//   ASK      the current activation wants the next intersection of its current operand from `start`: a plain geometry is intersected
//            here (all lanes of the wave that stand at ASK do it together), a CsgOp operand pushes a new activation;
//   DELIVER  an answer (found / not found) goes to the activation on top: pass 0 records the distance and asks again, or moves on to
//            the right operand, or -- both chains done -- sorts, walks the in / out states and starts pass 1 (or answers "no hit");
//            pass 1 counts up to the winner and answers with it.  An answer pops the activation and is delivered to the one below.
// Work counters: pass 1 repeats calls pass 0 counted, so everything below an activation in pass 1 is not counted (`quietAt`).
struct CsgFrame {
    double s[3];          // the ray start this CsgOp was asked about
    double winDist;
    double last;          // pass 0: the distance recorded last (is the current chain in non-decreasing order?)
    int32_t csg;          // index into DScene::csgs
    unsigned char n, k, cnt0, cnt1, winOp, winK, pass, op;
    uint32_t unsorted;    // pass 0: some chain came out of order (NaNs included): the walk needs std::sort's own order
    int32_t leftIndex, rightIndex, kinds;     // the CsgOp itself (DCsg), read once when the activation starts: kinds = leftKind | rightKind << 8 | op << 16
};
FD void csg_frame_op(CsgFrame& F, const FRAY_RO DCsg& G)
{
    F.leftIndex = G.leftIndex; F.rightIndex = G.rightIndex;
    F.kinds = G.leftKind | (G.rightKind << 8) | (G.op << 16);
}
// Round 5: what pass 0 computed is KEPT for the first FRAY_CSG_MEMO intersections of the FRAY_CSG_MEMO_LEVELS outermost activations, so that the winner's
// record is looked up instead of derived again (the reference computes every intersection once and keeps all thirty records per operand, geometry.cpp:144-152;
// deriving the winner again was this port's way of not holding them).  In a nested tree a pass 1 re-asks whole sub-trees, each with its own two passes: for
// csg_nested.fray's three-level objects a ray that hits asked 218 plain geometries, now 66.  Longer chains and deeper activations still take pass 1.
#ifndef FRAY_CSG_MEMO
#define FRAY_CSG_MEMO 8
#endif
#ifndef FRAY_CSG_MEMO_LEVELS
#define FRAY_CSG_MEMO_LEVELS 4
#endif
struct CsgMemo { double ip[3]; double l2, l3; int32_t code, leafKind, leafIndex, pad; };
template <int ST>
FD bool csg_intersect(const DScene& S, int rootCsg, V3 s, V3 d, V3 rd, GHit& win, bool& envelope, Cnt& c)
{
    // The activation on top of the stack lives in registers (F); fr[] holds the ones below it and is touched only when a CsgOp operand is entered or
    // left.  (Round 4 indexed fr[level] for every field: a scratch access each, on the path every ray of a scene with a CSG floor takes.)
    CsgFrame fr[FRAY_CSG_DEPTH];
    double dist[FRAY_CSG_DEPTH][2 * FRAY_CSG_MAX];
    unsigned char order[FRAY_CSG_DEPTH][2 * FRAY_CSG_MAX];
#if FRAY_CSG_MEMO
    CsgMemo memo[FRAY_CSG_MEMO_LEVELS][FRAY_CSG_MEMO];
#endif
    int level = 0, quietAt = -1;
    V3 start = s;
    CsgFrame F;
    F.s[0] = s.x; F.s[1] = s.y; F.s[2] = s.z;
    F.winDist = 0; F.last = 0; F.unsorted = 0;
    F.csg = rootCsg; F.n = F.k = F.cnt0 = F.cnt1 = F.winOp = F.winK = F.pass = F.op = 0;
    csg_frame_op(F, S.csgs[rootCsg]);
    for (;;) {
        // ---- ASK
        bool ok;
        GHit h;
        {
            // (the operator's record travels in the activation: a load per question -- its address depends on the lane's activation -- stood in front of every
            // operand's own load)
            const int op = F.op;
            const int kind = op == 0 ? (F.kinds & 255) : ((F.kinds >> 8) & 255), index = op == 0 ? F.leftIndex : F.rightIndex;
            if (kind == 4) {
                if (level + 1 < FRAY_CSG_DEPTH) {
                    fr[level] = F;
                    level++;
                    F.s[0] = start.x; F.s[1] = start.y; F.s[2] = start.z;
                    F.winDist = 0; F.last = 0; F.unsorted = 0;
                    F.csg = index; F.n = F.k = F.cnt0 = F.cnt1 = F.winOp = F.winK = F.pass = F.op = 0;
                    csg_frame_op(F, S.csgs[index]);
                    continue;
                }
                envelope = true;          // unreachable: frayhip_scene_create rejects deeper trees
                ok = false;
            } else {
                if constexpr ((ST & 1) != 0) {
                    const Cnt keep = c;
                    ok = prim_intersect<ST>(S, kind, index, start, d, rd, h, c);
                    if (quietAt >= 0) { const unsigned env = c.envelope; c = keep; c.envelope = env; }
                } else {
                    STAMP(20);
                    ok = prim_intersect<ST>(S, kind, index, start, d, rd, h, c);
                    if (kind != 3) STAMP(18);
                }
                h.leafKind = kind; h.leafIndex = index;
            }
        }
        // ---- DELIVER, until an activation asks again
        for (;;) {
            const V3 fs = v3(F.s[0], F.s[1], F.s[2]);
            bool answer = false;                                  // this activation is done: (ok, h) is ITS answer
            if (ok && F.pass == 0 && F.k == FRAY_CSG_MAX) ok = false;          // `counter-- > 0`: the 31st intersection is found and dropped
            if (ok) {
                if (F.pass == 0) {
                    const double dv = F.k > 0 ? length(h.ip - fs) : h.dist;    // geometry.cpp:155-156
                    if (F.k > 0 && !(dv >= F.last)) F.unsorted = 1;
                    if (!(dv == dv)) F.unsorted = 1;
                    F.last = dv;
                    dist[level][F.n] = dv;
#if FRAY_CSG_MEMO
                    if (level < FRAY_CSG_MEMO_LEVELS && F.n < FRAY_CSG_MEMO) {
                        CsgMemo& M = memo[level][F.n];
                        M.ip[0] = h.ip.x; M.ip[1] = h.ip.y; M.ip[2] = h.ip.z;
                        M.l2 = h.l2; M.l3 = h.l3;
                        M.code = h.code; M.leafKind = h.leafKind; M.leafIndex = h.leafIndex;
                    }
#endif
                    F.n++; F.k++;
                    start = h.ip + d * 1e-6;
                } else if (F.k == F.winK) {
                    h.dist = F.winDist;
                    answer = true;
                } else {
                    F.k++;
                    start = h.ip + d * 1e-6;
                }
            } else if (F.pass == 0) {
                if (F.op == 0) { F.cnt0 = F.k; F.op = 1; F.k = 0; start = fs; }
                else {
                    F.cnt1 = F.k;
                    const int n = F.n, c0 = F.cnt0, gop = F.kinds >> 16;
                    bool inL = (c0 & 1) == 1, inR = (F.cnt1 & 1) == 1;
                    auto bop = [&](bool l, bool r) { return gop == 0 ? (l || r) : (gop == 1 ? (l && r) : (l && !r)); };
                    const bool cur = bop(inL, inR);
                    int winE = -1;                                // the winner: entry e of dist[level] (e < c0: the left operand's e-th, else the right one's)
                    if (!F.unsorted && n <= 16) {
                        // Both chains in non-decreasing order and at most sixteen entries: libstdc++'s std::sort is then its final insertion sort alone (strict
                        // comparisons: equal elements keep their order), i.e. the stable merge of the two chains with the left operand's entry first among
                        // equals -- walked here without sorting anything.
                        int i = 0, j = c0;
                        double dl = i < c0 ? dist[level][i] : 0.0, dr = j < n ? dist[level][j] : 0.0;
                        while (winE < 0 && (i < c0 || j < n)) {
                            const bool takeL = j >= n || (i < c0 && !(dr < dl));
                            if (takeL) inL = !inL; else inR = !inR;
                            if (bop(inL, inR) != cur) winE = takeL ? i : j;
                            else if (takeL) { i++; if (i < c0) dl = dist[level][i]; }
                            else { j++; if (j < n) dr = dist[level][j]; }
                        }
                    } else {
                        for (int i = 0; i < n; i++) order[level][i] = (unsigned char)i;
                        StdSort sorter{dist[level], order[level]};
                        sorter.sort(n);
                        for (int i = 0; i < n && winE < 0; i++) {
                            const int e = order[level][i];
                            if (e < c0) inL = !inL; else inR = !inR;
                            if (bop(inL, inR) != cur) winE = e;
                        }
                    }
                    if (winE < 0) answer = true;                  // (false, -)
                    else {
                        const int o = winE < c0 ? 0 : 1;
                        F.winOp = (unsigned char)o; F.winK = (unsigned char)(o == 0 ? winE : winE - c0); F.winDist = dist[level][winE];
#if FRAY_CSG_MEMO
                        if (level < FRAY_CSG_MEMO_LEVELS && winE < FRAY_CSG_MEMO) {
                            const CsgMemo& M = memo[level][winE];
                            h.ip = v3(M.ip[0], M.ip[1], M.ip[2]);
                            h.l2 = M.l2; h.l3 = M.l3;
                            h.code = M.code; h.leafKind = M.leafKind; h.leafIndex = M.leafIndex;
                            h.dist = F.winDist;
                            ok = true;
                            answer = true;
                        } else
#endif
                        {
                            F.pass = 1; F.op = F.winOp; F.k = 0; start = fs;
                            if (quietAt < 0) quietAt = level;
                        }
                    }
                }
            } else answer = true;                                 // unreachable: pass 0 found this intersection
            if (!answer) break;
            if (quietAt == level) quietAt = -1;
            if (level == 0) { win = h; STAMP(19); return ok; }
            level--;
            F = fr[level];
        }
        STAMP(19);
    }
}

// ---- CsgOp over two PLAIN operands (plane / sphere / cube: DCsg::flat) without the machine ---------------------------
// The same calls in the same order on the same rays as csg_intersect makes for such a node -- findAllIntersections on the left operand, on the right
// operand, the sorted walk, the winner's chain again (geometry.cpp:139-194) -- with everything in registers: a convex operand is crossed twice, so four
// distances per operand are kept; a chain that finds a fifth intersection, or whose distances do not come out in non-decreasing order, sets `fallback` and
// the node goes through the general machine, which computes the same answer.  The order of the (at most eight) distances: libstdc++'s std::sort of up to
// sixteen elements is its final insertion sort alone (bits/stl_algo.h __final_insertion_sort: strict comparisons, equal elements keep their order), i.e. a
// stable merge of the two chains in which the left operand's entries come first among equals -- what the loop below walks.
// (csg_nested.fray, bokeh.fray: the floor every ray of the scene asks is Cube - Cube.)
FD double sel4(double a0, double a1, double a2, double a3, int i) { return i == 0 ? a0 : (i == 1 ? a1 : (i == 2 ? a2 : a3)); }
template <int ST>
FD bool prim_intersect_plain(const DScene& S, int kind, int index, V3 s, V3 d, GHit& h, Cnt& c)
{
    h.code = -1; h.l2 = 0; h.l3 = 0;
    bump<ST>(c.prim);
    if (kind == 0) {
        const FRAY_RO DPlane& P = S.planes[index];
        if (s.y > P.height && d.y >= 0) return false;
        if (s.y < P.height && d.y <= 0) return false;
        double scaling = fray_div(fabs(s.y - P.height), fabs(d.y));
        V3 ip = s + d * scaling;
        if (fabs(ip.x) > P.limit) return false;
        if (fabs(ip.z) > P.limit) return false;
        h.ip = ip;
        h.dist = length(s - ip);
        return true;
    }
    if (kind == 1) {
        const FRAY_RO DSphere& Sp = S.spheres[index];
        V3 H = s - ld3(Sp.O);
        double B = 2 * dot(d, H);
        double C = lengthSqr(H) - Sp.R * Sp.R;
        double Disc = B * B - 4 * 1 * C;
        if (Disc < 0) return false;
        double sq = fray_sqrt(Disc);
        double p1 = (-B + sq) / (2 * 1.0), p2 = (-B - sq) / (2 * 1.0);
        double smaller = p2 < p1 ? p2 : p1, larger = p1 < p2 ? p2 : p1;
        if (larger < 0) return false;
        double dd = (smaller >= 0) ? smaller : larger;
        h.ip = s + d * dd;
        h.dist = length(s - h.ip);
        return true;
    }
    return cube_intersect(S.cubes[index], s, d, h);
}
template <int ST>
FD bool csg_flat_intersect(const DScene& S, const FRAY_RO DCsg& G, V3 s, V3 d, GHit& win, bool& fallback, Cnt& c)
{
    fallback = false;
    double l0 = 0, l1 = 0, l2 = 0, l3 = 0, r0 = 0, r1 = 0, r2 = 0, r3 = 0;
    int nl = 0, nr = 0;
    // pass 0: findAllIntersections (geometry.cpp:144-158) on the left, then on the right operand
    for (int op = 0; op < 2; op++) {
        const int kind = op == 0 ? G.leftKind : G.rightKind, index = op == 0 ? G.leftIndex : G.rightIndex;
        V3 start = s;
        int k = 0;
        for (;;) {
            GHit h;
            if (!prim_intersect_plain<ST>(S, kind, index, start, d, h, c)) break;
            if (k == 4) { fallback = true; break; }
            const double dist = k > 0 ? length(h.ip - s) : h.dist;           // geometry.cpp:155-156
            if (op == 0) { l0 = k == 0 ? dist : l0; l1 = k == 1 ? dist : l1; l2 = k == 2 ? dist : l2; l3 = k == 3 ? dist : l3; }
            else { r0 = k == 0 ? dist : r0; r1 = k == 1 ? dist : r1; r2 = k == 2 ? dist : r2; r3 = k == 3 ? dist : r3; }
            k++;
            start = h.ip + d * 1e-6;
        }
        if (op == 0) nl = k; else nr = k;
    }
    // a chain out of order (never seen; NaNs included) leaves the insertion sort's result to the general machine
    const bool sortedL = (nl < 2 || l0 <= l1) && (nl < 3 || l1 <= l2) && (nl < 4 || l2 <= l3);
    const bool sortedR = (nr < 2 || r0 <= r1) && (nr < 3 || r1 <= r2) && (nr < 4 || r2 <= r3);
    if (!(sortedL && sortedR)) fallback = true;
    if (fallback) return false;
    // the walk over the sorted intersections (geometry.cpp:160-185)
    bool inL = (nl & 1) == 1, inR = (nr & 1) == 1;
    auto bop = [&](bool l, bool r) { return G.op == 0 ? (l || r) : (G.op == 1 ? (l && r) : (l && !r)); };
    const bool cur = bop(inL, inR);
    int i = 0, j = 0, winOp = -1, winK = 0;
    double winDist = 0;
    while (winOp < 0 && (i < nl || j < nr)) {
        const double dl = sel4(l0, l1, l2, l3, i), dr = sel4(r0, r1, r2, r3, j);
        const bool takeL = j >= nr || (i < nl && !(dr < dl));       // equal distances: the left operand's entry stands first (stable)
        if (takeL) inL = !inL; else inR = !inR;
        if (bop(inL, inR) != cur) { winOp = takeL ? 0 : 1; winK = takeL ? i : j; winDist = takeL ? dl : dr; }
        if (takeL) i++; else j++;
    }
    if (winOp < 0) return false;
    // pass 1: the winner's record, by walking its operand's chain again (the same calls on the same rays; not counted twice)
    const int kind = winOp == 0 ? G.leftKind : G.rightKind, index = winOp == 0 ? G.leftIndex : G.rightIndex;
    const Cnt keep = c;
    V3 start = s;
    GHit h;
    bool ok = true;
    for (int k = 0; k <= winK && ok; k++) {
        ok = prim_intersect_plain<ST>(S, kind, index, start, d, h, c);
        start = h.ip + d * 1e-6;
    }
    if (ST & 1) { const unsigned env = c.envelope; c = keep; c.envelope = env; }
    if (!ok) { fallback = true; return false; }          // unreachable: pass 0 found this intersection
    h.dist = winDist;
    h.leafKind = kind; h.leafIndex = index;
    win = h;
    return true;
}

// Geometry part of Node::intersect for node N on the local ray; on a hit returns the local
// intersection point and fills t / tri / l2 / l3.
struct LeafOut { int kind, index; };     // Cube / CSG nodes only: the plain geometry at the bottom of the tree that produced the hit
template <int ST>
FD bool geom_intersect(const DScene& S, const FRAY_RO DNode& N, int nodeIndex, LocalRay& lr, V3& ipl, double& t, int& tri, double& l2, double& l3, LeafOut* lo, Cnt& c)
{
    const V3 ls = lr.s, ld = lr.d;
    if (N.geomKind == 0) {   // Plane::intersect, geometry.cpp:30-50
        bump<ST>(c.prim);
        const FRAY_RO DPlane& P = S.planes[N.geomIndex];
        if (ls.y > P.height && ld.y >= 0) return false;
        if (ls.y < P.height && ld.y <= 0) return false;
        double travelByY = fabs(ls.y - P.height);
        double unitTravel = fabs(ld.y);
        double scaling = fray_div(travelByY, unitTravel);
        V3 ip = ls + ld * scaling;
        if (fabs(ip.x) > P.limit) return false;
        if (fabs(ip.z) > P.limit) return false;
        ipl = ip;
        t = scaling;
        return true;
    }
    if (N.geomKind == 1) {   // Sphere::intersect, geometry.cpp:52-83
        bump<ST>(c.prim);
        const FRAY_RO DSphere& Sp = S.spheres[N.geomIndex];
        V3 H = ls - ld3(Sp.O);
        double A = 1;
        double B = 2 * dot(ld, H);
        double C = lengthSqr(H) - Sp.R * Sp.R;
        double Disc = B * B - 4 * A * C;
        if (Disc < 0) return false;
        double sqrtDisc = fray_sqrt(Disc);
        double p1 = (-B + sqrtDisc) / (2 * A);
        double p2 = (-B - sqrtDisc) / (2 * A);
        double smaller = p2 < p1 ? p2 : p1;   // std::min(p1, p2)
        double larger = p1 < p2 ? p2 : p1;    // std::max(p1, p2)
        if (larger < 0) return false;
        double dd = (smaller >= 0) ? smaller : larger;
        ipl = ls + ld * dd;
        t = dd;
        return true;
    }
    if ((ST & 2) && N.geomKind == 2) {   // Cube
        bump<ST>(c.prim);
        GHit h;
        if (!cube_intersect(S.cubes[N.geomIndex], ls, ld, h)) return false;
        ipl = h.ip;
        tri = h.code;
        t = h.dist;
        lo->kind = 2; lo->index = N.geomIndex;
        return true;
    }
    if ((ST & 2) && N.geomKind == 4) {   // CSG: the winner is re-derived in finalize_hit
        if constexpr (!(ST & 1)) {
            // The reference's CsgOp has no bounding volume: every ray of the scene runs findAllIntersections on both operand trees.  A ray that
            // CERTIFIABLY passes the tree's bounding box by (dev_misscert.hpp: then no operand reports an intersection, as the reference computes
            // them) gets the reference's answer -- none -- without the machine.  (The counting variants run it: their counters are the reference's calls.)
            const FRAY_RO DNodeX& X = S.nodesX[nodeIndex];
            const bool passesBy = X.csgBox && ray_surely_misses_box(X.cc[0], X.cc[1], X.cc[2], X.ch[0], X.ch[1], X.ch[2], X.cM, ls.x, ls.y, ls.z, ld.x, ld.y, ld.z);
            STAMP(16);
            if (passesBy) return false;
        }
        GHit h;
        bool env = false, viaMachine = true, found = false;
        const FRAY_RO DCsg& G0 = S.csgs[N.geomIndex];
        if (G0.flat) {
            const Cnt before = c;
            bool fb;
            found = csg_flat_intersect<ST>(S, G0, ls, ld, h, fb, c);
            viaMachine = fb;                                   // (a chain of more than four intersections: the general machine answers ...
            if ((ST & 1) && fb) { const unsigned e0 = c.envelope; c = before; c.envelope = e0; }       // ... and counts the same calls again)
            STAMP(17);
        }
        if (viaMachine) { found = csg_intersect<ST>(S, N.geomIndex, ls, ld, ray_rdir(ld), h, env, c); STAMP(20); }
        if (!found) return false;
        if (env) c.envelope = 1;
        ipl = h.ip;
        tri = h.code;
        t = h.dist;
        l2 = h.l2; l3 = h.l3;
        lo->kind = h.leafKind; lo->index = h.leafIndex;
        return true;
    }
    // mesh
    double gamma;
    if (N.tlTris > 0) {      // no KD-tree: Mesh::intersect's brute-force loop (mesh.cpp:146-161) on the node's own copy of the mesh header
        prepare_ray<ST>(lr);
        Box6 box;
        box.lox = N.bmin[0]; box.loy = N.bmin[1]; box.loz = N.bmin[2];
        box.hix = N.bmax[0]; box.hiy = N.bmax[1]; box.hiz = N.bmax[2];
        // Scenes without KD meshes (cornell_box: seven boxes per ray) keep the reference's arithmetic with its wave-uniform exits: there the
        // interval form measured slower (headline 105.4 -> 109.1 ms: more registers, fewer waves).  Beside KD meshes the interval decides.
        bool rootHit;
        if constexpr (kd_variant(ST)) {
            TState st;
            rootHit = box_test_cert(box, lr, cert_ray(lr.rmax, lr.sMax, lr.dirOk, N.boxMax), st);
        } else {
            const FRAY_RO DNodeX& X = S.nodesX[nodeIndex];
            Box6 be;
            be.lox = X.bminE[0]; be.loy = X.bminE[1]; be.loz = X.bminE[2];
            be.hix = X.bmaxE[0]; be.hiy = X.bmaxE[1]; be.hiz = X.bmaxE[2];
            rootHit = box_test_pre(box, be, ls, ld, lr.rd);
        }
        STAMP(2);
        if (!rootHit) return false;
        gamma = 1e99;
        bool found = false;
        const int n = N.tlTris, culling = N.tlCulling;
        const FRAY_RO DTri* tris = N.tlPtr;
        for (int i = 0; i < n; i++)
            if (tri_test<ST>(tris + i, culling, ls, ld, gamma, l2, l3, c)) { found = true; tri = i; }
        STAMP(3);
        if (!found) return false;
    } else if (!kd_variant(ST)) {
        return false;        // a mesh with a KD-tree in a scene uploaded as having none: cannot happen (frayhip_scene_create picks the variant)
    } else if (!mesh_intersect<ST>(S.meshes[N.geomIndex], lr, gamma, tri, l2, l3, c)) return false;
    ipl = ls + ld * gamma;
    t = gamma;
    return true;
}

// Node::intersect (geometry.cpp:196-208) reduced to what the closest-hit comparison needs.
template <int ST>
FD bool node_intersect(const DScene& S, int i, V3 o, V3 d, LocalRay& lr, double& dist, double& t, int& tri, double& l2, double& l3, V3* iplOut, LeafOut* lo, Cnt& c)
{
    bump<ST>(c.node);
    const FRAY_RO DNode& N = S.nodes[i];
    if (N.xfClass != lr.cls) {
        // An untransformed node (offset 0, m = invM = I, bit for bit: cornell_box's seven meshes): v * I is v.x * 1 + v.y * 0 + v.z * 0 = v.x exactly whenever
        // v.x is not a zero (then the sum's sign of zero would depend on the other components), so the two 15-operation products are skipped unless some lane
        // of the wave holds an exact zero; the normalisation is the reference's.
        // Only in the leanest variants (identity_variant: no KD walk, no texture code -- cornell_box's): beside the KD walk the extra test cost the any-hit
        // kernel a fifth of its speed (boxed Whitted 6.97 -> 8.45 ms, forest DOF 16 10.0 -> 10.6: registers in the walk's loop), and in the textured
        // variants, whose scenes (smallpt: translated planes and spheres) have no such node, its mere presence 3 % of k_whitted (3.12 -> 3.22 ms).
        bool done = false;
        if constexpr (identity_variant(ST)) {
#if FRAY_ARITH
            // (relaxed arithmetic: the directions the path tracer makes are unit vectors to an ulp: no second normalisation, no zero check)
            if (N.xfIdentity) { lr.s = o; lr.d = d; done = true; }
#else
            const bool zeros = o.x == 0 || o.y == 0 || o.z == 0 || d.x == 0 || d.y == 0 || d.z == 0;
            if (N.xfIdentity && !__any(zeros)) { lr.s = o; lr.d = normalized(d); done = true; }
#endif
        }
        if (!done) {
            lr.s = mulM(o - ld3(N.T.off), N.T.inv);
            lr.d = normalized(mulM(d, N.T.inv));
        }
        lr.cls = N.xfClass;
        lr.haveRd = false;
    }
    STAMP(1);
    V3 ipl;
    const bool hit = geom_intersect<ST>(S, N, i, lr, ipl, t, tri, l2, l3, lo, c);
    STAMP(5);             // whatever geom_intersect did not stamp itself: planes, spheres, the KD walk
    if (!hit) return false;
    if constexpr ((ST & 2) != 0) { if (iplOut) *iplOut = ipl; }
    // (an untransformed node: ipl * I + 0 is ipl up to the sign of a zero, which the squares below do not see)
    V3 ipw;
    if constexpr (identity_variant(ST)) {
#if FRAY_ARITH
        // (relaxed arithmetic: along a unit direction in world space the ray parameter IS the distance -- planes, spheres, triangles)
        if (N.xfIdentity) { dist = t; STAMP(6); return true; }
#endif
        ipw = N.xfIdentity ? ipl : mulM(ipl, N.T.m) + ld3(N.T.off);
    } else {
        ipw = mulM(ipl, N.T.m) + ld3(N.T.off);
    }
    dist = length(o - ipw);
    STAMP(6);
    return true;
}

// RectLight::intersect (lights.cpp:79-103); point lights are never hit (lights.h:68-71).
template <int ST>
FD bool light_intersect(const FRAY_RO DLight& L, V3 o, V3 d, double& dist, Cnt& c)
{
    if (L.kind == 0) return false;
    bump<ST>(c.prim);
    V3 ls = mulM(o - ld3(L.T.off), L.T.inv);
    V3 ldir = normalized(mulM(d, L.T.inv));
    if (ls.y >= 0) return false;
    if (ldir.y <= 0) return false;
    double travelByY = fabs(ls.y);
    double unitTravel = fabs(ldir.y);
    double scaling = fray_div(travelByY, unitTravel);
    V3 ip = ls + ldir * scaling;
    if (fabs(ip.x) > 0.5 || fabs(ip.z) > 0.5) return false;
    ip = mulM(ip, L.T.m) + ld3(L.T.off);
    dist = length(o - ip);
    return true;
}

// The two loops of raytrace()/pathtrace(): first node wins ties (strict <), then lights.
// gateFree: the ray's producer PROVED that it misses every gate of the scene (kernels.hpp ray_gate_class with DScene::gatesExact; dev_misscert.hpp): the
// gated nodes' geometry then reports no intersection, as the reference computes it, and is not asked (the counting variants ask: a CsgOp node's
// calls are part of their counters).
template <int ST>
FD void closest_hit(const DScene& S, V3 o, V3 d, HitT<ST>& best, Cnt& c, bool gateFree = false)
{
    bump<ST>(c.closest);
    best.node = -1;
    best.tri = -1;
    best.dist = 1e99;
    best.t = 0; best.l2 = 0; best.l3 = 0;
    const int nn = S.nNodes;
    LocalRay lr;
    lr.cls = -1;
    lr.haveRd = false;
    for (int i = 0; i < nn; i++) {
        double dist, t, l2 = 0, l3 = 0;
        int tri = -1;
        if constexpr (!(ST & 1)) { if (S.nodes[i].gated) { if (gateFree) continue; } }        // (the node's word first: wave-uniform, a scalar branch for every node that is not gated)
        if constexpr ((ST & 2) != 0) {
            // Cube / CSG variants: the winning intersection as its geometry reported it travels with the hit record (finalize_hit)
            V3 ipl;
            LeafOut lo{0, 0};
            if (node_intersect<ST>(S, i, o, d, lr, dist, t, tri, l2, l3, &ipl, &lo, c) && dist < best.dist) {
                best.node = i; best.tri = tri; best.dist = dist; best.t = t; best.l2 = l2; best.l3 = l3;
                best.ipl = ipl; best.leafKind = lo.kind; best.leafIndex = lo.index;
            }
        } else if (node_intersect<ST>(S, i, o, d, lr, dist, t, tri, l2, l3, nullptr, nullptr, c) && dist < best.dist) {
            best.node = i; best.tri = tri; best.dist = dist; best.t = t; best.l2 = l2; best.l3 = l3;
        }
    }
    if ((ST & 1) && best.node >= 0) {   // byte model: the winner's corner normals / uvs
        const FRAY_RO DNode& W = S.nodes[best.node];
        if (W.geomKind == 3 && S.meshes[W.geomIndex].smooth) c.smooth++;
    }
    STAMP(6);
    const int nl = S.nLights;
    for (int i = 0; i < nl; i++) {
        double dist;
        if (light_intersect<ST>(S.lights[i], o, d, dist, c) && dist < best.dist) {
            best.node = -2 - i;
            best.dist = dist;
        }
    }
    STAMP(7);
}

// visible(a, b), main.cpp:64-80: lights do not occlude; the first node whose (full) intersection
// lies closer than b ends the loop.
// GATES: compiled with the test for gate-free rays (the shadow kernel takes this copy only for queues whose producer certified them: for a scene without
// exact gates the per-node test -- a divergent branch on a lane flag -- cost its shadow kernel 9 %: smallpt, 1.30 -> 1.43 ms per launch).
template <int ST, bool GATES = false>
FD bool visible(const DScene& S, V3 a, V3 b, Cnt& c, bool gateFree = false)
{
    bump<ST>(c.shadow);
    V3 d = b - a;
    // main.cpp:66-70 takes distance(a, b) = length(a - b) and then normalises b - a, i.e. divides by length(b - a): the two lengths are the same
    // bits (a - b is exactly -(b - a), and squares do not see the sign), so one square root serves both
    const double maxDist = length(d);
    d = d * fray_rcp(maxDist);
    const int nn = S.nNodes;
    LocalRay lr;
    lr.cls = -1;
    lr.haveRd = false;
    for (int i = 0; i < nn; i++) {
        double dist, t, l2, l3;
        int tri;
        LeafOut lo;
        if constexpr (GATES && !(ST & 1)) { if (S.nodes[i].gated) { if (gateFree) continue; } }
        if (node_intersect<ST>(S, i, a, d, lr, dist, t, tri, l2, l3, nullptr, (ST & 2) ? &lo : nullptr, c) && dist < maxDist) return false;
    }
    return true;
}
