// HIP kernels of the renderer (gfx950, wave64).  Every kernel template is compiled per flag word ST (dev_trace.hpp: bit 0 work counters,
// bit 1 Cube / CSG, bit 2 KD-tree meshes, bit 3 textures / environment): scenes only pay for the code they can reach.
//
//   k_primary      camera ray through integer (x, y) + closest hit                                  (MODE_PRIMARY_ID)
//   k_whitted      raytrace() per pixel for scenes with recursive shaders, samples looped in order   (MODE_RENDER, gi off)
//                  -- both as persistent waves claiming 8x8 pixel tiles (claim_items)
//   k_wh_shade     Whitted as a wavefront (scenes without recursive shaders): camera ray(s), closest hit, Lambert / Phong::shade up to visible()
//   k_wh_visible   visible() for every queued light-sample segment
//   k_wh_gather    per camera sample: base + sum over lights of the visible samples' terms, the reference's FP32 order
//   k_seed         x[397] of the mt19937 seeding recurrence per (pixel, sample)
//   k_pt_init      path-tracing batch: generator cursors, pixel jitter, camera rays -> dense path queue
//   k_pt_bounce    one pathtrace() iteration for every live path: closest hit, bump, discarded spawn, next-event sample (written to the
//                  wave's segment of the shadow queue where it is made), real spawn; survivors go to the wave's own segment of the next
//                  path queue (ballot rank, no global counter)
//   k_scan         per-wave survivor counts -> segment offsets (path queue and shadow queue)
//   k_pt_shadow    visible() for every queued next-event segment: the sample's radiance term of this bounce
//   k_pt_resolve_terms   mono: per pixel, every sample's terms added innermost first and the samples in sample order (the reference's FP32 order)
//   k_pt_fold, k_pt_resolve   stereo (two passes share the term lists): a sample's radiance from its terms; then the per-pixel sum and the anaglyph blend
//   k_pack         bucket-major <-> row-major copies for the multi-GPU gather (dev_pack.hpp)
#pragma once
#include "dev_shade.hpp"
#include "dev_whitted.hpp"
#include "dev_queues.hpp"
#include <type_traits>

#ifndef FRAY_PRIMARY_WAVES
#define FRAY_PRIMARY_WAVES 5
#endif
#ifndef FRAY_WHITTED_WAVES
#define FRAY_WHITTED_WAVES 3   // measured with persistent waves (boxed / forest DOF16 / zaphod ms): 2 -> 16.7 / 28.9 / 0.34, 3 -> 15.2 / 26.1 / 0.35, 4 -> 15.0 / 26.3 / 0.42, 5 -> 15.4 / 26.2 / 0.53
#endif
#ifndef FRAY_WHITTED_REFILL
#define FRAY_WHITTED_REFILL 64  // idle lanes of a wave before they are handed new pixels (k_whitted); measured on dragon / smallpt Whitted: 1 -> 28.9 / 3.53 ms
#endif
#ifndef FRAY_CSG_WAVES
#define FRAY_CSG_WAVES 3        // the Cube / CSG kernel variants (flag bit 1): the CsgOp machine's state on top of the KD walk wants registers more than the chip wants waves --
                                // tests/scenes/csg_nested.fray 960x720 path traced: 111 ms at 4 waves/SIMD (k_pt_bounce<2> 128 VGPR, 203 spilled), 80 ms at 3 (168, 129), 89 ms at 2 (227, none);
                                // bokeh.fray (k_whitted<2>, at 3 either way) 66.3 / 67.5 / 69.9 ms (profiles/r04_experiments/README.md D)
#endif
// waves per SIMD a kernel is register-allocated for: `n` for the common variants, FRAY_CSG_WAVES for the Cube / CSG ones
// The COUNTING variants (flag bit 0: instrumentation, never timed) get one wave per SIMD less than their twins: their ten 64-bit counters per lane then
// fit in registers instead of scratch memory.  Round 5 met the path tracer's counting kernels rendering wrong pictures -- at full frame size even faulting --
// exactly when they kept spilled registers in scratch (profiles/r05_experiments/README.md H); spill-free they are right.
constexpr int waves_for(int st, int n) { return (st & 2) ? (FRAY_CSG_WAVES < n ? FRAY_CSG_WAVES : n) : ((st & 1) && n > 2 ? n - 1 : n); }
#ifndef FRAY_MT_EARLY
#define FRAY_MT_EARLY 160       // words drawn by one lane of a k_whitted wave at which the whole wave materialises its generator states
#endif
#ifndef FRAY_WH_SHADE_WAVES
#define FRAY_WH_SHADE_WAVES 4   // the wavefront's closest-hit + shading kernel: 128 VGPRs, 9 spilled; forest DOF 16 24.1 -> 20.5 ms against 3 waves (168 VGPRs)
#endif
#ifndef FRAY_SHADOW_WAVES
#define FRAY_SHADOW_WAVES 5     // the any-hit kernels at 96 VGPRs (13 / 8 spilled): boxed Whitted 12.0 -> 11.3 ms, headline -1 % against 4 waves
#endif
#ifndef FRAY_ANYHIT_WAVES_KD
#define FRAY_ANYHIT_WAVES_KD 4  // k_primary / k_wh_visible / k_pt_shadow beside KD meshes: the walk with its two-pass leaves wants 117-125 VGPRs; at 5 waves (96, 30-40 spilled, some
#endif                          // inside the child-test loop) boxed Whitted 12.1 ms, at 4 waves 8.0; dragon primary 0.80 -> 0.78, forest DOF 16 13.0 -> 12.2
#ifndef FRAY_BOUNCE_WAVES
#define FRAY_BOUNCE_WAVES 4   // waves per SIMD the bounce kernel is register-allocated for: 128 VGPRs, 2 spilled (3 waves: 129 VGPRs; headline 123.5 vs 116.2 ms)
#endif
#ifndef FRAY_BOUNCE_WAVES_NOKD
#define FRAY_BOUNCE_WAVES_NOKD 5   // the variants without the KD walk need 100-106 VGPRs: 5 waves/SIMD at 94-96, 0-7 spilled (headline 111.8 -> 108.7 ms against 4 waves)
#endif
#ifndef FRAY_WHITTED_WAVES_KD
#define FRAY_WHITTED_WAVES_KD 2   // k_whitted beside KD meshes (and no Cube / CSG): 211 VGPR, nothing spilled; dragon Whitted 17.1 -> 16.4 ms against 3 waves (168 VGPR, 82 spilled),
#endif                            // 1 wave: 18.0; the Cube / CSG variants (bokeh) measure the same at 2 and 3 and stay at 3
constexpr int whitted_waves(int st) { return waves_for(st, (st & 6) == 4 ? FRAY_WHITTED_WAVES_KD : FRAY_WHITTED_WAVES); }
#ifndef FRAY_WHITTED_CHILD_WAVES_KD
#define FRAY_WHITTED_CHILD_WAVES_KD 3   // dragon Whitted 7.80 ms at 2 (180 VGPRs), 7.04 at 3 (168, 17 spilled), 6.98 at 4 (128, 105 spilled): pass B of the speculative fans (k_whitted<.., 2>: no pixel, no sample, no generator state) beside KD meshes
#endif
#ifndef FRAY_WHITTED_AC_WAVES_KD
#define FRAY_WHITTED_AC_WAVES_KD 3      // passes A and C beside KD meshes: dragon Whitted 6.45 ms at 2, 6.33 at 3 (72 / 91 spilled), 6.56 at 4
#endif
constexpr int whitted_waves(int st, int mode) { return (st & 6) != 4 || mode == 0 ? whitted_waves(st) : mode == 2 ? FRAY_WHITTED_CHILD_WAVES_KD : FRAY_WHITTED_AC_WAVES_KD; }
constexpr int primary_waves(int st) { return waves_for(st, kd_variant(st) ? FRAY_ANYHIT_WAVES_KD : FRAY_PRIMARY_WAVES); }
constexpr int anyhit_waves(int st) { return waves_for(st, kd_variant(st) ? FRAY_ANYHIT_WAVES_KD : FRAY_SHADOW_WAVES); }

// ---- kernel arguments, read where they are used ------------------------------------------------------------
// A kernel's by-value arguments are all loaded at its entry and kept alive; in the big kernels most of them (scene tables, queue
// arrays: ~150 scalar registers) end up spilled to VGPR lanes and cost a v_readlane -- a vector-ALU slot -- at every use.  These kernels
// therefore take ONE struct and read its fields through the kernarg segment (scalar loads, served by the scalar cache) inside the loop:
// kernel_args() returns the segment's address as a value the compiler cannot see through, once per iteration, so that nothing read
// through it is hoisted back out of the loop.  k_pt_bounce: 147 -> 13 spilled SGPRs, headline frame 125.4 -> 122.2 ms.
template <class A> FD const FRAY_RO A* kernel_args()
{
    const FRAY_RO A* p = (const FRAY_RO A*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return p;
}
#define KARG(A, P, f) (*(const decltype(A::f)*)(&(P)->f))

// ---- work item -> pixel ------------------------------------------------------------------------
// Inside a 48x48 bucket the items run over 8x8 pixel tiles (6x6 of them), so the 64 lanes of a wave
// hold a square of neighbouring pixels: their camera rays walk the same part of a KD-tree.  Every
// (pixel, sample) is independent, so the order has no effect on the picture.
FD bool item_pixel(const DFrame& F, int item, int& x, int& y)
{
    int k = item / 2304, local = item - k * 2304;
    int b = F.bucketFirst + k * F.bucketStride;
    int by = b / F.BW, bx = (b - by * F.BW + FRAYHIP_BUCKET_SKEW * by) % F.BW;      // frayhip_bucket_xy (include/frayhip.h)
    int tile = local >> 6, in = local & 63;
    x = bx * 48 + (tile % 6) * 8 + (in & 7);
    y = by * 48 + (tile / 6) * 8 + (in >> 3);
    return x < F.W && y < F.H;
}
// Persistent waves.  The work items of a frame form nItems / 64 tiles (one 8x8 pixel tile = one wave's
// worth; nItems is a multiple of 2304, hence of 64), split into 8 contiguous ranges with one cursor
// each (DCursors: one 128-byte line per cursor, zeroed by the host with the counters).  A wave claims
// one tile at a time, first from the range of its block's XCD (blocks go round-robin over the 8
// XCDs, so an XCD's L2 sees one part of the picture and of the KD-trees), then from the others in
// turn.  One cursor for the whole frame would serialise ~30 k atomics on one address (~6 ns each),
// which a 0.25 ms frame notices; eight run in parallel.  `r` counts the ranges this wave has seen
// exhausted (start at 0); after the eighth the wave leaves, so the grid always drains.
FD int claim_tile(DCursors* cur, int nTiles, int& r)            // a 64-item tile for this wave, or -1 (wave-uniform)
{
    const int per = (nTiles + 7) >> 3;
    const int home = (int)(blockIdx.x & 7);
    while (r < 8) {
        const int range = (home + r) & 7;
        unsigned int t = 0;
        if ((threadIdx.x & 63) == 0) t = atomicAdd(&cur->v[range][0], 1u);
        t = (unsigned int)__builtin_amdgcn_readfirstlane((int)t);
        const int tile = range * per + (int)t;
        if (t < (unsigned int)per && tile < nTiles) return tile;
        r++;
    }
    return -1;
}
// The same for kernels whose launches may be small: with `cur` null the wave walks the tiles with a fixed stride instead (a frame of a few tiles per
// wave pays more for the atomics -- ~30 k of them on eight addresses, and eight failing ones per wave at the end -- than it can win from balance:
// zaphod 800x600, 0.31 ms with the stride, 0.45 ms with claims).  `tile` is the wave's previous tile, -1 at the start.
FD int next_tile(DCursors* cur, int nTiles, int& r, int tile)
{
    if (cur) return claim_tile(cur, nTiles, r);
    const int w = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6), nW = (int)((gridDim.x * blockDim.x) >> 6);
    tile = tile < 0 ? w : tile + nW;
    return tile < nTiles ? tile : -1;
}
FD int claim_items(DCursors* cur, int nItems, int& r)
{
    const int tile = claim_tile(cur, nItems >> 6, r);
    return tile < 0 ? nItems : tile * 64 + (int)(threadIdx.x & 63);
}

FD void flush_stats(DStats* st, const Cnt& c)
{
    atomicAdd(&st->closest, c.closest); atomicAdd(&st->shadow, c.shadow); atomicAdd(&st->node, c.node);
    atomicAdd(&st->kdInner, c.kdInner); atomicAdd(&st->leafRefs, c.leafRefs); atomicAdd(&st->tri, c.tri);
    atomicAdd(&st->prim, c.prim); atomicAdd(&st->smooth, c.smooth); atomicAdd(&st->samples, c.samples);
    atomicAdd(&st->tex, c.tex);
}
FD Cnt zero_cnt() { Cnt c; c.closest = c.shadow = c.node = c.kdInner = c.leafRefs = c.tri = c.prim = c.smooth = c.samples = c.tex = 0; c.envelope = 0; return c; }

// ---- camera (camera.cpp:59-92) ------------------------------------------------------------------
// which: 0 = CAMERA_CENTER, 1 = CAMERA_LEFT, 2 = CAMERA_RIGHT
FD void screen_ray(const DCamera& C, double x, double y, V3& o, V3& d, int which = 0)
{
    V3 tl = ld3(C.topLeft);
    d = tl + (ld3(C.topRight) - tl) * (x / C.w) + (ld3(C.bottomLeft) - tl) * (y / C.h);
    d = normalized(d);
    o = ld3(C.pos);
    if (which == 1) o = o + ld3(C.rightDir) * -C.stereoSeparation;
    else if (which == 2) o = o + ld3(C.rightDir) * C.stereoSeparation;
}
template <class G>
FD void dof_ray(const DCamera& C, double x, double y, G& tab, V3& o, V3& d, int which = 0)
{
    screen_ray(C, x, y, o, d, which);
    double M = C.focalPlaneDist / dot(ld3(C.frontDir), d);
    V3 T = ld3(C.pos) + d * M;
    double u, v;
    rng_unit_disc(tab, u, v);
    u *= C.apertureSize;
    v *= C.apertureSize;
    o = o + (u * ld3(C.rightDir) + v * ld3(C.upDir));
    d = normalized(T - o);
}

// maxTraceDepth < 0: raytrace() / pathtrace() return black before they look at the scene (main.cpp:173-176, 248), every pixel of the frame is 0
static __global__ __launch_bounds__(256) void k_black(DFrame F, int nItems, int eyes, float* __restrict__ rgb, DStats* st)
{
    unsigned long long n = 0;
    for (int item = blockIdx.x * blockDim.x + threadIdx.x; item < nItems; item += gridDim.x * blockDim.x) {
        int x, y;
        if (!item_pixel(F, item, x, y)) continue;
        size_t p = ((size_t)y * F.W + x) * 3;
        rgb[p] = 0; rgb[p + 1] = 0; rgb[p + 2] = 0;
        n += (unsigned long long)(F.spp * eyes);
    }
    if (n) atomicAdd(&st->samples, n);
}

// ---- mt19937 seeding for a batch of camera samples -------------------------------------------------
// x397[s * nItems + item] = x[397] of the seeding recurrence started at the contract seed of
// (pixel(item), sample s0 + s).  The recurrence is a 397-long dependency chain of
// shift / xor / 32-bit multiply / add, so each lane runs several independent chains at once and the
// kernel keeps its register count low enough for full occupancy.
#ifndef FRAY_SEED_CHAINS
#define FRAY_SEED_CHAINS 4   // 4 and 8 measure the same (1.73 ms per 44 M seeds): the kernel is bound by v_mul_lo_u32 issue, not latency
#endif
static __global__ __launch_bounds__(256) void k_seed(DFrame F, int nItems, int s0, int chunk, uint32_t* __restrict__ x397)
{
    constexpr int NC = FRAY_SEED_CHAINS;
    const uint32_t total = (uint32_t)nItems * (uint32_t)chunk;
    const uint32_t groups = (total + NC - 1u) / NC;
    for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < groups; q += gridDim.x * blockDim.x) {
        uint32_t b[NC], slot[NC];
#pragma unroll
        for (int k = 0; k < NC; k++) {
            // strided so that consecutive lanes write consecutive words
            slot[k] = q + (uint32_t)k * groups;
            b[k] = 0;
            if (slot[k] < total) {
                int item = (int)(slot[k] % (uint32_t)nItems), s = (int)(slot[k] / (uint32_t)nItems);
                int x, y;
                item_pixel(F, item, x, y);
                b[k] = sample_seed(F.seed, (uint32_t)y * (uint32_t)F.W + (uint32_t)x, (uint32_t)(s0 + s));
            }
        }
#pragma unroll 1
        for (uint32_t i = 1; i <= 397; i++) {
#pragma unroll
            for (int k = 0; k < NC; k++) b[k] = mt_lcg(b[k], i);
        }
#pragma unroll
        for (int k = 0; k < NC; k++) if (slot[k] < total) x397[slot[k]] = b[k];
    }
}

// ---- MODE_PRIMARY_ID ------------------------------------------------------------------------------
struct PrimaryArgs { DScene S; DCamera C; DFrame F; int nItems; int32_t* hitId; double* hitDist; DStats* st; DCursors* cur; };
template <int ST>
static __global__ __launch_bounds__(256, primary_waves(ST)) void k_primary(PrimaryArgs A)
{
    Cnt c = zero_cnt();
    const int nItems = A.nItems;
    DCursors* const cur = A.cur;
#ifdef FRAY_STAMPS
    stamp_begin();
#endif
    for (int r = 0, item = claim_items(cur, nItems, r); item < nItems; item = claim_items(cur, nItems, r)) {
        STAMP(0);
        const FRAY_RO PrimaryArgs* AP = kernel_args<PrimaryArgs>();
        const DScene& S = KARG(PrimaryArgs, AP, S);
        const DCamera& C = KARG(PrimaryArgs, AP, C);
        const DFrame& F = KARG(PrimaryArgs, AP, F);
        int x, y;
        if (!item_pixel(F, item, x, y)) continue;
        V3 o, d;
        screen_ray(C, (double)x, (double)y, o, d);
        HitT<ST> h;
        closest_hit<ST>(S, o, d, h, c);
        size_t p = (size_t)y * F.W + x;
        int32_t* const hitId = KARG(PrimaryArgs, AP, hitId);
        double* const hitDist = KARG(PrimaryArgs, AP, hitDist);
        if (hitId) hitId[p] = h.node;
        if (hitDist) hitDist[p] = h.dist;
        STAMP(13);
    }
#ifdef FRAY_STAMPS
    if ((threadIdx.x & 63) < 24) { atomicAdd(&A.st->stamp[threadIdx.x & 63], g_stampAcc[threadIdx.x >> 6][threadIdx.x & 63]); atomicAdd(&A.st->stampLanes[threadIdx.x & 63], g_stampLanes[threadIdx.x >> 6][threadIdx.x & 63]); }
#endif
    if (ST & 1) flush_stats(A.st, c);
    if ((ST & 2) && c.envelope) atomicAdd(&A.st->rngOverflow, 1ull);
}

static __constant__ double kAAOffsets[5][2] = {{0, 0}, {0.6, 0}, {0.3, 0.3}, {0, 0.6}, {0.6, 0.6}};   // main.cpp:55-61

// Whitted, scenes with recursive shaders (Reflection / Refraction / Layered): raytrace() per camera sample, one lane walks the whole shade()
// tree (dev_whitted.hpp).  Scenes without them take the wavefront path below (k_wh_shade ...).
// Persistent waves with PER-LANE refill: a lane that has finished its pixel takes the next one from the wave's pool (64 work items claimed
// at a time from the XCD-affine cursors), so a lane whose pixel shows a 25-sample glossy floor does not keep 63 finished lanes waiting -- every
// round of the machine has a closest-hit search to run for (nearly) every lane until the frame's items are gone.
#ifdef FRAY_TILESTAT
// diagnostic build only (tools/tilestat_run.sh): when did which wave hold which 8x8 tile?  [tile] = {start, end (s_memtime), global wave, kernel start}
static __device__ unsigned long long g_tileStat[65536][8];   // + rounds, cheap-step iterations, lanes x rounds standing at a search, at a direct-light loop
FD unsigned long long tile_now() { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory"); return t; }
#endif
// Speculative glossy fans (MODE 1-3 of k_whitted).  Reflection::shade with glossiness < 1 at depth 0 (shading.cpp:172-204) traces numSamples rays one
// after the other, and the generator makes them a chain: sample i's direction is drawn after whatever sample i - 1's raytrace() drew.  With persistent
// waves that chain set the duration of the whole kernel (hw9/dragon.fray: 26 rounds of a search and a shadow ray per floor pixel, every round as slow as
// the wave's slowest lane; 1 % of the tiles took 6-14 ms each and the chip stood 64 % empty behind them).  Where the scene's lights draw nothing (point
// lights only) the rays under such a fan usually draw nothing either, and then the fan's directions can be drawn ahead:
//   pass A (MODE 1)  every camera sample as usual; a lane that reaches its sample's first depth-0 glossy activation draws the fan's directions back to
//                    back, as they would come if no child drew, files them (one entry per sample, one child per direction) and drops the sample
//   pass B (MODE 2)  every child is a raytrace() of its own (depth 1) on a generator that only records whether it was used; result + "drew nothing"
//   pass C (MODE 3)  the filed samples again from their seeds -- the same draws, the same pushes -- except that the fan's children are looked up while
//                    they drew nothing; the first child that did draw is traced in place and so is everything after it, with the true generator state
// Colours are added in the reference's order in pass C, by the same code; a child's colour does not depend on who traced it.  The counting variants
// (flag bit 0) run MODE 0: their counters are the reference's call counts, and a re-run sample would count its camera ray twice.
static __global__ void k_add4(const int* __restrict__ a, unsigned long long* __restrict__ sum) { if (threadIdx.x < 4) sum[threadIdx.x] += (unsigned long long)a[threadIdx.x]; }   // a frame's totals over its batches may pass 2^31
struct WhittedArgs { DScene S; DCamera C; DFrame F; int nItems; int s0, cn; float* rgb; float* rad; uint32_t* mtWork; const uint32_t* x397; DStats* st; DCursors* cur; SpecBuf sp; };
template <int ST, int MODE>
static __global__ __launch_bounds__(256, whitted_waves(ST, MODE)) void k_whitted(WhittedArgs A)
{
    Cnt c = zero_cnt();
    MtLong tab;
    tab.stride = gridDim.x * blockDim.x;
    tab.st = A.mtWork + (blockIdx.x * blockDim.x + threadIdx.x);
    MtSpy spy;
    spy.x = 0; spy.used = false;
    // A work item is one camera sample: (pixel item, sample s0 + k), k-major, so that the 64 lanes of a wave hold the SAME sample of an 8x8 pixel
    // tile and a pixel's samples -- independent given their seeds -- spread over the chip instead of queueing up in one lane (bokeh.fray,
    // 640x480 with 45 lens samples: 4 800 tiles of 45 samples each for 3 072 waves left the chip 22 % occupied).  The per-pixel sum in
    // sample order is k_pt_resolve's, as for the other integrators; a frame with one sample per pixel writes its pixel itself.
    // Pass B's items are the filed children, pass C's the filed samples (their numbers are on the device: no host round trip between the passes).
    const int nItems = A.nItems;
    const int nTot = MODE == 2 ? A.sp.counters[1] : MODE == 3 ? A.sp.counters[0] : A.nItems * A.cn;
    DCursors* const cur = A.cur;
    DStats* const st = A.st;
    const uint32_t lane = threadIdx.x & 63u;
    __shared__ double rightRay[6][256];
    WhittedLane L;
    L.mode = WM_NEXT_PIXEL; L.sp = 0;
    SpecLane SL;
    SL.state = 2; SL.sp = 0; SL.base = 0; SL.looked = 0; SL.missed = 0;
    // the lane's pixel and camera sample
    int item = 0, x = 0, y = 0, k = 0, eye = 0;
    C3 cl = c3(0, 0, 0);
    bool ovf = false;
    Mt rnd = tab.r;
    // the wave's pool of claimed work items [poolNext, poolEnd) and the claim state (wave-uniform)
    int poolNext = 0, poolEnd = 0, claimR = 0;
#ifdef FRAY_STAMPS
    stamp_begin();
#endif
#ifdef FRAY_TILESTAT
    const unsigned long long tsKernel = tile_now();
    int tsTile = -1;
    unsigned long long tsRounds = 0, tsCheap = 0, tsTrace = 0, tsDirect = 0;
#endif
    for (;;) {
        const FRAY_RO WhittedArgs* AP = kernel_args<WhittedArgs>();
        const DScene& S = KARG(WhittedArgs, AP, S);
        const DCamera& C = KARG(WhittedArgs, AP, C);
        const DFrame& F = KARG(WhittedArgs, AP, F);
        const SpecBuf& SP = KARG(WhittedArgs, AP, sp);
        const bool stereo = C.stereoSeparation > 0;
        // ---- cheap steps, until every lane stands at a search, a direct-light loop, or has nothing left
        for (;;) {
            // Lanes that want a pixel wait until FRAY_WHITTED_REFILL of them do (or nobody has anything else to do): a wave keeps working on
            // neighbouring pixels -- its rays walk the same parts of the KD-trees -- instead of filling every free lane at once with pixels from elsewhere
            if (MODE == 2 && spy.used && L.mode < WM_NEXT_SAMPLE) {           // this child drew: pass C will trace it in place
                SP.cok[item] = 0;
                spy.used = false;
                L.sp = 0;
                L.mode = WM_NEXT_PIXEL;
            }
            const unsigned long long need = __ballot(L.mode == WM_NEXT_PIXEL);
            const unsigned long long busy = __ballot(L.mode != WM_NEXT_PIXEL && L.mode != WM_EXHAUSTED);
            const bool refill = need && ((int)__popcll(need) >= FRAY_WHITTED_REFILL || !busy);
            // a lane's generator is about to leave its 227-word register window: every lane of the wave that is inside a camera sample makes its
            // full state now, together (MtLong::materialise)
            if (MODE != 2 && __any(L.mode < WM_NEXT_SAMPLE && tab.idx < 0 && tab.r.j >= FRAY_MT_EARLY)) {
                if (L.mode < WM_NEXT_SAMPLE && tab.idx < 0) tab.materialise();
            }
            const bool cheap = L.mode == WM_NEXT_PIXEL ? refill : (L.mode == WM_ROOT_RET || L.mode == WM_NEXT_SAMPLE || (L.mode < WM_ROOT_RET && wl_cheap(S, L)));
            if (!__any(cheap)) break;
#ifdef FRAY_TILESTAT
            tsCheap++;
#endif
            // work items for the lanes that need one: from the wave's pool, refilled one tile of 64 at a time
            if (refill) {
                if (poolNext == poolEnd && claimR < 8) {
                    const int tile = claim_tile(cur, (nTot + 63) >> 6, claimR);
                    if (tile >= 0) { poolNext = tile * 64; poolEnd = poolNext + 64 < nTot ? poolNext + 64 : nTot; }
#ifdef FRAY_TILESTAT
                    {
                        const unsigned long long t = tile_now();
                        if (lane == 0) {
                            if (tsTile >= 0 && tsTile < 65536) { g_tileStat[tsTile][1] = t; g_tileStat[tsTile][4] = tsRounds; g_tileStat[tsTile][5] = tsCheap; g_tileStat[tsTile][6] = tsTrace; g_tileStat[tsTile][7] = tsDirect; }
                            tsRounds = tsCheap = tsTrace = tsDirect = 0;
                            tsTile = tile;
                            if (tsTile >= 0 && tsTile < 65536) { g_tileStat[tsTile][0] = t; g_tileStat[tsTile][2] = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; g_tileStat[tsTile][3] = tsKernel; }
                        }
                    }
#endif
                }
                const int have = poolEnd - poolNext;
                const int rank = (int)__popcll(need & ((1ull << lane) - 1ull));
                if (L.mode == WM_NEXT_PIXEL) {
                    if (rank < have) {
                        int slot = poolNext + rank;
                        if (MODE == 2) {
                            item = slot;                                       // the child
                            L.mode = WM_NEXT_SAMPLE;
                        } else {
                            if (MODE == 3) { SL.base = SP.eChildBase[slot]; slot = SP.eSlot[slot]; }
                            k = slot / nItems;                                 // wave-uniform in passes 0 and A: nItems is a multiple of 64
                            item = slot - k * nItems;
                            if (item_pixel(F, item, x, y)) L.mode = WM_NEXT_SAMPLE;
                            // a slot of a ragged edge bucket outside the frame: nothing to render, ask again
                        }
                    } else if (have == 0 && claimR >= 8) {
                        L.mode = WM_EXHAUSTED;
                    }
                }
                const int want = (int)__popcll(need);
                poolNext += want < have ? want : have;
            }
            if (L.mode == WM_NEXT_SAMPLE) {
                if (MODE == 2) {                                              // a filed child: raytrace(ray) at depth 1
                    const int e = SP.cEntry[item];
                    wl_start(L, v3(SP.eo[0][e], SP.eo[1][e], SP.eo[2][e]), v3(SP.cd[0][item], SP.cd[1][item], SP.cd[2][item]));
                    L.depth = 1;
                    L.spBase = SP.cok[item];                                  // filed by pass A: a child that would overflow the stack in place must overflow here too
                    spy.used = false;
                } else {
                    const int i = KARG(WhittedArgs, AP, s0) + k;
                    const uint32_t* const x397 = KARG(WhittedArgs, AP, x397);
                    const uint32_t p = (uint32_t)y * (uint32_t)F.W + (uint32_t)x;
                    tab.reseed_with(sample_seed(F.seed, p, (uint32_t)i), x397[(size_t)k * nItems + item]);
                    rnd = tab.r;
                    float ox, oy;
                    if (F.jitter) { ox = rng_float(rnd); oy = rng_float(rnd); }
                    else { ox = (float)kAAOffsets[i][0]; oy = (float)kAAOffsets[i][1]; }
                    const double fx = (double)((float)x + ox), fy = (double)((float)y + oy);   // int + float, main.cpp:359
                    V3 o, d;
                    if (stereo) {                                             // raytraceSinglePixel, main.cpp:306-317: both rays first, then left, then right
                        V3 orr, dr;
                        if (C.dof) { dof_ray(C, fx, fy, tab, o, d, 1); dof_ray(C, fx, fy, tab, orr, dr, 2); }
                        else { screen_ray(C, fx, fy, o, d, 1); screen_ray(C, fx, fy, orr, dr, 2); }
                        // the right eye's ray waits in LDS (the thread's own slots) while the left eye's tree is walked
                        rightRay[0][threadIdx.x] = orr.x; rightRay[1][threadIdx.x] = orr.y; rightRay[2][threadIdx.x] = orr.z;
                        rightRay[3][threadIdx.x] = dr.x; rightRay[4][threadIdx.x] = dr.y; rightRay[5][threadIdx.x] = dr.z;
                        bump<ST>(c.samples, 2);
                    } else {
                        if (C.dof) dof_ray(C, fx, fy, tab, o, d); else screen_ray(C, fx, fy, o, d);
                        bump<ST>(c.samples);
                    }
                    eye = 0;
                    wl_start(L, o, d);
                    SL.state = (MODE == 1 || MODE == 3) && !stereo ? 0 : 2;
                }
            } else if (L.mode == WM_ROOT_RET) {                              // the camera ray's (pass B: the child's) raytrace() returned
                if (MODE == 2) {
                    SP.cc[0][item] = L.ret.r; SP.cc[1][item] = L.ret.g; SP.cc[2][item] = L.ret.b;
                    SP.cok[item] = spy.used ? 0 : 1;
                    spy.used = false;
                    L.mode = WM_NEXT_PIXEL;
                } else if (stereo && eye == 0) {
                    cl = L.ret;
                    eye = 1;
                    wl_start(L, v3(rightRay[0][threadIdx.x], rightRay[1][threadIdx.x], rightRay[2][threadIdx.x]),
                             v3(rightRay[3][threadIdx.x], rightRay[4][threadIdx.x], rightRay[5][threadIdx.x]));
                } else {
                    C3 smp = L.ret;                                           // this camera sample's colour
                    if (stereo) {
                        C3 cr = L.ret;
                        if (S.saturation != 1) {                              // Color::adjustSaturation, color.h:127-133
                            float ml = (cl.r + cl.g + cl.b) / 3.0f, mr = (cr.r + cr.g + cr.b) / 3.0f;
                            cl = c3(ml + (cl.r - ml) * S.saturation, ml + (cl.g - ml) * S.saturation, ml + (cl.b - ml) * S.saturation);
                            cr = c3(mr + (cr.r - mr) * S.saturation, mr + (cr.g - mr) * S.saturation, mr + (cr.b - mr) * S.saturation);
                        }
                        smp = cl * ldc(C.leftMask) + cr * ldc(C.rightMask);
                    }
                    ovf = ovf || rnd.j > 227;
                    if (F.spp == 1) {                                         // vfb[y][x] = (0 + sample) / 1, main.cpp:348-360
                        const C3 avg = (c3(0, 0, 0) + smp) / 1.0f;
                        const size_t q = ((size_t)y * F.W + x) * 3;
                        float* const rgb = KARG(WhittedArgs, AP, rgb);
                        rgb[q] = avg.r; rgb[q + 1] = avg.g; rgb[q + 2] = avg.b;
                    } else {                                                  // k_pt_resolve adds the pixel's samples in their order
                        float* const rad = KARG(WhittedArgs, AP, rad);
                        const size_t q = ((size_t)k * nItems + item) * 3;
                        rad[q] = smp.r; rad[q + 1] = smp.g; rad[q + 2] = smp.b;
                    }
                    L.mode = WM_NEXT_PIXEL;
                }
            } else if (L.mode < WM_ROOT_RET && wl_cheap(S, L)) {
                if (MODE == 2) {
                    bool deep = false;
                    wl_cheap_step<ST, MtSpy, 0>(S, L, spy, c, deep, SP, SL);
                    if (deep) spy.used = true;                               // not answered ahead: pass C traces this child in place, where the sample ends as the reference's would
                }
                else wl_cheap_step<ST, MtLong, MODE>(S, L, tab, c, ovf, SP, SL);
            }
        }
        STAMP(12);
        if (!__any(L.mode != WM_EXHAUSTED)) break;
        if (MODE == 1) {
            // ---- pass A: the lanes standing at a fan to file (WM_DEFER).  One pair of atomics per wave: entries by rank, children by prefix sum.
            const bool def = L.mode == WM_DEFER;
            const unsigned long long dm = __ballot(def);
            if (dm) {
                const int cnt = def ? L.stack[L.sp - 1].count : 0;
                int incl = cnt;
#pragma unroll
                for (int dlt = 1; dlt < 64; dlt <<= 1) { const int t = __shfl_up(incl, dlt); if ((int)lane >= dlt) incl += t; }
                const int total = __builtin_amdgcn_readlane(incl, 63);
                int e0 = 0, c0 = 0;
                if (lane == 0) { e0 = atomicAdd(&SP.counters[0], (int)__popcll(dm)); c0 = atomicAdd(&SP.counters[1], total); }
                e0 = __builtin_amdgcn_readfirstlane(e0); c0 = __builtin_amdgcn_readfirstlane(c0);
                if (def) {
                    const WFrame& f = L.stack[L.sp - 1];
                    const FRAY_RO DShader& sh = S.shaders[f.shader];
                    const int e = e0 + (int)__popcll(dm & ((1ull << lane) - 1ull)), cb = c0 + incl - cnt;
                    SP.eSlot[e] = k * nItems + item; SP.eChildBase[e] = cb;
                    const V3 n = faceforward(f.d, f.info.norm);
                    const V3 org = f.info.ip + n * 1e-6;
                    SP.eo[0][e] = org.x; SP.eo[1][e] = org.y; SP.eo[2][e] = org.z;
                    V3 bb, cc;
                    orthonormalSystem(n, bb, cc);
                    for (int q = 0; q < cnt; q++) {                           // the draws of shading.cpp:172-204, with nothing drawn in between
                        V3 reflected;
                        int draws = 0;
                        for (;;) {
                            double dx, dy;
                            rng_unit_disc(tab, dx, dy);
                            draws++;
                            dx *= sh.deflectionScaling;
                            dy *= sh.deflectionScaling;
                            const V3 nn = normalized(n + bb * dx + cc * dy);
                            reflected = reflect(f.d, nn);
                            if (dot(reflected, n) > 0) break;
                        }
                        SP.cEntry[cb + q] = e;
                        SP.cdraws[cb + q] = (unsigned char)(draws <= 255 ? draws : 0);
                        SP.cok[cb + q] = (unsigned char)L.sp;                 // for pass B: the activations the child finds on the stack in place (the fan's own included)
                        SP.cd[0][cb + q] = reflected.x; SP.cd[1][cb + q] = reflected.y; SP.cd[2][cb + q] = reflected.z;
                    }
                    L.sp = 0;
                    L.mode = WM_NEXT_PIXEL;                                   // pass C renders this sample
                }
            }
        }
#ifdef FRAY_TILESTAT
        tsRounds++; tsTrace += __popcll(__ballot(L.mode == WM_TRACE)); tsDirect += __popcll(__ballot(L.mode == WM_SHADE));
#endif
        if (L.mode == WM_TRACE) wl_trace_step<ST>(S, L, c);
        STAMP(14);
        if (L.mode == WM_SHADE && S.shaders[L.shader].kind <= 2) {
            if (MODE == 2) wl_direct_step<ST, MtSpy>(S, L, spy, c);
            else wl_direct_step<ST, MtLong>(S, L, tab, c);
        }
        STAMP(10);
    }
#ifdef FRAY_STAMPS
    if (lane < 24) { atomicAdd(&st->stamp[lane], g_stampAcc[threadIdx.x >> 6][lane]); atomicAdd(&st->stampLanes[lane], g_stampLanes[threadIdx.x >> 6][lane]); }
#endif
    if (MODE == 3) {                                                        // how the speculation went (frayhip_scene_get_option)
        int a = SL.looked, b = SL.missed;
#pragma unroll
        for (int dlt = 32; dlt > 0; dlt >>= 1) { a += __shfl_down(a, dlt); b += __shfl_down(b, dlt); }
        if (lane == 0 && (a | b)) { atomicAdd(&A.sp.counters[2], a); atomicAdd(&A.sp.counters[3], b); }
    }
    if (ovf && MODE != 2) atomicAdd(&st->rngOverflow, 1ull);
    if (ST & 1) flush_stats(st, c);
    if ((ST & 2) && c.envelope) atomicAdd(&st->rngOverflow, 1ull);
}

// ---- Whitted as a wavefront, for scenes whose shaders do not recurse ---------------------------------------------
// raytrace() (main.cpp:246-285) + Lambert::shade / Phong::shade (shading.cpp:48-144) split where they call visible():
//   k_wh_shade    per camera sample: camera ray(s), closest hit, attributes, bump; the light loops draw every sample in the
//                 reference's order and queue, per light sample, the segment to test and the term it would add
//   k_wh_visible  visible() for every queued segment (the lean any-hit kernel: no shading state in registers)
//   k_wh_gather   per camera sample: result = base + sum over lights of (sum of the visible samples' terms) / ns, the reference's
//                 FP32 order; then k_pt_resolve sums the samples of a pixel in order
// visible() draws no random numbers, so evaluating it after the light loops changes nothing.
// FUSED: visible() is asked in place and the sample's colour returned (`fused`), in Lambert / Phong::shade's own order of additions (shading.cpp:54-78) -- for
// scenes whose lights take a handful of samples per hit (zaphod, forest: ONE point light), where the three launches of the wavefront cost more than the
// shadow rays they lay side by side.  Nothing is queued then.
template <int ST, bool FUSED = false>
FD void wh_shade_eye(const DScene& S, V3 o, V3 d, MtLong& tab, const WhittedQueue& Q, size_t N, size_t e, Cnt& c, C3* fused = nullptr)
{
    HitT<ST> h;
    closest_hit<ST>(S, o, d, h, c);
    C3 base;
    unsigned char hit = 0;
    if (h.node <= -2) base = light_color(S.lights[-2 - h.node]);
    else if (h.node < 0) base = environment<ST>(S, d, c);
    else {
        const FRAY_RO DNode& N0 = S.nodes[h.node];
        const FRAY_RO DShader& sh = S.shaders[N0.shader];
        HitInfo info;
        finalize_hit<ST>(S, h, o, d, sh.usesUV || N0.bumpTex >= 0, info);
        apply_bump<ST>(S, h.node, info, c);
        if (sh.kind == 0) base = ldc(sh.color);                           // ConstantShader::shade
        else {                                                            // Lambert::shade / Phong::shade up to visible()
            const bool phong = sh.kind == 2;
            C3 diffuse = ldc(sh.color);
            if (sh.texture >= 0) diffuse = diffuse * texture_sample<ST>(S, sh.texture, d, info, c);
            base = diffuse * ldc(S.ambient);
            hit = 1;
            const V3 n = faceforward(d, info.norm);
            const V3 a = info.ip + n * 1e-6;
            if constexpr (!FUSED) { Q.ax[e] = a.x; Q.ay[e] = a.y; Q.az[e] = a.z; }
            size_t t = e;
            const int nl = S.nLights;
            for (int li = 0; li < nl; li++) {
                const FRAY_RO DLight& L = S.lights[li];
                const int ns = light_num_samples(L);
                C3 sum = c3(0, 0, 0);
                for (int k = 0; k < ns; k++, t += N) {
                    C3 lc;
                    V3 lp;
                    light_nth_sample(L, k, info.ip, tab, lp, lc);
                    double d2 = lengthSqr(info.ip - lp);
                    V3 wl = normalized(lp - info.ip);
                    float cosN = (float)dot(wl, n);
                    float lam = (float)(cosN / d2);
                    lam = lam > 0.0f ? lam : 0.0f;   // max(0.0f, x)
                    C3 r = diffuse * lc * lam;
                    if (phong) {
                        V3 wi = -wl;
                        V3 rr = reflect(wi, n);
                        double cosV = dot(-d, rr);
                        if (cosV > 0)
                            r = r + lc / (float)d2 * ldc(sh.specularColor) * (float)pow(cosV, sh.exponent) * (float)sh.specularMultiplier;
                    }
                    if constexpr (FUSED) {
                        if (visible<ST>(S, a, lp, c)) sum = sum + r;
                    } else {
                        Q.bx[t] = lp.x; Q.by[t] = lp.y; Q.bz[t] = lp.z;
                        Q.rr[t] = r.r; Q.rg[t] = r.g; Q.rb[t] = r.b;
                    }
                }
                if constexpr (FUSED) base = base + sum / (float)ns;             // result += sum / numSamples, light after light (k_wh_gather's order)
            }
        }
    }
    if constexpr (FUSED) { *fused = base; return; }
    Q.base[3 * e] = base.r; Q.base[3 * e + 1] = base.g; Q.base[3 * e + 2] = base.b;
    Q.hit[e] = hit;
}

// radL / radR / rgb: the fused form's outputs -- the samples' colours for k_pt_resolve, or, for a mono frame of one sample per pixel, the pixel itself (rgb != nullptr).
// x397 == nullptr (fused form only): no generator of this frame can be asked for a word (no jitter, no lens, no sampling light), none is seeded.
struct WhShadeArgs { DScene S; DCamera C; DFrame F; int nItems, s0, chunk; WhittedQueue Q; uint32_t* mtWork; const uint32_t* x397; DStats* st; DCursors* cur; float* radL; float* radR; float* rgb; };
template <int ST, bool FUSED = false>
static __global__ __launch_bounds__(256, waves_for(ST, FRAY_WH_SHADE_WAVES)) void k_wh_shade(WhShadeArgs A)
{
    Cnt c = zero_cnt();
    MtLong tab;
    tab.stride = gridDim.x * blockDim.x;
    tab.st = A.mtWork + (blockIdx.x * blockDim.x + threadIdx.x);
    const int nItems = A.nItems, s0 = A.s0;
    DStats* const st = A.st;
    const uint32_t total = (uint32_t)nItems * (uint32_t)A.chunk;
    const bool stereo = A.C.stereoSeparation > 0;
    const size_t N = (size_t)total * (stereo ? 2 : 1);
    bool ovf = false;
    __shared__ double rightRay[6][256];
#ifdef FRAY_STAMPS
    stamp_begin();
#endif
    // persistent waves claiming 64-slot tiles (next_tile; total is a multiple of 64): with a fixed stride from slot to slot the waves that drew the picture's expensive tiles
    // finished last and the chip stood 18-23 % empty (forest DOF: 3.1-3.3 of 4 waves per SIMD resident on average)
    DCursors* const cur = A.cur;
    for (int r = 0, tile = next_tile(cur, (int)(total >> 6), r, -1); tile >= 0; tile = next_tile(cur, (int)(total >> 6), r, tile)) {
        const uint32_t slot = (uint32_t)tile * 64u + (threadIdx.x & 63u);
        STAMP(13);
        const FRAY_RO WhShadeArgs* AP = kernel_args<WhShadeArgs>();
        const DScene& S = KARG(WhShadeArgs, AP, S);
        const DCamera& C = KARG(WhShadeArgs, AP, C);
        const DFrame& F = KARG(WhShadeArgs, AP, F);
        const WhittedQueue& Q = KARG(WhShadeArgs, AP, Q);
        const uint32_t* const x397 = KARG(WhShadeArgs, AP, x397);
        const int item = (int)(slot % (uint32_t)nItems), s = (int)(slot / (uint32_t)nItems);
        int x, y;
        if (!item_pixel(F, item, x, y)) {                 // ragged edge bucket: nothing to shade, nothing to test
            if constexpr (!FUSED) {
                for (int eye = 0; eye < (stereo ? 2 : 1); eye++) {
                    const size_t e = (size_t)eye * total + slot;
                    Q.hit[e] = 0;
                    Q.base[3 * e] = 0; Q.base[3 * e + 1] = 0; Q.base[3 * e + 2] = 0;
                }
            }
            continue;
        }
        const int i = s0 + s;
        const uint32_t p = (uint32_t)y * (uint32_t)F.W + (uint32_t)x;
        if (!FUSED || x397) tab.reseed_with(sample_seed(F.seed, p, (uint32_t)i), x397[slot]);
        Mt rnd = tab.r;
        float ox, oy;
        if (F.jitter) { ox = rng_float(rnd); oy = rng_float(rnd); }
        else { ox = (float)kAAOffsets[i][0]; oy = (float)kAAOffsets[i][1]; }
        const double fx = (double)((float)x + ox), fy = (double)((float)y + oy);   // int + float, main.cpp:359
        V3 o, d;
        if (stereo) {                                     // raytraceSinglePixel, main.cpp:306-317: both rays first, then left, then right
            V3 orr, dr;
            if (C.dof) { dof_ray(C, fx, fy, tab, o, d, 1); dof_ray(C, fx, fy, tab, orr, dr, 2); }
            else { screen_ray(C, fx, fy, o, d, 1); screen_ray(C, fx, fy, orr, dr, 2); }
            // the right eye's ray waits in LDS (the thread's own slots) while the left eye is traced and shaded: no register holds it
            rightRay[0][threadIdx.x] = orr.x; rightRay[1][threadIdx.x] = orr.y; rightRay[2][threadIdx.x] = orr.z;
            rightRay[3][threadIdx.x] = dr.x; rightRay[4][threadIdx.x] = dr.y; rightRay[5][threadIdx.x] = dr.z;
            bump<ST>(c.samples, 2);
        } else {
            if (C.dof) dof_ray(C, fx, fy, tab, o, d); else screen_ray(C, fx, fy, o, d);
            bump<ST>(c.samples);
        }
        if (!FUSED || x397) ovf = ovf || rnd.j > 227;              // (an unseeded generator -- nothing of this frame draws -- holds no count)
        for (int eye = 0; eye < (stereo ? 2 : 1); eye++) {          // one copy of the trace-and-shade code for both eyes
            if (eye == 1) {
                o = v3(rightRay[0][threadIdx.x], rightRay[1][threadIdx.x], rightRay[2][threadIdx.x]);
                d = v3(rightRay[3][threadIdx.x], rightRay[4][threadIdx.x], rightRay[5][threadIdx.x]);
            }
            STAMP(0);
            if constexpr (FUSED) {
                C3 col;
                wh_shade_eye<ST, true>(S, o, d, tab, Q, N, (size_t)eye * total + slot, c, &col);
                float* const rgb = KARG(WhShadeArgs, AP, rgb);
                if (rgb) {                                            // vfb[y][x] = (0 + sample) / 1, main.cpp:348-360
                    const C3 avg = (c3(0, 0, 0) + col) / 1.0f;
                    const size_t q = ((size_t)y * F.W + x) * 3;
                    rgb[q] = avg.r; rgb[q + 1] = avg.g; rgb[q + 2] = avg.b;
                } else {
                    float* const out = (eye == 0 ? KARG(WhShadeArgs, AP, radL) : KARG(WhShadeArgs, AP, radR)) + 3 * (size_t)slot;
                    out[0] = col.r; out[1] = col.g; out[2] = col.b;
                }
            } else {
                wh_shade_eye<ST>(S, o, d, tab, Q, N, (size_t)eye * total + slot, c);
            }
            STAMP(10);
        }
    }
#ifdef FRAY_STAMPS
    if ((threadIdx.x & 63) < 24) { atomicAdd(&st->stamp[threadIdx.x & 63], g_stampAcc[threadIdx.x >> 6][threadIdx.x & 63]); atomicAdd(&st->stampLanes[threadIdx.x & 63], g_stampLanes[threadIdx.x >> 6][threadIdx.x & 63]); }
#endif
    if (ovf) atomicAdd(&st->rngOverflow, 1ull);
    if (ST & 1) flush_stats(st, c);
    if ((ST & 2) && c.envelope) atomicAdd(&st->rngOverflow, 1ull);
}

struct WhVisibleArgs { DScene S; WhittedQueue Q; size_t N; int T; DStats* st; DCursors* cur; };
template <int ST>
static __global__ __launch_bounds__(256, anyhit_waves(ST)) void k_wh_visible(WhVisibleArgs A)
{
    Cnt c = zero_cnt();
    const size_t N = A.N;
    DStats* const st = A.st;
    const size_t total = N * (size_t)A.T;
#ifdef FRAY_STAMPS
    stamp_begin();
#endif
    // persistent waves claiming 64-test tiles, as k_wh_shade does (total <= 2^31: render_impl's batch size)
    DCursors* const cur = A.cur;
    const int nTiles = (int)((total + 63) >> 6);
    for (int r = 0, tile = next_tile(cur, nTiles, r, -1); tile >= 0; tile = next_tile(cur, nTiles, r, tile)) {
        const size_t t = (size_t)tile * 64 + (threadIdx.x & 63);
        if (t >= total) continue;
        STAMP(13);
        const FRAY_RO WhVisibleArgs* AP = kernel_args<WhVisibleArgs>();
        const DScene& S = KARG(WhVisibleArgs, AP, S);
        const WhittedQueue& Q = KARG(WhVisibleArgs, AP, Q);
        const size_t e = t % N;
        if (!Q.hit[e]) continue;
        STAMP(0);
        Q.vis[t] = visible<ST>(S, v3(Q.ax[e], Q.ay[e], Q.az[e]), v3(Q.bx[t], Q.by[t], Q.bz[t]), c) ? 1 : 0;
    }
#ifdef FRAY_STAMPS
    if ((threadIdx.x & 63) < 24) { atomicAdd(&st->stamp[threadIdx.x & 63], g_stampAcc[threadIdx.x >> 6][threadIdx.x & 63]); atomicAdd(&st->stampLanes[threadIdx.x & 63], g_stampLanes[threadIdx.x >> 6][threadIdx.x & 63]); }
#endif
    if (ST & 1) flush_stats(st, c);
    if ((ST & 2) && c.envelope) atomicAdd(&st->rngOverflow, 1ull);
}

static __global__ __launch_bounds__(256) void k_wh_gather(DScene S, WhittedQueue Q, size_t N, size_t slots, float* __restrict__ radL, float* __restrict__ radR)
{
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < N; e += (size_t)gridDim.x * blockDim.x) {
        C3 result = c3(Q.base[3 * e], Q.base[3 * e + 1], Q.base[3 * e + 2]);
        if (Q.hit[e]) {
            size_t t = e;
            const int nl = S.nLights;
            for (int li = 0; li < nl; li++) {
                const int ns = light_num_samples(S.lights[li]);
                C3 sum = c3(0, 0, 0);
                for (int k = 0; k < ns; k++, t += N)
                    if (Q.vis[t]) sum = sum + c3(Q.rr[t], Q.rg[t], Q.rb[t]);
                result = result + sum / (float)ns;
            }
        }
        float* out = e < slots ? radL + 3 * e : radR + 3 * (e - slots);
        out[0] = result.r; out[1] = result.g; out[2] = result.b;
    }
}

// ---- path tracer (main.cpp:171-244) as a wavefront ---------------------------------------------------
// Path state, structure of arrays (one lane = one path, consecutive lanes = consecutive entries,
// so every array is read and written fully coalesced).  92 bytes per path; radiance goes to the sample's term
// list (TermBuf, dev_queues.hpp), one term per bounce, folded innermost-first by k_pt_fold.
template <class G>
struct PathStateT {
    V3 o, d;
    C3 pm;
    uint32_t slot;
    int depth;
    unsigned flags;
    G rnd, tab;        // Mt, or MtPath when a path may draw more than 227 words (dev_rng.hpp)
};
typedef PathStateT<Mt> PathState;
FD Mt& cursor(Mt& g) { return g; }
FD Mt& cursor(MtPath& g) { return g.r; }
FD const Mt& cursor(const Mt& g) { return g; }
FD const Mt& cursor(const MtPath& g) { return g.r; }

// Direction class of a ray for the consumers' coherence sort (sort_share below): the octant of its direction in Gray-code order (neighbouring
// classes differ in ONE sign), and, as the high bit, whether it may enter one of the scene's gates (ray_gate_class).  A scheduling key, nothing else.
// Cube / CSG variants (ST & 2): what costs there is a CsgOp node's machine, which only the lanes that enter that node's box run -- so a ray that may enter a gate
// is classed by WHICH gate (`gate` = 1 + the first gate it may enter, 0 = none): the few rays of a share that run the same machine then sit in the same wave.
template <int ST>
FD uint32_t ray_sort_class(V3 d, uint32_t gate)
{
    const uint32_t g = (d.x < 0 ? 1u : 0u) | (d.y < 0 ? 2u : 0u) | (d.z < 0 ? 4u : 0u);
    if constexpr ((ST & 2) != 0) { if (gate) return 8u + (gate - 1u < 6u ? gate - 1u : 6u); }
    return (g ^ (g >> 1) ^ (g >> 2)) | (gate ? 8u : 0u);
}
// STORE_CLS: only the variants whose consumers sort (sort_variant) read the class array
template <bool STORE_CLS, class G>
FD void path_store(const PathQueue& Q, uint32_t i, const PathStateT<G>& s, uint32_t cls)
{
    PathRec* r = Q.rec + i;
    r->o[0] = s.o.x; r->o[1] = s.o.y; r->o[2] = s.o.z;
    r->d[0] = s.d.x; r->d[1] = s.d.y; r->d[2] = s.d.z;
    r->pm[0] = s.pm.r; r->pm[1] = s.pm.g; r->pm[2] = s.pm.b;
    r->slot = s.slot;
    r->depthFlags = (uint32_t)s.depth | (s.flags << 16);
    const Mt &g = cursor(s.rnd), &t = cursor(s.tab);
    r->rnd[0] = g.j; r->rnd[1] = g.a; r->rnd[2] = g.b;
    r->tab[0] = t.j; r->tab[1] = t.a; r->tab[2] = t.b;
    if constexpr (STORE_CLS) Q.cls[i] = (unsigned char)cls;
}
template <class G>
FD void path_load_ray(const PathQueue& Q, uint32_t i, PathStateT<G>& s)
{
    const PathRec* r = Q.rec + i;
    s.o = v3(r->o[0], r->o[1], r->o[2]);
    s.d = v3(r->d[0], r->d[1], r->d[2]);
}
// Everything but the ray: fetched after the closest-hit search so it is not live across it.
template <class G>
FD void path_load_rest(const PathQueue& Q, uint32_t i, PathStateT<G>& s)
{
    const PathRec* r = Q.rec + i;
    s.pm = c3(r->pm[0], r->pm[1], r->pm[2]);
    s.slot = r->slot;
    uint32_t df = r->depthFlags;
    s.depth = (int)(df & 0xffffu);
    s.flags = df >> 16;
    Mt &g = cursor(s.rnd), &t = cursor(s.tab);
    g.j = r->rnd[0]; g.a = r->rnd[1]; g.b = r->rnd[2];
    t.j = r->tab[0]; t.a = r->tab[1]; t.b = r->tab[2];
}

// ---- segmented path queues ---------------------------------------------------------------------------
// A single global append counter is the bottleneck of a wavefront tracer on this chip: one word
// sustains ~88 atomics/us, and a 1080p x 64spp frame needs ~8 M wave-level appends (measured: the
// counter alone cost ~90 ms of a 190 ms frame).  So there is no global counter.  Every wave of the
// producing kernel owns a contiguous range of input paths and writes its survivors, ranked by a
// ballot prefix count, into its own contiguous segment of the output queue (capacity = its input
// share, so it cannot overflow); it publishes one count.  A one-block scan turns the counts into
// offsets; a consuming wave owns a contiguous range of dense indices and follows the segments that hold them
// with a cursor (seg_map below).
static __global__ __launch_bounds__(1024) void k_scan(QMeta* m0, QMeta* m1)
{
    QMeta* m = blockIdx.x == 0 ? m0 : m1;
    __shared__ uint32_t part[1024];
    const uint32_t nSeg = m->nSeg;
    const uint32_t per = (nSeg + 1023u) / 1024u;
    const uint32_t b = threadIdx.x * per;
    uint32_t sum = 0;
    for (uint32_t k = 0; k < per; k++) if (b + k < nSeg) sum += m->cnt[b + k];
    part[threadIdx.x] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {      // Hillis-Steele inclusive scan
        uint32_t v = threadIdx.x >= d ? part[threadIdx.x - d] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - sum;         // exclusive prefix of this thread's slice
    for (uint32_t k = 0; k < per; k++)
        if (b + k < nSeg) { m->off[b + k] = run; run += m->cnt[b + k]; }
    if (threadIdx.x == 1023) { m->off[nSeg] = part[1023]; m->n = part[1023]; }
}

// Queue 0 of a batch is dense: a single segment that holds every slot.
static __global__ void k_meta_dense(QMeta* m, uint32_t n)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) { m->n = n; m->chunk = n; m->nSeg = 1; m->certFront = 0; m->nf[0] = n; m->off[0] = 0; m->off[1] = n; }
}

// Dense index -> storage index of a segmented queue without a table in LDS: a wave walks a contiguous range of
// dense indices, so the segment that holds them only ever moves forward.  `seg` is the wave's cursor (uniform);
// the offsets are read on the scalar path.  seg_map returns the storage index of dense entry `di` of the 64-entry
// batch that starts at `base` (garbage for lanes that are not `live`).
FD uint32_t seg_first(const FRAY_RO uint32_t* off, uint32_t nSeg, uint32_t begin)
{
    uint32_t lo = 0, hi = nSeg;       // largest s with off[s] <= begin
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (off[mid] <= begin) lo = mid; else hi = mid;
    }
    return lo;
}
FD uint32_t seg_slot_of(uint32_t s, uint32_t chunk, uint32_t e, uint32_t nf) { return s * chunk + (e < nf ? e : chunk - 1u - (e - nf)); }
// `front`: the entry sits at the FRONT of its segment, i.e. its producer filed it as gate-free (bit 31 of the result; storage indices stay below 2^31).
#define FRAY_FRONT_BIT 0x80000000u
FD uint32_t seg_map(const FRAY_RO uint32_t* off, const FRAY_RO uint32_t* nf, uint32_t nSeg, uint32_t chunk, uint32_t base, uint32_t di, bool live, uint32_t& seg)
{
    while (seg + 1 < nSeg && off[seg + 1] <= base) seg++;
    uint32_t i = 0, s = seg, lo = off[s];
    for (;;) {                        // the segments that overlap this batch: one, now and then two
        const uint32_t hi = off[s + 1];
        if (live && di >= lo && di < hi) i = seg_slot_of(s, chunk, di - lo, nf[s]) | (di - lo < nf[s] ? FRAY_FRONT_BIT : 0u);
        if (hi >= base + 64u || s + 1 >= nSeg) break;
        s++;
        lo = hi;
    }
    return i;
}

// ---- coherence sort (round 5) ------------------------------------------------------------------------------
// A wave of the bounce / shadow kernels used to take the 64 next entries of its share as they came: rays of unrelated directions, so that in the
// tree-less triangle loops (cornell_box: 34 triangles a ray, 10 of them the walls') every early exit of the triangle test -- backface culling first
// (mesh.cpp:106) -- had some lane that did not take it, and the wave ran every test to the end.  Now a wave first orders FRAY_SORT_N entries of its
// share by the class their producer stamped on them (ray_sort_class: gate bit, direction octant), by a counting sort in LDS (sixteen counters, two
// passes of LDS atomics, the storage indices in sortBuf), and then works through them in that order: waves whose lanes agree on the signs of their
// directions leave a back-facing wall after seven instructions, all together.  Which lane traces which ray changes no ray's result (every path carries
// its sample slot, the queues' internal order was arbitrary before): pictures and counters are bit for bit what they were.
#ifndef FRAY_SORT
#define FRAY_SORT 1
#endif
#ifndef FRAY_SORT_N
#define FRAY_SORT_N 1024      // entries sorted at a time: 4 KB of LDS per wave
#endif
// Which kernel variants sort: the ones that walk KD-trees (boxed.fray path traced at 960 x 540 x 16 spp: 65.3 -> 54.9 ms).  Without a tree the order costs more than it saves:
// cornell_box's bounce kernel issues 11 % fewer instructions sorted and takes 3 % less time, its shadow kernel 11 % MORE time (two passes of LDS atomics and segment look-ups
// per entry, against ~600 instructions a shadow ray costs there), smallpt -- planes and spheres, nothing to leave early -- +31 % / +70 %; the Cube / CSG variants +3 %.
#ifdef FRAY_SORT_BY_MATERIAL
// Experiment (profiles/r05_experiments/README.md, "binning by material"): EVERY variant sorts, and a path's class is the shader kind that spawned its ray
// (Lambert / mirror / glass / other) instead of its direction -- north_star's sort-by-material, as far as a queue between bounces can know a material.
constexpr bool sort_variant(int) { return true; }
#else
// FRAY_SORT_CSG (off): the Cube / CSG variants sorting too, a ray's class being WHICH gate it may enter (ray_sort_class<ST>).  Measured on csg_nested.fray path
// traced: 55.9 -> 62.0 ms; the share of a wave that enters a CsgOp machine stays 0.38 (0.39 unsorted) and the lanes inside it 0.23: a wave's share of the queue holds a
// dozen rays per object, and inside a machine a ray that hits asks sixty plain geometries where one that finds nothing asks six.  Regrouping the gate rays of all
// producer segments first (a second dense order out of k_scan, balanced slices per wave) was built as well: bit-identical, +1.5 ... +28 % (profiles/r05_experiments/README.md J).
#ifndef FRAY_SORT_CSG
#define FRAY_SORT_CSG 0
#endif
constexpr bool sort_variant(int st) { return (st & 6) == 4 || (FRAY_SORT_CSG && (st & 2) != 0); }
#endif
FD void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// keyOf(i): the class (0..15) of storage entry i.  On return buf[0 .. b1 - b0) holds the storage indices of dense entries [b0, b1) in class order and
// `seg` stands where seg_map left it after b1.
template <class KeyOf>
FD void sort_share(const FRAY_RO uint32_t* off, const FRAY_RO uint32_t* nf, uint32_t nSeg, uint32_t chunk, uint32_t b0, uint32_t b1, uint32_t& seg,
                   uint32_t* buf, uint32_t* cnt, KeyOf keyOf)
{
    const uint32_t lane = threadIdx.x & 63u;
    if (lane < 16) cnt[lane] = 0;
    wave_lds_sync();
    uint32_t s1 = seg;
    for (uint32_t base = b0; base < b1; base += 64u) {
        const uint32_t di = base + lane;
        const bool live = di < b1;
        const uint32_t i = seg_map(off, nf, nSeg, chunk, base, di, live, s1) & ~FRAY_FRONT_BIT;
        if (live) atomicAdd(cnt + keyOf(i), 1u);
    }
    wave_lds_sync();
    {   // exclusive prefix of the sixteen counts (lanes 0..15)
        const uint32_t v = lane < 16 ? cnt[lane] : 0u;
        uint32_t inc = v;
        for (int d = 1; d < 16; d <<= 1) { const uint32_t t = __shfl_up(inc, d, 64); if ((int)lane >= d) inc += t; }
        wave_lds_sync();
        if (lane < 16) cnt[lane] = inc - v;
    }
    wave_lds_sync();
    uint32_t s2 = seg;
    for (uint32_t base = b0; base < b1; base += 64u) {
        const uint32_t di = base + lane;
        const bool live = di < b1;
        const uint32_t raw = seg_map(off, nf, nSeg, chunk, base, di, live, s2);     // (the front bit travels with the index)
        if (live) buf[atomicAdd(cnt + keyOf(raw & ~FRAY_FRONT_BIT), 1u)] = raw;
    }
    seg = s2;
    wave_lds_sync();
}

// Stereo path tracing (raytraceSinglePixel, main.cpp:306-317): both eye rays are generated first, the
// left path is traced, and the right path CONTINUES both random generators where the left path stopped.
// So the left pass parks the right eye's ray and, when a left path ends, its generator cursors, per sample
// slot; the right pass starts from those.  g[0] == nullptr: nothing to save (mono, or the right pass).
// `own`: the path's last term, written here; without it the term of this bounce is the next-event contribution, which
// k_pt_shadow (or path_shade, when no segment was queued) writes.
template <class G>
FD void path_finish(const TermBuf& TB, DStats* st, const PathStateT<G>& s, const StereoBuf& SB)
{
    const Mt &r = cursor(s.rnd), &t = cursor(s.tab);
    if (SB.g[0]) {
        SB.g[0][s.slot] = r.j; SB.g[1][s.slot] = r.a; SB.g[2][s.slot] = r.b;
        SB.g[3][s.slot] = t.j; SB.g[4][s.slot] = t.a; SB.g[5][s.slot] = t.b;
    }
    TB.n[s.slot] = (unsigned short)(TB.b + 1);
    // the three-register streams end after 227 words; paths that may draw more run the MtPath variant of the bounce kernel
    if (sizeof(G) == sizeof(Mt) && (r.j > 227 || t.j > 227)) atomicAdd(&st->rngOverflow, 1ull);
}
FD void term_store(const TermBuf& TB, uint32_t slot, C3 v)
{
    const size_t q = (size_t)TB.b * 3 * TB.nPaths + slot;          // planar: consecutive slots are consecutive words
    TB.t[q] = v.r; TB.t[q + TB.nPaths] = v.g; TB.t[q + 2 * (size_t)TB.nPaths] = v.b;
}

// Batch = nItems pixels x `chunk` samples starting at sample s0; slot = s * nItems + item.  The first
// queue is dense (entry = slot); slots of pixels outside the frame (ragged edge buckets) are marked dead.
#define FRAY_DEAD 0xffffffffu
template <int ST>
static __global__ __launch_bounds__(256) void k_pt_init(DScene S, DCamera C, DFrame F, int nItems, int s0, int chunk, PathQueue Q,
                                                 unsigned short* __restrict__ termCount, const uint32_t* __restrict__ x397, StereoBuf SB, int eye, DStats* st)
{
    Cnt c = zero_cnt();
    const uint32_t total = (uint32_t)nItems * (uint32_t)chunk;
    const bool stereo = C.stereoSeparation > 0;
    for (uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x; slot < total; slot += gridDim.x * blockDim.x) {
        int item = (int)(slot % (uint32_t)nItems), s = (int)(slot / (uint32_t)nItems);
        int x, y;
        if (item_pixel(F, item, x, y)) {
            PathState ps;
            if (eye == 0) {
                const uint32_t p = (uint32_t)y * (uint32_t)F.W + (uint32_t)x;
                ps.rnd = mt_seed_with(sample_seed(F.seed, p, (uint32_t)(s0 + s)), x397[slot]);
                ps.tab = ps.rnd;
                float ox = rng_float(ps.rnd), oy = rng_float(ps.rnd);           // gi: always jittered (main.cpp:351-353)
                double fx = (double)((float)x + ox), fy = (double)((float)y + oy);
                const int which = stereo ? 1 : 0;
                if (C.dof) dof_ray(C, fx, fy, ps.tab, ps.o, ps.d, which); else screen_ray(C, fx, fy, ps.o, ps.d, which);
                if (stereo) {                                                   // the right eye's ray, drawn before any tracing
                    V3 ro, rd;
                    if (C.dof) dof_ray(C, fx, fy, ps.tab, ro, rd, 2); else screen_ray(C, fx, fy, ro, rd, 2);
                    SB.r[0][slot] = ro.x; SB.r[1][slot] = ro.y; SB.r[2][slot] = ro.z;
                    SB.r[3][slot] = rd.x; SB.r[4][slot] = rd.y; SB.r[5][slot] = rd.z;
                }
            } else {
                ps.rnd.j = SB.g[0][slot]; ps.rnd.a = SB.g[1][slot]; ps.rnd.b = SB.g[2][slot];
                ps.tab.j = SB.g[3][slot]; ps.tab.a = SB.g[4][slot]; ps.tab.b = SB.g[5][slot];
                ps.o = v3(SB.r[0][slot], SB.r[1][slot], SB.r[2][slot]);
                ps.d = v3(SB.r[3][slot], SB.r[4][slot], SB.r[5][slot]);
            }
            ps.pm = c3(1, 1, 1);
            ps.slot = slot;
            ps.depth = 0;
            ps.flags = 0;
            bump<ST>(c.samples);
            path_store<FRAY_SORT && sort_variant(ST)>(Q, slot, ps, ray_sort_class<ST>(ps.d, 0u));
        } else {
            // no path in this slot: a zero direction says so to the bounce kernel (which then needs no load beyond the ray's own 48 bytes to know)
            PathRec* r = Q.rec + slot;
            r->d[0] = 0; r->d[1] = 0; r->d[2] = 0;
            r->depthFlags = FRAY_DEAD;
            if constexpr (FRAY_SORT && sort_variant(ST)) Q.cls[slot] = 15;               // sorted last
        }
        termCount[slot] = 0;
    }
    if (ST & 1) flush_stats(st, c);
}

// ---- one bounce = trace, shade, shadow -------------------------------------------------------------------
// Every wave of these three kernels owns the same contiguous range of the input queue's dense indices (its
// share = the segment capacity of what it writes), and walks it in order.
struct WaveShare { uint32_t begin, end, chunk, w, W; };
FD WaveShare wave_share(uint32_t n)
{
    WaveShare r;
    r.W = gridDim.x * (blockDim.x >> 6);
    r.w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    r.chunk = (((n + r.W - 1u) / r.W) + 63u) & ~63u;
    r.begin = r.w * r.chunk;
    r.end = r.begin + r.chunk < n ? r.begin + r.chunk : n;
    if (r.begin > n) r.begin = r.end = n;
    return r;
}

// Which end of its wave's queue segment a ray goes to (QMeta, dev_queues.hpp): true = it may enter one of the scene's gates (DGate: the boxes
// of the meshes with long brute-force triangle loops).  A slab test in FP32 on the world-space box -- a scheduling hint, nothing else.
// Round 5: when every gate of the scene is EXACT (DGate::exact: untransformed nodes) the test is dev_misscert.hpp's FP32 certificate instead, "gate-free" is
// then proven, and the consumers skip the gated nodes for the rays filed at the front (closest_hit / visible, `gateFree`).
// Returns 0 for a gate-free ray, else 1 + the index of the first gate it may enter.
FD uint32_t ray_gate_class(const DScene& S, V3 o, V3 d)
{
    const int ng = S.nGates;
    if (ng == 0) return 0u;
    const float ox = (float)o.x, oy = (float)o.y, oz = (float)o.z;
    if (S.gatesExact) {
        const float dx = (float)d.x, dy = (float)d.y, dz = (float)d.z;
        const float omax = fmaxf(fmaxf(fabsf(ox), fabsf(oy)), fabsf(oz)), dsum = fabsf(dx) + fabsf(dy) + fabsf(dz);
        uint32_t first = 0;
        for (int g = ng - 1; g >= 0; g--) {
            const FRAY_RO DGate& G = S.gates[g];
            if (!ray_surely_misses_box_f32(G.cf[0], G.cf[1], G.cf[2], G.hf[0], G.hf[1], G.hf[2], G.Mf, ox, oy, oz, dx, dy, dz, omax, dsum)) first = (uint32_t)g + 1u;
        }
        return !(omax < 1e9f) && !first ? 1u : first;
    }
    const float rx = __builtin_amdgcn_rcpf((float)d.x), ry = __builtin_amdgcn_rcpf((float)d.y), rz = __builtin_amdgcn_rcpf((float)d.z);
    uint32_t first = 0;
    for (int g = ng - 1; g >= 0; g--) {
        const FRAY_RO DGate& G = S.gates[g];
        const float ax = ((float)G.lo[0] - ox) * rx, bx = ((float)G.hi[0] - ox) * rx;
        const float ay = ((float)G.lo[1] - oy) * ry, by = ((float)G.hi[1] - oy) * ry;
        const float az = ((float)G.lo[2] - oz) * rz, bz = ((float)G.hi[2] - oz) * rz;
        const float t0 = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), 0.0f));
        const float t1 = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
        if (!(t0 > t1 * 1.0001f + 1e-3f)) first = (uint32_t)g + 1u;          // NaNs (a direction component of 0 on a box face) count as "may enter"
    }
    return first;
}

// Where the lanes of one batch of 64 append to a wave's queue segment: the gate-free rays at the front (in order), the others at the back (in
// reverse), ranked by ballot -- no global counter.  `nF` / `nB`: entries so far at either end (wave-uniform).
struct SegEnds { uint32_t begin, chunk, nF, nB; };
// The same for appends made deep inside divergent code (the next-event segment is written where path_shade samples it, so that its fifteen
// registers are not carried to the end of the iteration): there the lanes of a batch may pass the append in several groups, so a ballot rank is not
// safe (two groups would both count from zero: measured -- 2 % of the segments of the instrumented kernel variant overwrote each other).  The
// wave's two counters live in LDS and every appending lane takes its own slot with an LDS atomic; the order inside the segment is then
// arbitrary, which changes nothing (every path carries its own sample slot).
struct SegEndsShared { uint32_t begin, chunk; uint32_t* n; };        // n[0] front count, n[1] back count: this wave's pair in LDS
FD uint32_t seg_take(const SegEndsShared& E, bool back)
{
    const uint32_t k = atomicAdd(E.n + (back ? 1 : 0), 1u);
    return back ? E.begin + E.chunk - 1u - k : E.begin + k;
}
// the slot of a lane that appends (callable where only the appending lanes of the batch are active: the counts are not touched) ...
FD uint32_t seg_slot(const SegEnds& E, bool put, bool back)
{
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long mF = __ballot(put && !back), mB = __ballot(put && back);
    const unsigned long long below = (1ull << lane) - 1ull;
    return back ? E.begin + E.chunk - 1u - (E.nB + (uint32_t)__popcll(mB & below)) : E.begin + E.nF + (uint32_t)__popcll(mF & below);
}
// ... and the counts, advanced once per batch where the whole wave is together again
FD void seg_advance(SegEnds& E, bool put, bool back)
{
    E.nF += (uint32_t)__popcll(__ballot(put && !back));
    E.nB += (uint32_t)__popcll(__ballot(put && back));
}

// What pathtrace() does with a path once its closest hit is known (main.cpp:201-242): light / environment hits end
// it; otherwise bump, the discarded spawnRay, the next-event sample (everything but its visibility query -> `shadow`
// segment sa -> sb carrying sc), the real spawnRay, the throughput update and the entry test of the next iteration
// (`cont`: ps is the path to continue).
template <int ST, bool BARY, class G>
FD void path_shade(const DScene& S, PathStateT<G>& ps, const HitT<ST>& h, const TermBuf& TB, DStats* st, const StereoBuf& SB, bool& cont, bool& shadow, uint32_t& shadowBack,
                   const ShadowQueue& SQ, const SegEndsShared& shadowEnds, Cnt& c)
{
    C3 own = c3(0, 0, 0);          // this bounce's term, unless a queued next-event segment will provide it
    if (h.node <= -2) {                                       // main.cpp:201-208
        own = (ps.flags & RF_DIFFUSE) ? c3(0, 0, 0) : light_color(S.lights[-2 - h.node]) * ps.pm;
        path_finish(TB, st, ps, SB);
    } else if (h.node < 0) {                                  // main.cpp:210-215
        own = environment<ST>(S, ps.d, c) * ps.pm;
        path_finish(TB, st, ps, SB);
    } else {
        const FRAY_RO DNode& N = S.nodes[h.node];
        const FRAY_RO DShader& sh = S.shaders[N.shader];
        HitInfo info;
        finalize_hit<ST, BARY>(S, h, ps.o, ps.d, sh.usesUV || N.bumpTex >= 0, info);
        apply_bump<ST>(S, h.node, info, c);
        mt_skip(ps.tab, spawn_words(sh));                     // the discarded spawnRay (main.cpp:219-224)
        {
            // the next-event segment goes to the wave's shadow segment right here (ballot rank among the lanes that sampled a light), so
            // that it is not carried in registers across the spawn
            V3 sa, sb;
            C3 sc;
            shadow = nee_prepare(S, ps.d, info, ps.pm, sh, ps.rnd, ps.tab, sa, sb, sc);
            if (shadow) {
                shadowBack = ray_gate_class(S, sa, sb - sa);
                const uint32_t j = seg_take(shadowEnds, shadowBack != 0);
                SQ.ax[j] = sa.x; SQ.ay[j] = sa.y; SQ.az[j] = sa.z;
                SQ.bx[j] = sb.x; SQ.by[j] = sb.y; SQ.bz[j] = sb.z;
                SQ.cr[j] = sc.r; SQ.cg[j] = sc.g; SQ.cb[j] = sc.b;
                SQ.slot[j] = ps.slot;
                if constexpr (FRAY_SORT && sort_variant(ST)) SQ.cls[j] = (unsigned char)ray_sort_class<ST>(sb - sa, shadowBack);
            }
        }
        PathRay win, wout;
        win.o = ps.o; win.d = ps.d; win.depth = ps.depth; win.flags = ps.flags;
        C3 brdf;
        float pdf;
        spawn_ray(sh, info, win, ps.tab, wout, brdf, pdf);
        // main.cpp:238-239 returns without the light contribution when pdf is -1 or 0: no spawn_ray above produces either value
        // (1 / 2 pi, 1e9, 1), so the segment queued above always stands
        {
            ps.pm = ps.pm * brdf / pdf;
            ps.o = wout.o; ps.d = wout.d; ps.depth = wout.depth; ps.flags = wout.flags;
#ifdef FRAY_SORT_BY_MATERIAL
            ps.flags = (ps.flags & 0xffu) | ((sh.kind == 1 ? 0u : sh.kind == 3 ? 1u : sh.kind == 4 ? 2u : 3u) << 8);      // experiment: the spawning shader's kind rides in the flags
#endif
            // entry test of the next pathtrace() call (main.cpp:173-176): it returns black, this bounce's term stays the light contribution
            if (ps.depth > S.maxTraceDepth || intensity(ps.pm) < 0.01) path_finish(TB, st, ps, SB);
            else cont = true;
        }
    }
    if (!shadow) term_store(TB, ps.slot, own);
}

// **Dominant kernel**: one pathtrace() iteration (main.cpp:171-244) for every live path of the queue: closest hit,
// then path_shade.  Every wave owns a contiguous share of the queue's dense indices and writes its survivors and
// next-event segments into its own segments of the output queues.
// LONG: the generators are MtPath (paths that may draw more than 227 words, i.e. maxTraceDepth >= 20); `LR` says where a path's two
// 624-word columns live and how to recompute its seed (slot -> pixel, sample).
struct LongRng { uint32_t* cols; uint32_t nPaths; DFrame F; int nItems, s0; };
// FIRST: the batch's first bounce makes its own camera rays (what k_pt_init does for the other cases: stereo, long generators) instead of
// reading them from a queue -- no 92-byte path record written and read back per camera sample, one launch less per batch.
struct FirstArgs { DCamera C; DFrame F; int nItems, s0; uint32_t n; const uint32_t* x397; unsigned short* termCount; };
struct BounceArgs { DScene S; PathQueue Qin, Qout; ShadowQueue SQ; QMetaRO metaIn; QMeta* metaOut; QMeta* metaShadow; TermBuf TB; StereoBuf SB; LongRng LR; DStats* st; FirstArgs FA; };
// ARITH: how the translation unit was compiled -- 0 = the reference's arithmetic (-ffp-contract=off: every hit record and every colour is the CPU reference build's, bit
// for bit), 1 = -ffp-contract=fast (render_contract.hip: multiply-add pairs fused; only for rays AFTER a sample's first closest hit, i.e. colour within
// north_star's 1e-4 RMS, option "fp_contract").  The parameter only names the kernel apart in profiles; the code is the same source.
#ifndef FRAY_ARITH
#define FRAY_ARITH 0
#endif
template <int ST, bool LONG, bool FIRST = false, int ARITH = FRAY_ARITH>
static __global__ __launch_bounds__(256, waves_for(ST, kd_variant(ST) ? FRAY_BOUNCE_WAVES : FRAY_BOUNCE_WAVES_NOKD)) void k_pt_bounce(BounceArgs A)
{
    static_assert(!(LONG && FIRST), "long generators start from k_pt_init");
    typedef typename std::conditional<LONG, MtPath, Mt>::type G;
    Cnt c = zero_cnt();
    const QMetaRO metaIn = A.metaIn;
    QMeta* const metaOut = A.metaOut;
    QMeta* const metaShadow = A.metaShadow;
    DStats* const st = A.st;
    const FRAY_RO uint32_t* off = FIRST ? nullptr : metaIn.p->off;
    const FRAY_RO uint32_t* nfIn = FIRST ? nullptr : metaIn.p->nf;
    const uint32_t nSeg = FIRST ? 1u : metaIn.p->nSeg, chunkIn = FIRST ? A.FA.n : metaIn.p->chunk;
    const bool certFront = !FIRST && metaIn.p->certFront != 0;
    const WaveShare ws = wave_share(FIRST ? A.FA.n : metaIn.p->n);
    const uint32_t lane = threadIdx.x & 63u;
    SegEnds outEnds{ws.begin, ws.chunk, 0, 0};                                            // wave-uniform
    __shared__ uint32_t shadowCount[4][2];
    if (lane < 2) shadowCount[threadIdx.x >> 6][lane] = 0;
    const SegEndsShared shadowEnds{ws.begin, ws.chunk, shadowCount[threadIdx.x >> 6]};
    uint32_t seg = (!FIRST && ws.begin < ws.end) ? seg_first(off, nSeg, ws.begin) : 0;
#ifdef FRAY_STAMPS
    stamp_begin();
#endif
    constexpr bool SORT = FRAY_SORT && !FIRST && sort_variant(ST);
    const uint32_t sortN = (uint32_t)FRAY_SORT_N;
    for (uint32_t b0 = ws.begin, b1 = 0; b0 < ws.end; b0 = b1) {            // (a kernel that does not sort takes its whole share in one go)
    b1 = (SORT && ws.end - b0 > sortN) ? b0 + sortN : ws.end;
    uint32_t* sorted = nullptr;
    if constexpr (SORT) {
        __shared__ uint32_t sortBuf[4][FRAY_SORT_N];
        __shared__ uint32_t sortCnt[4][16];
        sorted = sortBuf[threadIdx.x >> 6];
        const unsigned char* const cls = KARG(BounceArgs, kernel_args<BounceArgs>(), Qin).cls;
        sort_share(off, nfIn, nSeg, chunkIn, b0, b1, seg, sorted, sortCnt[threadIdx.x >> 6], [&](uint32_t i) { return (uint32_t)cls[i] & 15u; });
    }
    for (uint32_t base = b0; base < b1; base += 64u) {
        const FRAY_RO BounceArgs* AP = kernel_args<BounceArgs>();
        const FirstArgs& FA = KARG(BounceArgs, AP, FA);
        const DScene& S = KARG(BounceArgs, AP, S);
        const PathQueue& Qin = KARG(BounceArgs, AP, Qin);
        const PathQueue& Qout = KARG(BounceArgs, AP, Qout);
        const ShadowQueue& SQ = KARG(BounceArgs, AP, SQ);
        const TermBuf& TB = KARG(BounceArgs, AP, TB);
        const StereoBuf& SB = KARG(BounceArgs, AP, SB);
        const LongRng& LR = KARG(BounceArgs, AP, LR);
        const uint32_t di = base + lane;
        bool cont = false, shadow = false, gateFree = false;
        uint32_t shadowBack = 0;
        PathStateT<G> ps;
        bool live = di < b1;
        uint32_t i = di, seed0 = 0;
        if constexpr (FIRST) {
            // the camera sample of slot di (k_pt_init's arithmetic): generators seeded by the contract seed, two jitter words, the lens sample
            if (live) {
                FA.termCount[di] = 0;
                int px, py;
                live = item_pixel(FA.F, (int)(di % (uint32_t)FA.nItems), px, py);      // slots of a ragged edge bucket outside the frame: no path
                if (live) {
                    seed0 = sample_seed(FA.F.seed, (uint32_t)py * (uint32_t)FA.F.W + (uint32_t)px, (uint32_t)(FA.s0 + (int)(di / (uint32_t)FA.nItems)));
                    Mt rnd = mt_seed_with(seed0, FA.x397[di]), tab = rnd;
                    const float ox = rng_float(rnd), oy = rng_float(rnd);                  // gi: always jittered (main.cpp:351-353)
                    const double fx = (double)((float)px + ox), fy = (double)((float)py + oy);
                    if (FA.C.dof) dof_ray(FA.C, fx, fy, tab, ps.o, ps.d, 0); else screen_ray(FA.C, fx, fy, ps.o, ps.d, 0);
                }
            }
        } else {
            if constexpr (SORT) i = live ? sorted[di - b0] : 0u;
            else i = seg_map(off, nfIn, nSeg, chunkIn, base, di, live, seg);
            gateFree = certFront && (i & FRAY_FRONT_BIT);
            i &= ~FRAY_FRONT_BIT;
            if (live) {
                path_load_ray(Qin, i, ps);
                live = !(ps.d.x == 0 && ps.d.y == 0 && ps.d.z == 0);            // k_pt_init's mark of a slot without a path
            }
        }
        if (live) {
            STAMP(0);
            // entry test of pathtrace() (main.cpp:173-176) was applied before this path was queued
            HitT<ST> h;
            closest_hit<ST>(S, ps.o, ps.d, h, c, gateFree);
            if constexpr (FIRST) {
                // the rest of the path's state, made after the search so that it is not live across it: the generators stand where the camera ray left them
                ps.pm = c3(1, 1, 1); ps.slot = di; ps.depth = 0; ps.flags = 0;
                ps.rnd = mt_seed_with(seed0, FA.x397[di]);
                ps.tab = ps.rnd;
                mt_skip(ps.rnd, 2);
                if (FA.C.dof) mt_skip(ps.tab, 4);
                bump<ST>(c.samples);
            } else {
                path_load_rest(Qin, i, ps);
            }
#ifdef FRAY_QCHECK
            // diagnostic build: a queue entry that no producer wrote (or that was consumed before) is counted and dropped
            if (ps.slot >= TB.nPaths) { atomicAdd(&st->rngOverflow, 1ull << 32); ps.slot = 0; ps.pm = c3(0, 0, 0); ps.depth = 0x7fff; }
            else Qin.rec[i].slot = 0xffffffffu;
#endif
            if constexpr (LONG) {
                int px, py;
                item_pixel(LR.F, (int)(ps.slot % (uint32_t)LR.nItems), px, py);
                const uint32_t sd = sample_seed(LR.F.seed, (uint32_t)py * (uint32_t)LR.F.W + (uint32_t)px, (uint32_t)(LR.s0 + (int)(ps.slot / (uint32_t)LR.nItems)));
                ps.rnd.seed = ps.tab.seed = sd;
                ps.rnd.stride = ps.tab.stride = LR.nPaths;
                ps.rnd.col = LR.cols + ps.slot;
                ps.tab.col = LR.cols + (size_t)624 * LR.nPaths + ps.slot;
            }
            STAMP(8);
            path_shade<ST, false>(S, ps, h, TB, st, SB, cont, shadow, shadowBack, SQ, shadowEnds, c);
            STAMP(10);
        }
        // survivors and next-event segments of this batch of 64 paths go to the wave's own segments of the output queues: gate-free rays to
        // the front, the others to the back (ballot ranks, no global counter)
        const uint32_t gate = cont ? ray_gate_class(S, ps.o, ps.d) : 0u;
        const bool back = gate != 0;
        const uint32_t slotOut = seg_slot(outEnds, cont, back);
#ifdef FRAY_SORT_BY_MATERIAL
        if (cont) path_store<true>(Qout, slotOut, ps, (ps.flags >> 8) & 3u);
#else
        if (cont) path_store<FRAY_SORT && sort_variant(ST)>(Qout, slotOut, ps, ray_sort_class<ST>(ps.d, gate));
#endif
        seg_advance(outEnds, cont, back);
        STAMP(13);
    }
    }
#ifdef FRAY_STAMPS
    if (lane < 24) { atomicAdd(&st->stamp[lane], g_stampAcc[threadIdx.x >> 6][lane]); atomicAdd(&st->stampLanes[lane], g_stampLanes[threadIdx.x >> 6][lane]); }
#endif
    if (lane == 0) {
        metaOut->cnt[ws.w] = outEnds.nF + outEnds.nB; metaOut->nf[ws.w] = outEnds.nF;
        metaShadow->cnt[ws.w] = shadowCount[threadIdx.x >> 6][0] + shadowCount[threadIdx.x >> 6][1]; metaShadow->nf[ws.w] = shadowCount[threadIdx.x >> 6][0];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        metaOut->chunk = ws.chunk; metaOut->nSeg = ws.W; metaShadow->chunk = ws.chunk; metaShadow->nSeg = ws.W;
        metaOut->certFront = metaShadow->certFront = A.S.gatesExact ? 1u : 0u;          // what ray_gate_class filed at the front is proven gate-free
    }
    if (ST & 1) flush_stats(st, c);
    if ((ST & 2) && c.envelope) atomicAdd(&st->rngOverflow, 1ull);
}

// visible() for every queued next-event segment (main.cpp:64-80, 143-144): the sample's term of this bounce is the
// segment's radiance if it is unobstructed, black otherwise.
struct ShadowArgs { DScene S; ShadowQueue SQ; QMetaRO meta; TermBuf TB; DStats* st; };
template <int ST, int ARITH = FRAY_ARITH>
static __global__ __launch_bounds__(256, anyhit_waves(ST)) void k_pt_shadow(ShadowArgs A)
{
    Cnt c = zero_cnt();
    const QMetaRO meta = A.meta;
    DStats* const st = A.st;
    const FRAY_RO uint32_t* off = meta.p->off;
    const FRAY_RO uint32_t* nfIn = meta.p->nf;
    const uint32_t nSeg = meta.p->nSeg, chunkIn = meta.p->chunk;
    const bool certFront = meta.p->certFront != 0;
    const WaveShare ws = wave_share(meta.p->n);
    const uint32_t lane = threadIdx.x & 63u;
    if (ws.begin >= ws.end) return;
    uint32_t seg = seg_first(off, nSeg, ws.begin);
#ifdef FRAY_STAMPS
    stamp_begin();
#endif
    constexpr bool SORT = FRAY_SORT && sort_variant(ST);
    const uint32_t sortN = (uint32_t)FRAY_SORT_N;
    for (uint32_t b0 = ws.begin, b1 = 0; b0 < ws.end; b0 = b1) {            // (a kernel that does not sort takes its whole share in one go)
    b1 = (SORT && ws.end - b0 > sortN) ? b0 + sortN : ws.end;
    uint32_t* sorted = nullptr;
    if constexpr (SORT) {
        __shared__ uint32_t sortBuf[4][FRAY_SORT_N];
        __shared__ uint32_t sortCnt[4][16];
        sorted = sortBuf[threadIdx.x >> 6];
        const unsigned char* const cls = KARG(ShadowArgs, kernel_args<ShadowArgs>(), SQ).cls;
        sort_share(off, nfIn, nSeg, chunkIn, b0, b1, seg, sorted, sortCnt[threadIdx.x >> 6], [&](uint32_t i) { return (uint32_t)cls[i] & 15u; });
    }
    for (uint32_t base = b0; base < b1; base += 64u) {
        const FRAY_RO ShadowArgs* AP = kernel_args<ShadowArgs>();
        const DScene& S = KARG(ShadowArgs, AP, S);
        const ShadowQueue& SQ = KARG(ShadowArgs, AP, SQ);
        const TermBuf& TB = KARG(ShadowArgs, AP, TB);
        const uint32_t di = base + lane;
        const bool live = di < b1;
        uint32_t i;
        if constexpr (SORT) i = live ? sorted[di - b0] : 0u;
        else i = seg_map(off, nfIn, nSeg, chunkIn, base, di, live, seg);
        const bool gateFree = certFront && (i & FRAY_FRONT_BIT);
        i &= ~FRAY_FRONT_BIT;
        if (live) {
            const V3 a = v3(SQ.ax[i], SQ.ay[i], SQ.az[i]), b = v3(SQ.bx[i], SQ.by[i], SQ.bz[i]);
            STAMP(0);
            const bool vis = certFront ? visible<ST, true>(S, a, b, c, gateFree) : visible<ST, false>(S, a, b, c);       // (wave-uniform choice)
            const uint32_t sl = SQ.slot[i];
#ifdef FRAY_QCHECK
            // diagnostic build: an entry whose slot is not a slot of this batch was never written by the bounce kernel (or was consumed before): count it
            // (the frame then fails with E_UNSUPPORTED), do not store; consumed entries are poisoned
            if (sl >= TB.nPaths) { atomicAdd(&st->rngOverflow, 1ull); continue; }
            SQ.slot[i] = 0xffffffffu;
#endif
            term_store(TB, sl, vis ? c3(SQ.cr[i], SQ.cg[i], SQ.cb[i]) : c3(0, 0, 0));
        }
        STAMP(13);
    }
    }
#ifdef FRAY_STAMPS
    if (lane < 24) { atomicAdd(&st->stamp[lane], g_stampAcc[threadIdx.x >> 6][lane]); atomicAdd(&st->stampLanes[lane], g_stampLanes[threadIdx.x >> 6][lane]); }
#endif
    if (ST & 1) flush_stats(st, c);
    if ((ST & 2) && c.envelope) atomicAdd(&st->rngOverflow, 1ull);
}

// The same two kernels compiled with fused multiply-adds (render_contract.hip, one translation unit per flag word like render_variant.hip)
namespace frayhip_detail {
template <int ST> void launch_bounce_contracted(int grid, hipStream_t stream, const BounceArgs& A);
template <int ST> void launch_shadow_contracted(int grid, hipStream_t stream, const ShadowArgs& A);
}

// A camera sample's radiance from its terms, innermost first (TermBuf): result = term[n-1]; result = term[k] + result for k = n-2 .. 0.
static __global__ __launch_bounds__(256) void k_pt_fold(TermBuf TB, uint32_t total, float* __restrict__ sampleRad)
{
    for (uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x; slot < total; slot += gridDim.x * blockDim.x) {
        C3 result = c3(0, 0, 0);
        for (int k = (int)TB.n[slot] - 1; k >= 0; k--) {
            const size_t q = (size_t)k * 3 * TB.nPaths + slot;
            result = c3(TB.t[q], TB.t[q + TB.nPaths], TB.t[q + 2 * (size_t)TB.nPaths]) + result;
        }
        const size_t o = (size_t)slot * 3;
        sampleRad[o] = result.r; sampleRad[o + 1] = result.g; sampleRad[o + 2] = result.b;
    }
}

// The mono frame's fold and resolve in one pass: per pixel, every sample's terms are added innermost first (k_pt_fold's order) and the samples
// in sample order (k_pt_resolve's), without the per-sample radiance making a round trip through memory.
static __global__ __launch_bounds__(256) void k_pt_resolve_terms(DFrame F, int nItems, int s0, int chunk, TermBuf TB, float* __restrict__ sum, float* __restrict__ rgb)
{
    for (int item = blockIdx.x * blockDim.x + threadIdx.x; item < nItems; item += gridDim.x * blockDim.x) {
        int x, y;
        if (!item_pixel(F, item, x, y)) continue;
        const size_t si = (size_t)item * 3;
        C3 a = s0 == 0 ? c3(0, 0, 0) : c3(sum[si], sum[si + 1], sum[si + 2]);
        for (int s = 0; s < chunk; s++) {
            const uint32_t slot = (uint32_t)s * (uint32_t)nItems + (uint32_t)item;
            const int n = (int)TB.n[slot];
            C3 result = c3(0, 0, 0);
            if (n <= 8) {
                // up to eight terms (maxTraceDepth 6, the usual case): all their loads issued before the first addition -- with a loop whose length is
                // the loaded count every term waited for the one before it (1.6 ms per 66 M samples at 1.8 TB/s, the kernel idle on latency)
                float tr[8], tg[8], tb[8];
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    tr[k] = tg[k] = tb[k] = 0.0f;
                    if (k < n) {
                        const size_t q = (size_t)k * 3 * TB.nPaths + slot;
                        tr[k] = TB.t[q]; tg[k] = TB.t[q + TB.nPaths]; tb[k] = TB.t[q + 2 * (size_t)TB.nPaths];
                    }
                }
#pragma unroll
                for (int k = 7; k >= 0; k--)
                    if (k < n) result = c3(tr[k], tg[k], tb[k]) + result;
            } else {
                for (int k = n - 1; k >= 0; k--) {
                    const size_t q = (size_t)k * 3 * TB.nPaths + slot;
                    result = c3(TB.t[q], TB.t[q + TB.nPaths], TB.t[q + 2 * (size_t)TB.nPaths]) + result;
                }
            }
            a = a + result;
        }
        if (s0 + chunk >= F.spp) {
            a = a / (float)F.spp;
            const size_t p = ((size_t)y * F.W + x) * 3;
            rgb[p] = a.r; rgb[p + 1] = a.g; rgb[p + 2] = a.b;
        } else {
            sum[si] = a.r; sum[si + 1] = a.g; sum[si + 2] = a.b;
        }
    }
}

// vfb[y][x] = (sum over samples in order) / spp  (main.cpp:348-360).  `sum` carries the running
// FP32 sum across batches so the addition order is the reference's.
static __global__ __launch_bounds__(256) void k_pt_resolve(DFrame F, DCamera C, float saturation, int nItems, int s0, int chunk,
                                                    const float* __restrict__ sampleRad, const float* __restrict__ sampleRadR,
                                                    float* __restrict__ sum, float* __restrict__ rgb)
{
    for (int item = blockIdx.x * blockDim.x + threadIdx.x; item < nItems; item += gridDim.x * blockDim.x) {
        int x, y;
        if (!item_pixel(F, item, x, y)) continue;
        size_t si = (size_t)item * 3;
        C3 a = s0 == 0 ? c3(0, 0, 0) : c3(sum[si], sum[si + 1], sum[si + 2]);
        for (int s = 0; s < chunk; s++) {
            size_t q = ((size_t)s * nItems + item) * 3;
            C3 cl = c3(sampleRad[q], sampleRad[q + 1], sampleRad[q + 2]);
            if (sampleRadR) {                                 // anaglyph blend, main.cpp:306-317
                C3 cr = c3(sampleRadR[q], sampleRadR[q + 1], sampleRadR[q + 2]);
                if (saturation != 1) {                        // Color::adjustSaturation, color.h:127-133
                    float ml = (cl.r + cl.g + cl.b) / 3.0f, mr = (cr.r + cr.g + cr.b) / 3.0f;
                    cl = c3(ml + (cl.r - ml) * saturation, ml + (cl.g - ml) * saturation, ml + (cl.b - ml) * saturation);
                    cr = c3(mr + (cr.r - mr) * saturation, mr + (cr.g - mr) * saturation, mr + (cr.b - mr) * saturation);
                }
                cl = cl * ldc(C.leftMask) + cr * ldc(C.rightMask);
            }
            a = a + cl;
        }
        if (s0 + chunk >= F.spp) {
            a = a / (float)F.spp;
            size_t p = ((size_t)y * F.W + x) * 3;
            rgb[p] = a.r; rgb[p + 1] = a.g; rgb[p + 2] = a.b;
        } else {
            sum[si] = a.r; sum[si + 1] = a.g; sum[si + 2] = a.b;
        }
    }
}
