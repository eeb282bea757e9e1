// render() of the reference (src/main.cpp:373-405) as kernel launches: frame set-up, the Whitted /
// primary-ray persistent kernels, the path-tracing batches on several streams, timing and counters.
// A template over the kernel flag word ST (bit 0: work counters, bit 1: Cube / CSG geometry);
// render_variant.hip instantiates one ST per translation unit.
#pragma once
#include <algorithm>
#include <chrono>
#include <cstring>
#include <cstdio>

#include "render_state.hpp"
#include "kernels.hpp"

namespace frayhip_detail {

namespace {
// Carves a path queue / a shadow queue of n entries out of the workspace.
unsigned char* carve_queue(unsigned char* p, size_t n, PathQueue& Q)
{
    auto take = [&](size_t bytes) { unsigned char* r = p; p += (bytes + 255) / 256 * 256; return r; };
    Q.rec = (PathRec*)take(n * sizeof(PathRec));
    Q.cls = take(n);
    return p;
}
size_t queue_bytes(size_t n)
{
    auto r = [](size_t b) { return (b + 255) / 256 * 256; };
    return r(n * sizeof(PathRec)) + r(n);
}
unsigned char* carve_shadow(unsigned char* p, size_t n, ShadowQueue& Q)
{
    auto take = [&](size_t bytes) { unsigned char* r = p; p += (bytes + 255) / 256 * 256; return r; };
    Q.ax = (double*)take(n * 8); Q.ay = (double*)take(n * 8); Q.az = (double*)take(n * 8);
    Q.bx = (double*)take(n * 8); Q.by = (double*)take(n * 8); Q.bz = (double*)take(n * 8);
    Q.cr = (float*)take(n * 4); Q.cg = (float*)take(n * 4); Q.cb = (float*)take(n * 4);
    Q.slot = (uint32_t*)take(n * 4);
    Q.cls = take(n);
    return p;
}
size_t shadow_bytes(size_t n)
{
    auto r = [](size_t b) { return (b + 255) / 256 * 256; };
    return 6 * r(n * 8) + 4 * r(n * 4) + r(n);
}
}  // namespace

template <int ST>
int render_impl(frayhip_scene* sc, const frayhip_frame* f, float* d_rgb, int32_t* d_id, double* d_dist, hipStream_t stream, frayhip_stats* st)
{
    const auto t0 = std::chrono::steady_clock::now();
    const frayhip_settings& set = sc->settings;
    const int W = set.frameWidth, H = set.frameHeight;
    DFrame F{};
    F.W = W; F.H = H;
    F.BW = (W - 1) / 48 + 1; F.BH = (H - 1) / 48 + 1;
    F.bucketStride = f->bucket_stride > 0 ? f->bucket_stride : 1;
    F.bucketFirst = f->bucket_first;
    F.nBuckets = frayhip_bucket_count(W, H, F.bucketFirst, F.bucketStride);
    if (F.nBuckets < 0) { set_error("frayhip_render: bad bucket_first / bucket_stride"); return FRAYHIP_E_ARG; }
    if ((long long)F.nBuckets * 2304 > (1ll << 30)) { set_error("frayhip_render: more than 2^30 pixels in one call (shard the frame with bucket_first / bucket_stride)"); return FRAYHIP_E_UNSUPPORTED; }
    int spp = set.wantAA ? 5 : 1;                                   // main.cpp:395-400
    if (sc->camera.dof) spp = std::max(spp, sc->camera.numDOFSamples);
    if (set.gi) spp = std::max(spp, set.numPaths);
    F.spp = spp;
    F.seed = f->seed;
    F.jitter = (sc->camera.dof || set.gi) ? 1 : 0;
    const int nItems = F.nBuckets * 2304;
    DScene S = sc->S;
    S.ambient[0] = set.ambientLight[0]; S.ambient[1] = set.ambientLight[1]; S.ambient[2] = set.ambientLight[2];
    S.maxTraceDepth = set.maxTraceDepth;
    S.gi = set.gi;
    S.saturation = set.saturation;
    DCamera C = camera_begin_frame(sc->camera, W, H);

    // An early (error) return below must not leave work in flight on the side lanes' streams, which the caller cannot see: whatever
    // was enqueued is drained before the call returns.
    struct Drain { bool armed = true; ~Drain() { if (armed) (void)hipDeviceSynchronize(); } } drain;
    HIP_TRY(hipMemsetAsync(sc->d_stats, 0, kStatsBytes, stream));
    DCursors* cursors = (DCursors*)((unsigned char*)sc->d_stats + kCursorOffset);
    HIP_TRY(hipEventRecord(sc->evA, stream));
    size_t nTraceEvents = 0, nShadowEvents = 0;
    long long nContracted = 0;          // launches of this frame that ran a kernel of render_contract.hip (option "fp_contract")
    const unsigned long long* fanTotals = nullptr;          // device: the speculative fans' counters summed over the frame's batches

    if (f->mode == FRAYHIP_MODE_PRIMARY_ID) {
        if (nItems > 0) {
            hipEvent_t a = pool_event(sc->evPool, 0), b = pool_event(sc->evPool, 1);
            HIP_TRY(hipEventRecord(a, stream));
            hipLaunchKernelGGL(k_primary<ST>, dim3(persistent_grid(nItems, primary_waves(ST))), dim3(256), 0, stream, PrimaryArgs{S, C, F, nItems, d_id, d_dist, sc->d_stats, cursors});
            HIP_TRY(hipEventRecord(b, stream));
            nTraceEvents = 2;
        }
    } else if (f->mode == FRAYHIP_MODE_RENDER) {
        if (!d_rgb) { set_error("frayhip_render: MODE_RENDER needs an rgb buffer"); return FRAYHIP_E_ARG; }
        if (set.maxTraceDepth < 0) {
            if (nItems > 0) hipLaunchKernelGGL(k_black, dim3(grid_for(nItems)), dim3(256), 0, stream, F, nItems, sc->camera.stereoSeparation > 0 ? 2 : 1, d_rgb, sc->d_stats);
        } else if (!set.gi) {
            if (!F.jitter && spp > 5) { set_error("frayhip_render: bad sample count"); return FRAYHIP_E_ARG; }
            sc->lastWhittedPath = 0;
            if (nItems > 0 && sc->whittedNeedsRecursion) {
                // workspace: per-thread mt19937 state columns for samples that draw more than 227 words, the pixels' running sums, then per
                // (pixel, sample) of a batch the sample's colour and x[397] of its seed.  A work item of k_whitted is one camera sample, so the
                // grid is sized by the samples, and a frame whose samples do not fit the budget (or 2^31 items) is rendered in batches of `chunk` samples per pixel.
                auto r256 = [](size_t b) { return (b + 255) / 256 * 256; };
                const int grid = persistent_grid((size_t)nItems * spp, whitted_waves(ST));                 // one pass
                const int gridAC = persistent_grid((size_t)nItems * spp, whitted_waves(ST, 1));            // passes A and C of the speculative fans
                const size_t colBytes = r256((size_t)std::max(grid, gridAC) * 256 * 624 * sizeof(uint32_t));
                // Speculative glossy fans (dev_whitted.hpp): three passes per batch, with room for one filed entry per camera sample and specFanMax children each.
                // Not in the counting variants (their counters are the reference's call counts), not in stereo frames (the right eye continues the left eye's generator).
                int fan = (!(ST & 1) && sc->speculateFans && !(sc->camera.stereoSeparation > 0)) ? sc->specFanMax : 0;
                int chunk = 0;
                for (;;) {          // planned again with half the budget when the allocation fails (ensure_work_or_shrink)
                    const size_t wb = work_budget(sc);
                    const size_t budget = wb > colBytes + (64u << 20) ? wb - colBytes : (64u << 20);
                    // a fan so large that one sample per pixel of it does not fit the budget (or 2^31 children) is not traced ahead
                    if (fan > 0 && ((size_t)nItems * (48 + (size_t)fan * 42) > budget || (size_t)nItems * (size_t)fan >= ((size_t)1 << 31))) fan = 0;
                    const size_t perSample = 16 + (fan > 0 ? 32 + (size_t)fan * 42 : 0);
                    chunk = f->spp_chunk > 0 ? f->spp_chunk : (int)std::max<size_t>(1, budget / ((size_t)nItems * perSample));
                    if (chunk > spp) chunk = spp;
                    while (chunk > 1 && (size_t)nItems * chunk * (size_t)std::max(fan, 1) >= ((size_t)1 << 31)) chunk /= 2;
                    const size_t slots = (size_t)nItems * chunk, kids = slots * (size_t)fan;
                    const bool canRetry = f->spp_chunk <= 0 && chunk > 1;
                    const int rc = ensure_work_or_shrink(sc, colBytes + r256((size_t)nItems * 12) + r256(slots * 12) + r256(slots * 4) +
                                                             (fan > 0 ? 256 + 2 * r256(slots * 4) + 3 * r256(slots * 8) + r256(kids * 4) + 3 * r256(kids * 8) + 3 * r256(kids * 4) + 2 * r256(kids) : 0), canRetry);
                    if (rc == FRAYHIP_RETRY_SMALLER) continue;
                    if (rc) return rc;
                    break;
                }
                unsigned char* p = (unsigned char*)sc->d_work;
                auto take = [&](size_t b) { unsigned char* r = p; p += r256(b); return r; };
                uint32_t* mtWork = (uint32_t*)take(colBytes);
                float* sum = (float*)take((size_t)nItems * 12);
                float* rad = (float*)take((size_t)nItems * chunk * 12);
                uint32_t* x397 = (uint32_t*)take((size_t)nItems * chunk * 4);
                SpecBuf SPB{};
                if (fan > 0) {
                    const size_t slots = (size_t)nItems * chunk, kids = slots * (size_t)fan;
                    SPB.counters = (int*)take(256);
                    SPB.minCount = 8;
                    SPB.eSlot = (int*)take(slots * 4); SPB.eChildBase = (int*)take(slots * 4);
                    for (int q = 0; q < 3; q++) SPB.eo[q] = (double*)take(slots * 8);
                    SPB.cEntry = (int*)take(kids * 4);
                    for (int q = 0; q < 3; q++) SPB.cd[q] = (double*)take(kids * 8);
                    for (int q = 0; q < 3; q++) SPB.cc[q] = (float*)take(kids * 4);
                    SPB.cok = take(kids);
                    SPB.cdraws = take(kids);
                }
                for (int s0 = 0; s0 < spp; s0 += chunk) {
                    const int cn = std::min(chunk, spp - s0);
                    if (s0 > 0) HIP_TRY(hipMemsetAsync(cursors, 0, 3 * sizeof(DCursors), stream));  // the tile cursors of the previous batch
                    if (fan > 0) HIP_TRY(hipMemsetAsync(SPB.counters, 0, s0 == 0 ? 256 : 32, stream));
                    hipLaunchKernelGGL(k_seed, dim3(seed_grid(((size_t)nItems * cn + FRAY_SEED_CHAINS - 1) / FRAY_SEED_CHAINS)), dim3(256), 0, stream, F, nItems, s0, cn, x397);
                    hipEvent_t a = pool_event(sc->evPool, nTraceEvents), b = pool_event(sc->evPool, nTraceEvents + 1);
                    if (!a || !b) return FRAYHIP_E_NOMEM;
                    HIP_TRY(hipEventRecord(a, stream));
                    const WhittedArgs WA{S, C, F, nItems, s0, cn, d_rgb, rad, mtWork, x397, sc->d_stats, cursors, SPB};
                    if constexpr (!(ST & 1)) {
                        if (fan > 0) {
                            WhittedArgs WB = WA, WC = WA;
                            WB.cur = cursors + 1; WC.cur = cursors + 2;
                            // three launches, three event pairs: frayhip_stats.trace_launches counts LAUNCHES (round 4 timed the three as one and reported
                            // launches_per_step 1, which no per-kernel profile could agree with)
                            hipEvent_t b1 = pool_event(sc->evPool, nTraceEvents + 2), b2 = pool_event(sc->evPool, nTraceEvents + 3);
                            hipEvent_t c1 = pool_event(sc->evPool, nTraceEvents + 4), c2 = pool_event(sc->evPool, nTraceEvents + 5);
                            if (!b1 || !b2 || !c1 || !c2) return FRAYHIP_E_NOMEM;
                            hipLaunchKernelGGL((k_whitted<ST, 1>), dim3(gridAC), dim3(256), 0, stream, WA);   // samples; fans are filed
                            HIP_TRY(hipEventRecord(b, stream));
                            HIP_TRY(hipEventRecord(b1, stream));
                            hipLaunchKernelGGL((k_whitted<ST, 2>), dim3(persistent_grid((size_t)nItems * cn * (size_t)fan, whitted_waves(ST, 2))), dim3(256), 0, stream, WB);     // the fans' children
                            HIP_TRY(hipEventRecord(b2, stream));
                            HIP_TRY(hipEventRecord(c1, stream));
                            hipLaunchKernelGGL((k_whitted<ST, 3>), dim3(gridAC), dim3(256), 0, stream, WC);   // the filed samples, children looked up
                            nTraceEvents += 4;
                            b = c2;
                        } else {
                            hipLaunchKernelGGL((k_whitted<ST, 0>), dim3(grid), dim3(256), 0, stream, WA);
                        }
                    } else {
                        hipLaunchKernelGGL((k_whitted<ST, 0>), dim3(grid), dim3(256), 0, stream, WA);
                    }
                    HIP_TRY(hipEventRecord(b, stream));
                    nTraceEvents += 2;
                    if (spp > 1) hipLaunchKernelGGL(k_pt_resolve, dim3(grid_for(nItems)), dim3(256), 0, stream, F, C, set.saturation, nItems, s0, cn, rad, (const float*)nullptr, sum, d_rgb);
                    if (fan > 0) hipLaunchKernelGGL(k_add4, dim3(1), dim3(64), 0, stream, SPB.counters, (unsigned long long*)(SPB.counters + 8));       // the frame's totals over its batches
                }
                fanTotals = fan > 0 ? (const unsigned long long*)(SPB.counters + 8) : nullptr;
            } else if (nItems > 0) {
                // Wavefront Whitted (no recursive shader in the scene): batches of `chunk` samples per pixel through
                // k_wh_shade -> k_wh_visible -> k_wh_gather, then the ordered per-pixel sum (k_pt_resolve).
                const int T = sc->lightSampleCount;
                const bool stereo = sc->camera.stereoSeparation > 0;
                const size_t eyes = stereo ? 2 : 1;
                auto r256 = [](size_t b) { return (b + 255) / 256 * 256; };
                const int grid = persistent_grid((size_t)nItems * spp, waves_for(ST, FRAY_WH_SHADE_WAVES));
                const size_t colBytes = r256((size_t)grid * 256 * 624 * sizeof(uint32_t));
                // per (pixel, sample): base 12 + a 24 + hit 1 + radiance 12 per eye, 37 per light sample and eye, 4 for the seed
                const size_t perSlot = eyes * (49 + (size_t)T * 37) + 4;
                int chunk = 0;
                size_t slots = 0, N = 0, NT = 0;
                for (;;) {          // planned again with half the budget when the allocation fails (ensure_work_or_shrink)
                    const size_t wb = work_budget(sc);
                    const size_t budget = wb > colBytes + (64u << 20) ? wb - colBytes : (64u << 20);
                    chunk = f->spp_chunk > 0 ? f->spp_chunk : (int)std::max<size_t>(1, budget / ((size_t)nItems * perSlot));
                    if (chunk > spp) chunk = spp;
                    while (chunk > 1 && (size_t)nItems * chunk * eyes * (size_t)std::max(T, 1) > ((size_t)1 << 31)) chunk /= 2;
                    slots = (size_t)nItems * chunk; N = slots * eyes; NT = N * (size_t)T;
                    const size_t bytes = colBytes + r256((size_t)nItems * 12) + r256(N * 12) + 3 * r256(N * 8) + r256(N) + 3 * r256(NT * 8) + 3 * r256(NT * 4) + r256(NT) +
                                         2 * r256(slots * 12) + r256(slots * 4) + 4096;
                    const int rc = ensure_work_or_shrink(sc, bytes, f->spp_chunk <= 0 && chunk > 1);
                    if (rc == FRAYHIP_RETRY_SMALLER) continue;
                    if (rc) return rc;
                    break;
                }
                unsigned char* p = (unsigned char*)sc->d_work;
                auto take = [&](size_t b) { unsigned char* r = p; p += r256(b); return r; };
                uint32_t* mtWork = (uint32_t*)take(colBytes);
                float* sum = (float*)take((size_t)nItems * 12);
                WhittedQueue Q;
                Q.base = (float*)take(N * 12);
                Q.ax = (double*)take(N * 8); Q.ay = (double*)take(N * 8); Q.az = (double*)take(N * 8);
                Q.hit = take(N);
                Q.bx = (double*)take(NT * 8); Q.by = (double*)take(NT * 8); Q.bz = (double*)take(NT * 8);
                Q.rr = (float*)take(NT * 4); Q.rg = (float*)take(NT * 4); Q.rb = (float*)take(NT * 4);
                Q.vis = take(NT);
                float* radL = (float*)take(slots * 12);
                float* radR = (float*)take(slots * 12);
                uint32_t* x397 = (uint32_t*)take(slots * 4);
                // Few light samples per hit (zaphod, forest: one point light): the fused form of k_wh_shade asks visible() in place -- one launch instead of
                // shade + visible + gather; no k_seed when no generator can be asked for a word; no resolve when the kernel writes the pixel itself.
                // (not beside KD meshes: there the fused kernel spills 165 registers and the lean any-hit kernel wins -- forest DOF 256 162.4 against 168.2 ms fused)
                const bool fusedShade = T <= sc->fusedWhittedMax && !kd_variant(ST);
                sc->lastWhittedPath = fusedShade ? 2 : 1;
                const bool draws = F.jitter || sc->camera.dof || sc->lightDraws;
                for (int s0 = 0; s0 < spp; s0 += chunk) {
                    const int cn = std::min(chunk, spp - s0);
                    const size_t bs = (size_t)nItems * cn, bN = bs * eyes;      // this batch's slots: arrays are used with stride bN
                    if (!fusedShade || draws)
                        hipLaunchKernelGGL(k_seed, dim3(seed_grid((bs + FRAY_SEED_CHAINS - 1) / FRAY_SEED_CHAINS)), dim3(256), 0, stream, F, nItems, s0, cn, x397);
                    hipEvent_t ea = pool_event(sc->evPool, nTraceEvents), eb = pool_event(sc->evPool, nTraceEvents + 1);
                    hipEvent_t ec = pool_event(sc->evPoolShadow, nShadowEvents), ed = pool_event(sc->evPoolShadow, nShadowEvents + 1);
                    if (!ea || !eb || !ec || !ed) return FRAYHIP_E_NOMEM;
                    HIP_TRY(hipEventRecord(ea, stream));
                    if (s0 > 0) HIP_TRY(hipMemsetAsync(cursors, 0, 2 * sizeof(DCursors), stream));        // the previous batch's tile cursors (one set per kernel)
                    // tiles are claimed when a wave gets at least 16 of them, walked with a fixed stride otherwise (next_tile, kernels.hpp)
                    const bool claimShade = bs / 64 >= (size_t)grid * 4 * 16;
                    if (fusedShade) {
                        const bool direct = spp == 1 && !stereo;
                        hipLaunchKernelGGL((k_wh_shade<ST, true>), dim3(grid), dim3(256), 0, stream,
                                           WhShadeArgs{S, C, F, nItems, s0, cn, Q, mtWork, draws ? x397 : nullptr, sc->d_stats, claimShade ? cursors : nullptr, radL, radR, direct ? d_rgb : nullptr});
                        HIP_TRY(hipEventRecord(eb, stream));
                        nTraceEvents += 2;
                        if (!direct) hipLaunchKernelGGL(k_pt_resolve, dim3(grid_for(nItems)), dim3(256), 0, stream, F, C, set.saturation, nItems, s0, cn, radL, stereo ? radR : nullptr, sum, d_rgb);
                        continue;
                    }
                    const int gridVis = persistent_grid(bN * (size_t)T, anyhit_waves(ST));
                    const bool claimVis = (bN * (size_t)T) / 64 >= (size_t)gridVis * 4 * 16;
                    hipLaunchKernelGGL(k_wh_shade<ST>, dim3(grid), dim3(256), 0, stream, WhShadeArgs{S, C, F, nItems, s0, cn, Q, mtWork, x397, sc->d_stats, claimShade ? cursors : nullptr, nullptr, nullptr, nullptr});
                    HIP_TRY(hipEventRecord(eb, stream));
                    nTraceEvents += 2;
                    HIP_TRY(hipEventRecord(ec, stream));
                    if (T > 0) hipLaunchKernelGGL(k_wh_visible<ST>, dim3(gridVis), dim3(256), 0, stream, WhVisibleArgs{S, Q, bN, T, sc->d_stats + 1, claimVis ? cursors + 1 : nullptr});
                    HIP_TRY(hipEventRecord(ed, stream));
                    nShadowEvents += 2;
                    hipLaunchKernelGGL(k_wh_gather, dim3(grid_for(bN)), dim3(256), 0, stream, S, Q, bN, bs, radL, radR);
                    hipLaunchKernelGGL(k_pt_resolve, dim3(grid_for(nItems)), dim3(256), 0, stream, F, C, set.saturation, nItems, s0, cn, radL, stereo ? radR : nullptr, sum, d_rgb);
                }
            }
        } else if (nItems > 0) {
            if (set.maxTraceDepth > 2000) { set_error("frayhip_render: maxTraceDepth above 2000 is not supported (three launches per level and batch)"); return FRAYHIP_E_UNSUPPORTED; }
            // Random words a camera sample may draw from one generator: lens samples (DOF, both eyes) and ten per Lambert bounce (main.cpp:219-236,
            // lights.cpp:62-63).  Up to 227 the generators are three registers; beyond that every path gets two 624-word columns (MtPath).
            const bool longRng = 8 + 10 * (set.maxTraceDepth + 2) > 227;
            const size_t termBytes = (size_t)(set.maxTraceDepth + 2) * 12 + 2;       // one FP32 RGB term per bounce and sample, and their count
            const size_t perPath = 240 + termBytes + (longRng ? 2 * 624 * sizeof(uint32_t) : 0);
            // Batches of `chunk` samples per pixel; up to FRAY_PT_LANES batches are in flight at once, each on its own
            // stream with its own queues, so one batch's launch gaps, scans and kernel tails are filled by the others'
            // blocks.  Only the resolves are ordered (evResolved): the per-pixel sum runs in sample order.
            int maxLanes = std::max(1, std::min(sc->ptLanes, FRAY_PT_LANES));
            // The Cube / CSG kernel variants keep their hit lists in scratch memory (17 KB per lane at sixteen CsgOp levels): every stream that runs
            // one needs its own scratch arena, and three streams asking for theirs at once aborted inside the runtime (tests/test_fuzz_parity.py,
            // seed 5).  Those scenes run their batches one after the other.
            if (ST & 2) maxLanes = std::max(1, std::min(maxLanes, sc->csgLanes));
            // Round 5: kernels that keep spilled registers in scratch memory must not share the chip with kernels of ANOTHER scratch size launched from other
            // streams.  The counting variants showed it: their first bounce (76 B of scratch per lane) and their later bounces (112 B) on three batch lanes at once
            // rendered wrong pixels in one frame in ten -- right on one lane, right when built without spills (tools/dbg_fuzz_seed.py, profiles/r05_experiments
            // README H) -- while the timed variants of the same scenes, whose first bounce and shadow kernels use no scratch at all, never did.  So: (1) the
            // counting variants, which are instrumentation, run their batches on ONE lane; (2) every lane's stream is primed once per scene and kernel set
            // with an empty launch of each kernel of the set, one stream after the other, so that no queue has to enlarge its scratch arena while another
            // lane's waves are resident.
            if (ST & 1) maxLanes = 1;
            const bool longRngW = 8 + 10 * (set.maxTraceDepth + 2) > 227;
            const unsigned warmKey = 1u << ((longRngW ? 1 : 0) | (sc->fpContract ? 2 : 0) | (sc->camera.stereoSeparation > 0 ? 4 : 0));
            if (maxLanes > 1 && !(sc->warmMask & warmKey) && !getenv("FRAYHIP_NO_PRIME")) {
                HIP_TRY(hipMemsetAsync(sc->d_qmeta, 0, 3 * FRAY_PT_LANES * sizeof(QMeta), stream));
                HIP_TRY(hipStreamSynchronize(stream));
                for (int k = 0; k < maxLanes; k++) {
                    hipStream_t ws = k == 0 ? stream : sc->laneStream[k];
                    QMeta* m = sc->d_qmeta + 3 * k;
                    const QMetaRO mIn{(const FRAY_RO QMeta*)m};
                    BounceArgs BA{};
                    BA.S = S; BA.metaIn = mIn; BA.metaOut = m + 1; BA.metaShadow = m + 2; BA.st = sc->d_stats;
                    ShadowArgs SA{};
                    SA.S = S; SA.meta = mIn; SA.st = sc->d_stats + 1;
                    if (longRngW) hipLaunchKernelGGL((k_pt_bounce<ST, true>), dim3(1), dim3(256), 0, ws, BA);
                    else {
                        hipLaunchKernelGGL((k_pt_bounce<ST, false>), dim3(1), dim3(256), 0, ws, BA);
                        hipLaunchKernelGGL((k_pt_bounce<ST, false, true>), dim3(1), dim3(256), 0, ws, BA);
                    }
                    hipLaunchKernelGGL(k_pt_shadow<ST>, dim3(1), dim3(256), 0, ws, SA);
                    if constexpr (!(ST & 2)) {
                        if (sc->fpContract) { launch_bounce_contracted<ST>(1, ws, BA); launch_shadow_contracted<ST>(1, ws, SA); }
                    }
                    HIP_TRY(hipStreamSynchronize(ws));
                }
                sc->warmMask |= warmKey;
            }
            // a small frame (an eighth of 1080p x 64 spp, i.e. one rank's share of an 8-rank run) is cut into fewer, larger batches:
            // measured 15.5 ms on three lanes against 15.9 on four; from a quarter of that frame upwards four lanes win
            if (maxLanes > 3 && (size_t)nItems * (size_t)spp < ((size_t)24 << 20)) maxLanes = 3;
            const bool stereo = sc->camera.stereoSeparation > 0;
            int chunk = 0, nBatches = 0, nLanes = 0;
            size_t nPaths = 0, nQueue = 0, laneBytes = 0;
            for (;;) {              // planned again with half the budget when the allocation fails (ensure_work_or_shrink)
                const size_t budget = std::max<size_t>(work_budget(sc) / perPath, 1);       // paths in flight over all lanes
                chunk = f->spp_chunk > 0 ? f->spp_chunk : (int)std::max<size_t>(1, budget / maxLanes / (size_t)nItems);
                if (chunk > spp) chunk = spp;
                if (f->spp_chunk <= 0 && spp >= 2 * maxLanes && chunk * maxLanes > spp) chunk = (spp + maxLanes - 1) / maxLanes;   // enough batches to fill the lanes
                while (chunk > 1 && (size_t)nItems * chunk > ((size_t)1 << 30)) chunk /= 2;   // slots are 32-bit
                nBatches = (spp + chunk - 1) / chunk;
                nLanes = std::min(nBatches, maxLanes);
                nPaths = (size_t)nItems * chunk;
                // per-wave segments round their share up to a multiple of 64: one extra wave-load per wave of the grid
                nQueue = nPaths + (size_t)bounce_grid(nPaths, (ST & 2) != 0 && maxLanes == 1) * 4 * 128;
                laneBytes = 2 * queue_bytes(nQueue) + shadow_bytes(nQueue) + nPaths * 12 + nPaths * 4 + 4096 + nPaths * termBytes + 512 + (longRng ? nPaths * 2 * 624 * sizeof(uint32_t) + 256 : 0) +
                            (stereo ? nPaths * (12 + 6 * 8 + 6 * 4) + 16 * 256 : 0);
                const int rc = ensure_work_or_shrink(sc, (size_t)nLanes * laneBytes + (size_t)nItems * 12 + 4096, f->spp_chunk <= 0 && (chunk > 1 || nLanes > 1));
                if (rc == FRAYHIP_RETRY_SMALLER) { if (chunk == 1 && maxLanes > 1) maxLanes--; continue; }
                if (rc) return rc;
                break;
            }
            struct Lane {
                hipStream_t stream;
                PathQueue Q[2];
                ShadowQueue SQ;
                float *sampleRad, *sampleRadR;
                uint32_t* x397;
                uint32_t* mtCols;
                float* terms;
                unsigned short* termCount;
                StereoBuf SB;
                QMeta* meta;
            } lane[FRAY_PT_LANES];
            unsigned char* p = (unsigned char*)sc->d_work;
            float* sum = (float*)p; p += ((size_t)nItems * 12 + 255) / 256 * 256;
            for (int k = 0; k < nLanes; k++) {
                Lane& L = lane[k];
                L.stream = k == 0 ? stream : sc->laneStream[k];
                L.meta = sc->d_qmeta + 3 * k;
                p = carve_queue(p, nQueue, L.Q[0]);
                p = carve_queue(p, nQueue, L.Q[1]);
                p = carve_shadow(p, nQueue, L.SQ);
                L.sampleRad = (float*)p; p += (nPaths * 12 + 255) / 256 * 256;
                L.x397 = (uint32_t*)p; p += (nPaths * 4 + 255) / 256 * 256;
                L.terms = (float*)p; p += (nPaths * (size_t)(set.maxTraceDepth + 2) * 12 + 255) / 256 * 256;
                L.termCount = (unsigned short*)p; p += (nPaths * 2 + 255) / 256 * 256;
                L.mtCols = nullptr;
                if (longRng) { L.mtCols = (uint32_t*)p; p += (nPaths * 2 * 624 * sizeof(uint32_t) + 255) / 256 * 256; }
                L.SB = StereoBuf{};
                L.sampleRadR = nullptr;
                if (stereo) {
                    L.sampleRadR = (float*)p; p += (nPaths * 12 + 255) / 256 * 256;
                    for (int q = 0; q < 6; q++) { L.SB.r[q] = (double*)p; p += (nPaths * 8 + 255) / 256 * 256; }
                    for (int q = 0; q < 6; q++) { L.SB.g[q] = (uint32_t*)p; p += (nPaths * 4 + 255) / 256 * 256; }
                }
            }
            const StereoBuf SBnone{};
            const int nBounce = set.maxTraceDepth + 2;
            HIP_TRY(hipEventRecord(sc->evLaneStart, stream));               // the side lanes start after whatever precedes this frame on the caller's stream
            for (int k = 1; k < nLanes; k++) HIP_TRY(hipStreamWaitEvent(sc->laneStream[k], sc->evLaneStart, 0));
            int batch = 0;
            for (int s0 = 0; s0 < spp; s0 += chunk, batch++) {
                const int cn = std::min(chunk, spp - s0);
                Lane& L = lane[batch % nLanes];
                hipStream_t ls = L.stream;
                hipLaunchKernelGGL(k_seed, dim3(seed_grid(((size_t)nItems * cn + FRAY_SEED_CHAINS - 1) / FRAY_SEED_CHAINS)), dim3(256), 0, ls, F, nItems, s0, cn, L.x397);
                for (int eye = 0; eye < (stereo ? 2 : 1); eye++) {
                    // A mono frame with register generators makes its camera rays inside the first bounce (k_pt_bounce<.., FIRST>); a stereo frame (the
                    // right eye continues the left eye's streams) and long generators start from a dense queue written by k_pt_init
                    const bool fused = !stereo && !longRng;
                    float* rad = eye == 0 ? L.sampleRad : L.sampleRadR;
                    // left pass of a stereo frame saves generator cursors at path end; mono and the right pass do not
                    const StereoBuf& save = (stereo && eye == 0) ? L.SB : SBnone;
                    if (!fused) {
                        // queue 0 is dense: one segment holding every slot of the batch
                        hipLaunchKernelGGL(k_meta_dense, dim3(1), dim3(64), 0, ls, L.meta, (uint32_t)((size_t)nItems * cn));
                        hipLaunchKernelGGL(k_pt_init<ST>, dim3(grid_for((size_t)nItems * cn)), dim3(256), 0, ls, S, C, F, nItems, s0, cn, L.Q[0],
                                           L.termCount, L.x397, L.SB, eye, sc->d_stats);
                    }
                    for (int b = 0; b < nBounce; b++) {
                        const QMetaRO mIn{(const FRAY_RO QMeta*)(L.meta + (b & 1))}, mSh{(const FRAY_RO QMeta*)(L.meta + 2)};
                        const int grid = bounce_grid((size_t)nItems * cn, (ST & 2) != 0 && nLanes == 1);
                        hipEvent_t ea = pool_event(sc->evPool, nTraceEvents), eb = pool_event(sc->evPool, nTraceEvents + 1);
                        hipEvent_t ec = pool_event(sc->evPoolShadow, nShadowEvents), ed = pool_event(sc->evPoolShadow, nShadowEvents + 1);
                        if (!ea || !eb || !ec || !ed) return FRAYHIP_E_NOMEM;
                        HIP_TRY(hipEventRecord(ea, ls));
                        const LongRng LR{L.mtCols, (uint32_t)nPaths, F, nItems, s0};
                        const TermBuf TB{L.terms, L.termCount, (uint32_t)nPaths, b};
                        const FirstArgs FA{C, F, nItems, s0, (uint32_t)((size_t)nItems * cn), L.x397, L.termCount};
                        const BounceArgs BA{S, L.Q[b & 1], L.Q[(b + 1) & 1], L.SQ, mIn, L.meta + ((b + 1) & 1), L.meta + 2, TB, save, LR, sc->d_stats, FA};
                        // option "fp_contract": bounces after a sample's first closest hit (and every visibility query) are colour, bounded by RMS and not by
                        // bits -- they run the kernels compiled with fused multiply-adds (render_contract.hip; not built for the Cube / CSG variants, which
                        // measured slower with it)
                        bool contractedLaunch = false;
                        if constexpr (!(ST & 2)) {
                            if (sc->fpContract && !longRng && b > 0) { launch_bounce_contracted<ST>(grid, ls, BA); contractedLaunch = true; nContracted++; }
                        }
                        if (contractedLaunch) {}
                        else if (longRng) hipLaunchKernelGGL((k_pt_bounce<ST, true>), dim3(grid), dim3(256), 0, ls, BA);
                        else if (fused && b == 0) hipLaunchKernelGGL((k_pt_bounce<ST, false, true>), dim3(grid), dim3(256), 0, ls, BA);
                        else hipLaunchKernelGGL((k_pt_bounce<ST, false>), dim3(grid), dim3(256), 0, ls, BA);
                        HIP_TRY(hipEventRecord(eb, ls));
                        nTraceEvents += 2;
                        hipLaunchKernelGGL(k_scan, dim3(2), dim3(1024), 0, ls, L.meta + ((b + 1) & 1), L.meta + 2);
                        HIP_TRY(hipEventRecord(ec, ls));
                        bool contractedShadow = false;
                        if constexpr (!(ST & 2)) {
                            if (sc->fpContract) { launch_shadow_contracted<ST>(grid, ls, ShadowArgs{S, L.SQ, mSh, TB, sc->d_stats + 1}); contractedShadow = true; nContracted++; }
                        }
                        if (!contractedShadow) hipLaunchKernelGGL(k_pt_shadow<ST>, dim3(grid), dim3(256), 0, ls, ShadowArgs{S, L.SQ, mSh, TB, sc->d_stats + 1});
                        HIP_TRY(hipEventRecord(ed, ls));
                        nShadowEvents += 2;
                    }
                    // the samples' terms -> their radiance, in the reference's innermost-first order.  A stereo frame's two passes share the
                    // term lists, so each pass folds its own; a mono frame folds inside the resolve below
                    if (stereo) hipLaunchKernelGGL(k_pt_fold, dim3(grid_for((size_t)nItems * cn)), dim3(256), 0, ls, TermBuf{L.terms, L.termCount, (uint32_t)nPaths, 0},
                                                   (uint32_t)((size_t)nItems * cn), rad);
                }
                // the running per-pixel sum takes the batches in sample order
                if (batch > 0 && nLanes > 1) HIP_TRY(hipStreamWaitEvent(ls, sc->evResolved[(batch - 1) % nLanes], 0));
                if (stereo) hipLaunchKernelGGL(k_pt_resolve, dim3(grid_for(nItems)), dim3(256), 0, ls, F, C, set.saturation, nItems, s0, cn, L.sampleRad, L.sampleRadR, sum, d_rgb);
                else hipLaunchKernelGGL(k_pt_resolve_terms, dim3(grid_for(nItems)), dim3(256), 0, ls, F, nItems, s0, cn, TermBuf{L.terms, L.termCount, (uint32_t)nPaths, 0}, sum, d_rgb);
                HIP_TRY(hipEventRecord(sc->evResolved[batch % nLanes], ls));
            }
            // the last resolve follows every earlier one, and each resolve is the last launch of its batch
            if (nLanes > 1) HIP_TRY(hipStreamWaitEvent(stream, sc->evResolved[(batch - 1) % nLanes], 0));
        }
    } else {
        set_error("frayhip_render: unknown mode");
        return FRAYHIP_E_ARG;
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(sc->evB, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    drain.armed = false;                    // every lane was joined into `stream` above
    DStats dsv[2];
    HIP_TRY(hipMemcpy(dsv, sc->d_stats, sizeof dsv, hipMemcpyDeviceToHost));
    {
        unsigned long long ft[4] = {0, 0, 0, 0};
        if (fanTotals) HIP_TRY(hipMemcpy(ft, fanTotals, sizeof ft, hipMemcpyDeviceToHost));
        for (int q = 0; q < 4; q++) sc->lastFans[q] = (long long)ft[q];
        sc->lastContracted = nContracted;
    }
#ifdef FRAY_LEAFSTAT
    {
        unsigned long long ls[4] = {0, 0, 0, 0}, zero[4] = {0, 0, 0, 0};
        (void)hipMemcpyFromSymbol(ls, HIP_SYMBOL(g_leafStat), sizeof ls);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_leafStat), zero, sizeof zero);
        if (ls[0]) fprintf(stderr, "[leafstat] wave leaf phases %llu, all active lanes in ONE leaf %.1f %%, active lanes per phase %.1f, distinct leaves per phase %.2f\n",
                           ls[0], 100.0 * (double)ls[1] / (double)ls[0], (double)ls[2] / (double)ls[0], (double)ls[3] / (double)ls[0]);
    }
#endif
#ifdef FRAY_TILESTAT
    if (const char* out = getenv("FRAY_TILESTAT_OUT")) {
        std::vector<unsigned long long> ts(65536 * 8);
        (void)hipMemcpyFromSymbol(ts.data(), HIP_SYMBOL(g_tileStat), ts.size() * sizeof(unsigned long long));
        if (FILE* fp = fopen(out, "wb")) { fwrite(ts.data(), sizeof(unsigned long long), ts.size(), fp); fclose(fp); }
    }
#endif
#ifdef FRAY_STAMPS
    {
        static const char* names[24] = {"queue lookup + ray load / camera ray", "local ray (transform)", "root box test", "tree-less triangle loop", "KD walk: child tests", "other geometry (plane / sphere / KD leaf accept)",
                                        "node result + world distance", "lights", "load rest of path", "KD walk: climbs", "shading (finalize .. spawn / light loops)", "KD leaves: FP64 tests of the candidates",
                                        "k_whitted: cheap steps (returns, loop heads, pushes; pixel / sample set-up)", "queue stores / loop overhead", "k_whitted: rest of the trace step (attributes, bump)", "KD leaves: FP32 filter",
                                        "CsgOp: bounding-box certificate", "CsgOp over two plain operands (no machine)", "CsgOp machine: plain operand asked", "CsgOp machine: answers delivered (sort, in / out walk)", "CsgOp machine: pushes, mesh operand's remainder", "", "", ""};
        for (int q = 0; q < 2; q++) {
            double tot = 0;
            for (int k = 0; k < 24; k++) tot += (double)dsv[q].stamp[k];
            fprintf(stderr, "[stamps] %s: total %.4g wave-cycles\n", q == 0 ? "closest-hit kernel (k_pt_bounce / k_primary / k_wh_shade)" : "any-hit kernel (k_pt_shadow / k_wh_visible)", tot);
            for (int k = 0; k < 24; k++) if (dsv[q].stamp[k]) fprintf(stderr, "[stamps]   %2d %-36s %6.2f %%   lanes %.3f   (%llu)\n", k, names[k], 100.0 * (double)dsv[q].stamp[k] / tot, (double)dsv[q].stampLanes[k] / (64.0 * (double)dsv[q].stamp[k]), (unsigned long long)dsv[q].stamp[k]);
        }
    }
#endif
    if (dsv[0].rngOverflow || dsv[1].rngOverflow) {
#ifdef FRAY_QCHECK
        fprintf(stderr, "[qcheck] bounce kernel: %llu stale path entries, %llu other events; shadow kernel: %llu stale segments\n", (unsigned long long)(dsv[0].rngOverflow >> 32),
                (unsigned long long)(dsv[0].rngOverflow & 0xffffffffull), (unsigned long long)dsv[1].rngOverflow);
#endif
        set_error("frayhip_render: a camera sample left the supported envelope (Whitted: shade() nesting deeper than 40, or a camera sample whose pixel jitter stream passed 227 words)");
        return FRAYHIP_E_UNSUPPORTED;
    }
    if (st) {
        // SURVEY 8(d) byte model, evaluated from the counters (zero unless FRAYHIP_FRAME_STATS)
        auto model = [](const DStats& d) {
            return 88.0 * (double)d.closest + 73.0 * (double)d.shadow + 168.0 * (double)d.node + 16.0 * (double)d.kdInner + 4.0 * (double)d.leafRefs +
                   120.0 * (double)d.tri + 32.0 * (double)d.prim + 144.0 * (double)d.smooth + 12.0 * (double)d.tex;
        };
        frayhip_stats o{};
        const DStats &a = dsv[0], &b = dsv[1];
        o.closest_rays = a.closest + b.closest; o.shadow_rays = a.shadow + b.shadow; o.node_tests = a.node + b.node;
        o.kd_inner_visits = a.kdInner + b.kdInner; o.leaf_refs = a.leafRefs + b.leafRefs; o.tri_tests = a.tri + b.tri;
        o.prim_tests = a.prim + b.prim; o.smooth_hits = a.smooth + b.smooth; o.samples = a.samples + b.samples;
        o.texture_fetches = a.tex + b.tex;
        float ms = 0;
        (void)hipEventElapsedTime(&ms, sc->evA, sc->evB);
        o.ms_kernels = ms;
        auto sumEvents = [&](std::vector<hipEvent_t>& pool, size_t n) {
            double t = 0;
            for (size_t i = 0; i + 1 < n; i += 2) {
                float m2 = 0;
                (void)hipEventElapsedTime(&m2, pool[i], pool[i + 1]);
                t += m2;
            }
            return t;
        };
        o.ms_trace = sumEvents(sc->evPool, nTraceEvents);
        o.trace_launches = nTraceEvents / 2;
        o.alg_bytes_trace = model(a);
        o.ms_shadow = sumEvents(sc->evPoolShadow, nShadowEvents);
        o.shadow_launches = nShadowEvents / 2;
        o.alg_bytes_shadow = model(b);
        // SURVEY 8(d) operation counts: Node::intersect ~90, triangle test ~45, primitive ~30, box test ~30 (one per node, two per inner KD node)
        auto flops = [](const DStats& d) {
            return 90.0 * (double)d.node + 45.0 * (double)d.tri + 30.0 * (double)d.prim + 30.0 * ((double)d.node + 2.0 * (double)d.kdInner);
        };
        o.alg_flops_trace = flops(a);
        o.alg_flops_shadow = flops(b);
        o.ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        *st = o;
    }
    return FRAYHIP_OK;
}

}  // namespace frayhip_detail
