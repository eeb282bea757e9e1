// Host-side state behind a frayhip_scene handle and the helpers shared by the translation units of
// the library: capi.hip (scene upload, C entry points) and render_variant.hip (render_impl<ST>, compiled
// once per kernel flag word so the eight variants build in parallel).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <string>
#include <vector>

#include "capi_common.h"
#include "dev_scene.hpp"
#include "dev_queues.hpp"

// A failing HIP call: the runtime's own text goes to frayhip_last_error(); the code tells allocation
// failures (FRAYHIP_E_NOMEM) and a missing device (FRAYHIP_E_NODEVICE) from everything else (FRAYHIP_E_HIP).
namespace frayhip_detail {
inline int hip_error_code(hipError_t e)
{
    switch (e) {
        case hipErrorOutOfMemory: return FRAYHIP_E_NOMEM;
        case hipErrorNoDevice: case hipErrorInvalidDevice: case hipErrorInsufficientDriver: case hipErrorNotInitialized: return FRAYHIP_E_NODEVICE;
        default: return FRAYHIP_E_HIP;
    }
}
}  // namespace frayhip_detail
#define HIP_TRY(expr)                                                                             \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            frayhip_detail::set_error(std::string(#expr) + ": " + hipGetErrorString(e_));         \
            return frayhip_detail::hip_error_code(e_);                                            \
        }                                                                                         \
    } while (0)

#ifndef FRAY_PT_BUDGET_MIB
#define FRAY_PT_BUDGET_MIB 24576   // default workspace budget of a path-traced / wavefront-Whitted frame (frayhip_scene_set_option "pt_budget_mib"): the headline frame then runs 8 batches of 8 spp on 4 lanes (338 B per path in flight)
#endif
#ifndef FRAY_PT_LANES
#define FRAY_PT_LANES 4   // headline frame / smallpt 64 spp, ms: 1 lane 150.0 / 136.3, 2 -> 136.9 / 126.5, 3 -> 135.8 / 125.1, 4 -> 135.8 / 123.8, 6 -> 135.2 / 123.9
#endif

struct frayhip_scene {
    void* d_arena = nullptr;
    size_t arena_bytes = 0;
    DScene S{};
    frayhip_camera camera{};
    frayhip_settings settings{};
    bool whittedNeedsRecursion = false;
    int specFanMax = 0;               // > 0: the largest numSamples of a glossy Refl shader, in a scene whose lights draw no random numbers (speculative glossy fans, dev_whitted.hpp)
    bool speculateFans = true;        // option "speculate_fans"
    bool lightDraws = false;          // some light draws random numbers when sampled (RectLight::getNthSample, lights.cpp:62-63)
    int lastWhittedPath = 0;          // the last Whitted frame: 0 = k_whitted (recursive shaders), 1 = shade / visible / gather launches, 2 = the fused k_wh_shade (get_option "whitted_path")
    int fusedWhittedMax = 4;          // scenes whose lights take at most this many samples per hit render Whitted frames with the one-kernel path (k_whitted) instead of shade / visible / gather launches (FRAYHIP_FUSED_WHITTED_MAX)
    int csgLanes = 1;                 // batches in flight for the Cube / CSG kernel variants (FRAYHIP_CSG_LANES; their scratch arenas are made one stream at a time, render_impl)
    unsigned warmMask = 0;            // kernel sets whose scratch arenas the lanes' streams already hold (render_impl: empty launches, one stream after the other)
    bool fpContract = false;          // option "fp_contract": path tracing past a sample's first closest hit on the kernels built with fused multiply-adds (render_contract.hip)
    long long lastContracted = 0;     // the last frame's launches of contracted kernels (frayhip_scene_get_option "contracted_launches")
    long long lastFans[4] = {0, 0, 0, 0};   // the last frame's fans filed, children traced ahead, children looked up, fans given up part of the way (frayhip_scene_get_option)
    int lightSampleCount = 0;         // sum over lights of Light::getNumSamples(): segments a Lambert / Phong hit queues (wavefront Whitted)
    bool extGeometry = false;         // Cube / CSG nodes present
    bool textured = false;            // textures or a loaded environment map present: selects the <ST | 8> variants when neither bit 1 nor bit 2 is set
    bool kdMeshes = false;            // some mesh has a KD-tree: selects the <ST | 4> kernel variants (the others are compiled without the KD walk)
    // per-frame workspace, grown on demand and kept between frames
    void* d_work = nullptr;
    size_t work_bytes = 0;
    DStats* d_stats = nullptr;
    QMeta* d_qmeta = nullptr;         // [3] segment tables: ping-pong path queues + shadow queue
    hipEvent_t evA = nullptr, evB = nullptr;
    std::vector<hipEvent_t> evPool, evPoolShadow;
    // path tracing: batches of a frame run on FRAY_PT_LANES streams at once (lane 0 = the caller's stream)
    hipStream_t laneStream[FRAY_PT_LANES] = {};
    hipEvent_t evLaneStart = nullptr, evResolved[FRAY_PT_LANES] = {};
    // tunables (frayhip_scene_set_option; defaults from FRAYHIP_PT_LANES / FRAYHIP_PT_BUDGET_MIB in the environment)
    int ptLanes = FRAY_PT_LANES;
    size_t ptBudgetBytes = (size_t)FRAY_PT_BUDGET_MIB << 20;
    // what a frame plans with: ptBudgetBytes clamped to the device's free memory ONCE (scene creation, a change of the option) and halved when an
    // allocation fails all the same -- never re-derived per frame (other processes' allocations would move the batch size, and every growth of the
    // workspace is a hipFree + hipMalloc in the middle of a run).  0 = not computed yet.
    size_t ptBudgetEff = 0;
};

namespace frayhip_detail {

// d_stats: two DStats blocks, then (256-byte aligned) the work cursors; one memset clears all of it per frame
constexpr size_t kCursorOffset = (2 * sizeof(DStats) + 255) / 256 * 256;
constexpr size_t kStatsBytes = kCursorOffset + 3 * sizeof(DCursors);     // sets of tile cursors: a batch's closest-hit and any-hit kernels; the three passes of speculative Whitted

DCamera camera_begin_frame(const frayhip_camera& c, int W, int H);
int persistent_grid(size_t n, int wavesPerSimd);
size_t work_budget(frayhip_scene* sc);
enum { FRAYHIP_RETRY_SMALLER = 1 };      // internal: ensure_work_or_shrink halved the budget, plan the frame again
int ensure_work_or_shrink(frayhip_scene* sc, size_t bytes, bool canRetry);
int bounce_grid(size_t n, bool alone);
int grid_for(size_t n);
int seed_grid(size_t n);
int ensure_work(frayhip_scene* sc, size_t bytes);
// i-th event of a pool, created on first use; nullptr (and the error text set) when hipEventCreate fails
hipEvent_t pool_event(std::vector<hipEvent_t>& pool, size_t i);

template <int ST>
int render_impl(frayhip_scene* sc, const frayhip_frame* f, float* d_rgb, int32_t* d_id, double* d_dist, hipStream_t stream, frayhip_stats* st);
extern template int render_impl<0>(frayhip_scene*, const frayhip_frame*, float*, int32_t*, double*, hipStream_t, frayhip_stats*);
extern template int render_impl<1>(frayhip_scene*, const frayhip_frame*, float*, int32_t*, double*, hipStream_t, frayhip_stats*);
extern template int render_impl<2>(frayhip_scene*, const frayhip_frame*, float*, int32_t*, double*, hipStream_t, frayhip_stats*);
extern template int render_impl<3>(frayhip_scene*, const frayhip_frame*, float*, int32_t*, double*, hipStream_t, frayhip_stats*);
extern template int render_impl<4>(frayhip_scene*, const frayhip_frame*, float*, int32_t*, double*, hipStream_t, frayhip_stats*);
extern template int render_impl<5>(frayhip_scene*, const frayhip_frame*, float*, int32_t*, double*, hipStream_t, frayhip_stats*);
extern template int render_impl<8>(frayhip_scene*, const frayhip_frame*, float*, int32_t*, double*, hipStream_t, frayhip_stats*);
extern template int render_impl<9>(frayhip_scene*, const frayhip_frame*, float*, int32_t*, double*, hipStream_t, frayhip_stats*);

}  // namespace frayhip_detail
