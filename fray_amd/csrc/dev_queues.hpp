// Path-tracing work queues and their segment tables: plain-pointer records shared by the kernels
// (kernels.hpp) and by the host code that carves them out of the workspace (render_impl.hpp).
#pragma once
#include <stdint.h>
#include "dev_scene.hpp"   // FRAY_RO

// One path of the queue: 96 bytes = six 16-byte loads per lane, the ray in the first three (what the closest-hit search needs), the rest of the
// state in the last three (fetched after the search).  Records, not one array per field: a wave that works through its share in the order of the
// coherence sort (kernels.hpp sort_share) reads entries in an order of its own, and a 16-byte-per-lane load costs the L1 the same 64 accesses wherever
// its lanes point, while seventeen gathered field arrays cost seventeen times 64 and as many cache lines (measured with the arrays: bounce kernel
// -11 % instructions, +20 % time, 3.3 x the L2 read requests).
struct alignas(16) PathRec {
    double o[3], d[3];
    float pm[3];                          // pathMultiplier
    uint32_t slot;                        // sample-major slot in the batch
    uint32_t depthFlags;                  // depth | flags << 16; FRAY_DEAD: no path (a slot of a ragged edge bucket outside the frame)
    uint32_t rnd[3], tab[3];              // the two generators' cursors {j, x[j], x[j + 397]}
    uint32_t pad;
};
struct PathQueue {
    PathRec* rec;
    unsigned char* cls;                   // the ray's class (kernels.hpp ray_sort_class): all the coherence sort reads of an entry
};

// Shadow (next-event) queue: segment a->b, the radiance to add if it is unobstructed, the sample slot.  One array per field: as 64-byte records (tried with
// the path queue's) the shadow kernel was 4 % (cornell_box) to 12 % (smallpt, whose shadow rays cost ~600 instructions each) slower -- it reads the six
// coordinates before the search and the rest after it, and the records' four 16-byte loads all issue up front.
struct ShadowQueue {
    double* ax; double* ay; double* az;
    double* bx; double* by; double* bz;
    float* cr; float* cg; float* cb;
    uint32_t* slot;
    unsigned char* cls;                   // the segment's direction class (kernels.hpp ray_sort_class): what the shadow kernel's coherence sort reads
};

// Whitted frames of scenes whose shaders do not recurse (Lambert / Phong / Const): what Lambert::shade / Phong::shade need from
// visible() (shading.cpp:54-78) is queued per light sample.  N = slots x eyes entries per array; task j of entry e sits at j * N + e
// (task-major: neighbouring pixels' segments to the same light sample are neighbours in the visibility kernel).
struct WhittedQueue {
    float* base;                        // [N][3] the colour that needs no visibility: light / environment / Const colour, or the ambient term
    double* ax; double* ay; double* az; // [N] the shading point every segment of the entry starts at: ip + n * 1e-6
    unsigned char* hit;                 // [N] 1: a Lambert / Phong surface was hit and its T segments are queued
    double* bx; double* by; double* bz; // [T][N] the light sample
    float* rr; float* rg; float* rb;    // [T][N] what the sample adds to its light's sum if it is visible
    unsigned char* vis;                 // [T][N] filled by k_wh_visible
};

// Radiance terms of the camera samples of a batch.  pathtrace() returns contribLight + (pathtrace of the next bounce)
// (main.cpp:240-243): a sample's value is term_0 + (term_1 + (... + term_last)), FP32 additions from the INNERMOST
// outwards.  Bounce b of a path writes its term -- the next-event contribution, or what ended the path -- to
// t[(b * 3 + channel) * nPaths + slot]; n[slot] = number of terms; k_pt_fold adds them up in the reference's order.
struct TermBuf {
    float* t;
    unsigned short* n;
    uint32_t nPaths;
    int b;             // the bounce the launch belongs to
};

#ifndef FRAY_MAXSEG
#define FRAY_MAXSEG 32768  // = 8192 blocks x 4 waves, the largest grid bounce_grid() returns (the Cube / CSG variants, whose batches run one at a time)
#endif
// A producing wave owns one segment of `chunk` entries and fills it from BOTH ends: rays that miss every gate (dev_scene.hpp DGate) from the
// front, the others from the back.  cnt[w] = entries of wave w's segment, nf[w] = how many of them sit at the front; entry e of the segment
// (dense order: the front ones in order, then the back ones, last written first) lives at w * chunk + (e < nf ? e : chunk - 1 - (e - nf))
// (seg_slot_of, kernels.hpp).  (Two table entries per segment -- front run, back run -- measured +0.2 ms per launch at 8192 segments.)
struct alignas(256) QMeta {
    uint32_t n;        // live paths in the queue
    uint32_t chunk;    // capacity (and stride) of one segment
    uint32_t nSeg;
    uint32_t certFront; // 1: the entries at the FRONT of the segments were certified gate-free by their producer (DScene::gatesExact); 0: a dense queue, or a hint only
    uint32_t cnt[FRAY_MAXSEG];
    uint32_t nf[FRAY_MAXSEG];
    uint32_t off[FRAY_MAXSEG + 1];
};

// The same table as a kernel argument that is only read: typed into the constant address space on the device, so
// the offsets come through the scalar cache (as the scene tables do, dev_scene.hpp).
struct QMetaRO { const FRAY_RO QMeta* p; };

// Right-eye state parked by the left pass of a stereo path-traced frame (kernels.hpp, k_pt_init).
struct StereoBuf {
    double* r[6];      // right-eye ray: origin xyz, direction xyz
    uint32_t* g[6];    // rnd {j, a, b}, tab {j, a, b} at the end of the left path
};

