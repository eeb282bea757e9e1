// Path-tracing work queues and their segment tables: plain-pointer records shared by the kernels
// (kernels.hpp) and by the host code that carves them out of the workspace (render_impl.hpp).
#pragma once
#include <stdint.h>
#include "dev_scene.hpp"   // FRAY_RO

struct PathQueue {
    double* ox; double* oy; double* oz;
    double* dx; double* dy; double* dz;
    float* tr; float* tg; float* tb;      // pathMultiplier
    uint32_t* slot;                       // sample-major slot in the batch
    uint32_t* depthFlags;                 // depth | flags << 16
    uint32_t* rndJ; uint32_t* rndA; uint32_t* rndB;
    uint32_t* tabJ; uint32_t* tabA; uint32_t* tabB;
};

// Shadow (next-event) queue: segment a->b, the radiance to add if it is unobstructed, the sample slot.
struct ShadowQueue {
    double* ax; double* ay; double* az;
    double* bx; double* by; double* bz;
    float* cr; float* cg; float* cb;
    uint32_t* slot;
};

#ifndef FRAY_MAXSEG
#define FRAY_MAXSEG 8192   // = 2048 blocks x 4 waves, the largest grid grid_for() returns
#endif
struct QMeta {
    uint32_t n;        // live paths in the queue
    uint32_t chunk;    // capacity (and stride) of one segment
    uint32_t nSeg;
    uint32_t pad;
    uint32_t cnt[FRAY_MAXSEG];
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

