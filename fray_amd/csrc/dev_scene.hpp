// Device-side scene layout (gfx950).  Built once per scene by upload.cpp-side code in capi.hip
// from frayhip_scene_desc; everything lives in one HBM arena and is read-only during a frame.
//
// Layout choices (DESIGN.md "Data layout in HBM"):
//  * per-triangle test record = exactly the 15 doubles Triangle::intersectFast + the culling test
//    read (gnormal, A, ABxAC, AC, AB: SURVEY 8(a) a9), padded to 128 B so a lane fetches it with
//    eight aligned 16-byte loads; shading attributes (pre-gathered corner normals / uvs, dNdx,
//    dNdy) live in a separate array touched only for the winning triangle;
//  * KD nodes are split into a 16-byte hot record {split | leaf range, child0, axis + leaf flags} (DKd: all a step of the walk reads) and
//    a 48-byte box (DKdBox: read when a pending child is taken from the walk's stack, when a leaf's hit has to pass inside(), and when a
//    child test falls within the margins of the certified box test; the reference recomputes child boxes by BBox::split on the way down,
//    mesh.cpp:373-376);
//  * nodes / lights / shaders / textures are tiny tables indexed wave-uniformly, so they are
//    fetched through the scalar cache.
#pragma once
#include <stdint.h>
#include "frayhip.h"      // FRAYHIP_BUCKET_SKEW: the bucket numbering is part of the C ABI
#include "dev_tricert.hpp"  // DTri32

// Scene tables are read-only for the whole frame.  On the device their pointers are typed into the
// constant address space (4): loads through them are known not to alias the kernels' stores, so a
// wave-uniform access (node loop, brute-force triangle loop, light / shader tables) becomes a
// scalar-cache s_load into SGPRs instead of a 64-lane vector load, and the compiler may batch them.
#ifdef __HIP_DEVICE_COMPILE__
#define FRAY_RO __attribute__((address_space(4)))
#else
#define FRAY_RO
#endif

// Envelope of the device CSG code (dev_trace.hpp csg_intersect); frayhip_scene_create checks scenes against it.
#define FRAY_CSG_MAX 30   // intersections kept per operand: the reference's own limit (geometry.cpp:144)
#ifndef FRAY_CSG_DEPTH
#define FRAY_CSG_DEPTH 16 // CsgOp activations the device's stack holds (dev_trace.hpp csg_intersect): 1 = operands are plain geometries; deeper scenes are rejected at upload
#endif

// The KD walk's stack of pending children (dev_trace.hpp): the 16 most recent entries of a lane in LDS, older ones in scratch; deeper trees are
// rejected at upload (the reference's builder stops at depth 65, constants.h:39).
#define FRAY_KD_LDS_STACK 16
#define FRAY_KD_MAX_DEPTH 72

struct DXform { double off[3]; double m[9]; double inv[9]; };

struct DTri;
struct DNode {
    DXform T;
    int32_t geomKind, geomIndex, shader, bumpTex;
    int32_t xfClass;      // index of the first node whose {offset, m, invM} are bitwise equal: nodes of one class see the
                          // same local ray, so it is computed once per ray (e.g. the 7 untransformed Cornell-box meshes)
    // A mesh WITHOUT a KD-tree (Mesh::intersect's brute-force loop, mesh.cpp:157-161: at most 20 triangles) carries a
    // copy of what its root box test and triangle loop read, so neither needs a second dependent load through DMesh:
    int32_t tlTris;       // number of triangles; 0 for every other geometry
    int32_t tlCulling;
    int32_t xfIdentity;   // offset = 0 and m = invM = I, bit for bit: the local ray is the world ray (node_intersect)
    double bmin[3], bmax[3];
    const FRAY_RO DTri* tlPtr;
    double boxMax;        // max |coordinate| of bmin / bmax (margins of the certified box test, dev_boxcert.hpp)
    int32_t gated, padN;  // this node's geometry lies inside an EXACT gate (DGate::exact): a ray its producer certified to miss every gate skips the node
};                        // 272 B
// What only the box test of the scenes WITHOUT KD meshes reads (DScene::nodesX): the node's box widened by inside()'s tolerance
struct DNodeX {
    double bminE[3], bmaxE[3];   // bmin - 1e-6, bmax + 1e-6: what BBox::inside compares with (bbox.h:81-83), computed once on the host
    // A CsgOp node whose tree is bounded (no unbounded operand where it matters): centre and half extents of a LOCAL-space box that holds every
    // geometry an intersection of the tree can come from, widened by 1e-5, and M = max_k (|c_k| + h_k) -- what ray_surely_misses_box
    // (dev_misscert.hpp) needs to let a ray that passes the object by skip CsgOp::intersect altogether.  csgBox = 0: no such box.
    double cc[3], ch[3], cM;
    int32_t csgBox, padX;
};

struct DPlane { double limit, height; };
struct DSphere { double O[3]; double R; };
struct DCube { double O[3]; double halfSide; };
struct DCsg { int32_t op, leftKind, leftIndex, rightKind, rightIndex, leftGeom, rightGeom, flat; };   // kinds/indices as in DNode; flat: both operands are planes / spheres / cubes (csg_flat_intersect)

struct DTri {          // 128 B
    double g[3];       // gnormal
    double A[3];       // vertex A
    double N[3];       // AB x AC
    double AC[3];
    double AB[3];
    int32_t index;     // the triangle's index in its mesh (leaf-packed copies, DMesh::ltris, are found by position, not by index)
    int32_t pad;
};

struct DTriAttr {      // 168 B, winning triangle only
    double nA[3], nB[3], nC[3];
    double tA[2], tB[2], tC[2];
    double dNdx[3], dNdy[3];
};

// KD nodes in the builder's depth-first order, hot and cold halves in arrays of their own: a walking lane reads 16 bytes per inner node
// and nothing else, so eight nodes share a 128-byte line instead of one and a third (incoherent walks live on what the caches hold).
struct alignas(16) DKd {   // 16 B: what a step of the walk reads
    union {
        double split;                            // inner node
        struct { int32_t triBegin, triCount; };  // leaf: its range of DMesh::ltris / ltris32
    };
    int32_t child0;
    int32_t meta;      // axis (bits 0-1; 3 = leaf) | inner: bit 2+c set = child c is a leaf
};
// the node's own box, exactly the coordinates BBox::split hands down (bbox.h:205-211).  A walking lane carries the ray's
// parameter interval against the current box (dev_boxcert.hpp), not the box: the coordinates are read when a pending node is taken
// from the stack, when a leaf's hit has to pass inside(), and when a child test has to run the reference's own arithmetic.
struct DKdBox { double lo[3], hi[3]; };   // 48 B

struct DMesh {
    double bmin[3], bmax[3];
    const FRAY_RO DTri* tris;
    const FRAY_RO DTriAttr* attrs;
    const FRAY_RO DKd* kd;
    const FRAY_RO DKdBox* kdBox;
    const FRAY_RO int32_t* refs;
    // KD meshes: the triangle records again, one copy per leaf reference in leaf order (triBegin .. triBegin + triCount), so a
    // leaf's triangles are consecutive 128-byte records instead of an index list into `tris` (one dependent load less per triangle)
    const FRAY_RO DTri* ltris;
    int32_t nTris, hasKd, smooth, culling, hasUV, pad;
    double boxMax;     // max |coordinate| of the bounding box (margins of the certified box test, dev_boxcert.hpp)
    // beside ltris, same order: the FP32 records of the certified "surely misses" filter (dev_tricert.hpp), relative to `ref`
    const FRAY_RO DTri32* ltris32;
    double ref[3];     // centre of the bounding box
};

struct DTexture {
    int32_t kind, width, height, pad;
    float color1[3], color2[3];
    double scaling, bumpIntensity, ior;
    const FRAY_RO float* texels;
};

struct DShader {
    int32_t kind, texture;
    float color[3], specularColor[3], mult[3];
    int32_t numSamples;
    double exponent, specularMultiplier, glossiness, deflectionScaling, ior;
    int32_t layerBegin, layerCount;
    int32_t usesUV, pad;   // some texture in this shader (or its layers) reads info.u / info.v
};

struct DLayer { int32_t shader, texture; float opacity[3]; int32_t pad; };

struct DLight {
    int32_t kind, xSubd, ySubd, pad;
    float color[3], power;
    double pos[3];
    DXform T;
    double center[3];
    double area;
    double areaXsize, areaYsize;   // 1.0 / xSubd, 1.0 / ySubd (RectLight::getNthSample, lights.cpp:54-55), divided once on the host
};

struct DEnv {
    int32_t present, loaded;
    int32_t width[6], height[6];
    const FRAY_RO float* face[6];
};

// Per-frame camera state, Camera::beginFrame (camera.cpp:34-57), computed on the host.
struct DCamera {
    double topLeft[3], topRight[3], bottomLeft[3];
    double frontDir[3], upDir[3], rightDir[3], pos[3];
    double w, h, apertureSize, focalPlaneDist, stereoSeparation;
    float leftMask[3], rightMask[3];
    int32_t dof, pad;
};

// Scheduling hint for the path tracer's queues (kernels.hpp, ray_gate_class): the world-space bounding boxes of the nodes whose triangle loops
// are worth skipping for a whole wave -- meshes without a KD-tree and with FRAY_GATE_MIN_TRIS triangles or more (cornell_box: the two
// blocks).  A ray that misses all of them goes to the front of its wave's queue segment, the others to the back, so that the waves of the next
// launch are (nearly) all-miss or all-pass and the all-miss ones skip those loops at the wave-uniform root box test.  A hint only: every ray
// still runs the reference's tests on every node.
#define FRAY_MAX_GATES 8
#define FRAY_GATE_MIN_TRIS 6
// exact: the gate's box is the geometry's own box in the space the reference tests it in (an untransformed node), and cf / hf / Mf are what
// ray_surely_misses_box_f32 (dev_misscert.hpp) needs: FP32 centre, half extents widened by 1e-5 and by the centre's rounding, rounded up, and max (|cf| + hf).
// When EVERY gate of a scene is exact (DScene::gatesExact) "gate-free" is a proof, not a hint, and the consumers skip the gated nodes for such rays.
struct DGate { double lo[3], hi[3]; float cf[3], hf[3]; float Mf; int32_t exact; };

struct DScene {
    const FRAY_RO DNode* nodes;
    const FRAY_RO DNodeX* nodesX;
    const FRAY_RO DGate* gates;
    int32_t nGates, gatesExact;
    const FRAY_RO DPlane* planes;
    const FRAY_RO DSphere* spheres;
    const FRAY_RO DCube* cubes;
    const FRAY_RO DCsg* csgs;
    const FRAY_RO DMesh* meshes;
    const FRAY_RO DShader* shaders;
    const FRAY_RO DLayer* layers;
    const FRAY_RO DTexture* textures;
    const FRAY_RO DLight* lights;
    DEnv env;
    int32_t nNodes, nLights;
    float probPickLight;           // 1.0f / lights.size() (main.cpp:160)
    float ambient[3];
    int32_t maxTraceDepth, gi;
    float saturation;
};

// Work decomposition of a frame: the reference's 48x48 buckets (sdl.cpp:243-262), of which this
// rank owns b = first + k*stride.  Work item `i` of a frame maps to bucket k = i / 2304.
struct DFrame {
    int32_t W, H, BW, BH;
    int32_t bucketFirst, bucketStride, nBuckets /* owned */, spp;
    uint32_t seed;
    int32_t jitter;     // dof || gi: pixel offsets come from the RNG (main.cpp:351-357)
};

// Device counters mirroring frayhip_stats (only maintained by the *_stats kernel variants).
struct DStats {
    unsigned long long closest, shadow, node, kdInner, leafRefs, tri, prim, smooth, samples, tex, rngOverflow;   // rngOverflow: any "left the supported envelope" event
#ifdef FRAY_STAMPS
    unsigned long long stamp[24];   // diagnostic build: wave cycles per section (dev_math.hpp STAMP)
    unsigned long long stampLanes[24];
#endif
};
// Work cursors of the persistent kernels (kernels.hpp claim_items): one per cache line.
struct DCursors { unsigned int v[8][32]; };
