// C ABI, multi-GPU exchange step: the buckets every rank rendered are gathered into rank `root`'s frame
// with direct peer -> root transfers over xGMI -- one grouped ncclRecv per peer on the root, one ncclSend on every
// other rank (RCCL; no ring: with seven point-to-point links per GPU the seven peers send concurrently on seven
// links) -- between a pack and an unpack kernel (dev_pack.hpp).  Stands behind what the reference's render() does
// when its bucket threads have all written the one shared `vfb` (src/main.cpp:360,404).
//
// RCCL is bound at run time (dlopen), so the library loads on hosts without it and a process that already holds an
// RCCL (PyTorch brings its own copy under the same soname) keeps exactly one.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "render_state.hpp"
#include "dev_pack.hpp"

namespace {

using frayhip_detail::set_error;

struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    std::string why, path;          // why binding failed; the file ncclCommInitRank was bound from
};

Rccl* rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // A process that already holds an RCCL (PyTorch maps its own copy) must keep exactly one: first ask for the copy that is
        // already loaded (RTLD_NOLOAD binds it without loading anything), and only then load one.
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        // FRAYHIP_RCCL_LIBRARY names the RCCL build to bind (a deployment's own build; the loopback stand-in of the test suite): it wins over
        // whatever the process holds, is loaded RTLD_LOCAL so that it shadows nobody else's symbols, and nothing else is tried when it fails.
        const char* forced = getenv("FRAYHIP_RCCL_LIBRARY");
        if (forced && *forced) {
            r.lib = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
            if (!r.lib) { r.why = std::string("FRAYHIP_RCCL_LIBRARY: ") + dlerror(); return; }
        }
        for (const char* n : names) {
            if (r.lib) break;
            r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);
        }
        if (!r.lib && dlsym(RTLD_DEFAULT, "ncclCommInitRank")) r.lib = dlopen(nullptr, RTLD_NOW);   // linked into the process under another name
        for (const char* n : names) {
            if (r.lib) break;
            r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        }
        if (!r.lib) { r.why = std::string("RCCL is not available: ") + dlerror(); return; }
        auto sym = [&](const char* n) { void* p = dlsym(r.lib, n); if (!p) r.why = std::string("RCCL lacks ") + n; return p; };
        r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
        r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
        r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
        r.Send = (decltype(r.Send))sym("ncclSend");
        r.Recv = (decltype(r.Recv))sym("ncclRecv");
        r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
        r.CommCount = (decltype(r.CommCount))sym("ncclCommCount");
        r.CommUserRank = (decltype(r.CommUserRank))sym("ncclCommUserRank");
        Dl_info info;
        if (r.CommInitRank && dladdr((void*)r.CommInitRank, &info) && info.dli_fname) r.path = info.dli_fname;
    });
    return r.why.empty() ? &r : nullptr;
}

int grid_for_items(size_t n)
{
    size_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    return (int)(blocks < 1 ? 1 : blocks);
}

}  // namespace

struct frayhip_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    bool owned = true;
    float* d_stage = nullptr;       // root: one packed block per peer, back to back; other ranks: this rank's packed buckets
    size_t stage_floats = 0;
    hipEvent_t stageFree = nullptr; // recorded after a gather's last use of d_stage: the next gather's stream waits for it, whatever stream that is
};

#define NCCL_TRY(expr)                                                                                          \
    do {                                                                                                        \
        ncclResult_t r_ = (expr);                                                                               \
        if (r_ != ncclSuccess) { set_error(std::string(#expr) + ": " + R->GetErrorString(r_)); return FRAYHIP_E_HIP; } \
    } while (0)

extern "C" {

int frayhip_comm_unique_id(void* id128)
{
    if (!id128) { set_error("frayhip_comm_unique_id: null argument"); return FRAYHIP_E_ARG; }
    Rccl* R = rccl();
    if (!R) { set_error("frayhip_comm_unique_id: RCCL is not available on this host"); return FRAYHIP_E_UNSUPPORTED; }
    static_assert(sizeof(ncclUniqueId) == FRAYHIP_COMM_ID_BYTES, "ncclUniqueId size");
    ncclUniqueId id;
    NCCL_TRY(R->GetUniqueId(&id));
    memcpy(id128, &id, sizeof id);
    return FRAYHIP_OK;
}

int frayhip_comm_create(const void* id128, int rank, int world, frayhip_comm** out)
{
    if (!out || world < 1 || rank < 0 || rank >= world || (world > 1 && !id128)) { set_error("frayhip_comm_create: bad argument"); return FRAYHIP_E_ARG; }
    frayhip_comm* c = new frayhip_comm();
    c->rank = rank; c->world = world;
    if (world > 1) {
        Rccl* R = rccl();
        if (!R) { delete c; set_error("frayhip_comm_create: RCCL is not available on this host"); return FRAYHIP_E_UNSUPPORTED; }
        ncclUniqueId id;
        memcpy(&id, id128, sizeof id);
        ncclResult_t r = R->CommInitRank(&c->comm, world, id, rank);     // uses the calling thread's current HIP device (frayhip_init)
        if (r != ncclSuccess) { set_error(std::string("ncclCommInitRank: ") + R->GetErrorString(r)); delete c; return FRAYHIP_E_HIP; }
    }
    *out = c;
    return FRAYHIP_OK;
}

int frayhip_comm_from_nccl(void* nccl_comm, int rank, int world, frayhip_comm** out)
{
    if (!out || !nccl_comm || world < 1 || rank < 0 || rank >= world) { set_error("frayhip_comm_from_nccl: bad argument"); return FRAYHIP_E_ARG; }
    Rccl* R = rccl();
    if (!R) { set_error("frayhip_comm_from_nccl: RCCL is not available on this host"); return FRAYHIP_E_UNSUPPORTED; }
    // the communicator's own idea of its size and of this rank must be the caller's: a gather with the wrong world deadlocks or scatters
    int n = 0, me = -1;
    NCCL_TRY(R->CommCount((ncclComm_t)nccl_comm, &n));
    NCCL_TRY(R->CommUserRank((ncclComm_t)nccl_comm, &me));
    if (n != world || me != rank) { set_error("frayhip_comm_from_nccl: the communicator says rank " + std::to_string(me) + " of " + std::to_string(n) + ", the caller rank " + std::to_string(rank) + " of " + std::to_string(world)); return FRAYHIP_E_ARG; }
    frayhip_comm* c = new frayhip_comm();
    c->comm = (ncclComm_t)nccl_comm; c->rank = rank; c->world = world; c->owned = false;
    *out = c;
    return FRAYHIP_OK;
}

void frayhip_comm_destroy(frayhip_comm* c)
{
    if (!c) return;
    if (c->d_stage) (void)hipFree(c->d_stage);
    if (c->stageFree) (void)hipEventDestroy(c->stageFree);
    if (c->comm && c->owned) if (Rccl* R = rccl()) (void)R->CommDestroy(c->comm);
    delete c;
}

int frayhip_gather_buckets(frayhip_comm* c, float* d_frame, int width, int height, int channels, int root, void* hip_stream)
{
    if (!c || !d_frame || width <= 0 || height <= 0 || channels < 1 || root < 0 || root >= c->world) { set_error("frayhip_gather_buckets: bad argument"); return FRAYHIP_E_ARG; }
    if (c->world == 1) return FRAYHIP_OK;                     // the one rank rendered every bucket in place
    Rccl* R = rccl();
    if (!R) { set_error("frayhip_gather_buckets: RCCL is not available on this host"); return FRAYHIP_E_UNSUPPORTED; }
    hipStream_t stream = (hipStream_t)hip_stream;
    DFrame F{};
    F.W = width; F.H = height; F.BW = (width - 1) / 48 + 1; F.BH = (height - 1) / 48 + 1;
    F.bucketStride = c->world;
    auto floats_of = [&](int r) { return (size_t)frayhip_bucket_count(width, height, r, c->world) * 2304 * (size_t)channels; };
    size_t need = 0;
    if (c->rank == root) { for (int r = 0; r < c->world; r++) if (r != root) need += floats_of(r); }
    else need = floats_of(c->rank);
    // one staging buffer per communicator, whatever stream a gather runs on: a gather starts after the previous one's last use of it
    if (!c->stageFree) HIP_TRY(hipEventCreateWithFlags(&c->stageFree, hipEventDisableTiming));
    else HIP_TRY(hipStreamWaitEvent(stream, c->stageFree, 0));
    if (need > c->stage_floats) {
        if (c->d_stage) { HIP_TRY(hipEventSynchronize(c->stageFree)); (void)hipFree(c->d_stage); }
        c->d_stage = nullptr; c->stage_floats = 0;
        if (hipMalloc((void**)&c->d_stage, need * sizeof(float)) != hipSuccess) { set_error("frayhip_gather_buckets: out of device memory for the staging buffer"); return FRAYHIP_E_NOMEM; }
        c->stage_floats = need;
    }
    if (c->rank != root) {
        const size_t n = floats_of(c->rank);
        if (n == 0) return FRAYHIP_OK;
        F.bucketFirst = c->rank; F.nBuckets = (int)(n / 2304 / channels);
        const int items = F.nBuckets * 2304;
        hipLaunchKernelGGL(k_pack, dim3(grid_for_items(items)), dim3(256), 0, stream, F, items, channels, d_frame, c->d_stage, 0);
        HIP_TRY(hipGetLastError());
        NCCL_TRY(R->Send(c->d_stage, n, ncclFloat, root, c->comm, stream));
        HIP_TRY(hipEventRecord(c->stageFree, stream));
        return FRAYHIP_OK;
    }
    // root: every peer's block arrives concurrently (one point-to-point link each), then goes to its pixels.  An error inside the
    // group must not leave it open (every later RCCL call of this thread -- torch's too -- would queue into a group that never ends):
    // the first failure is remembered, the group is closed, then the failure is reported.
    NCCL_TRY(R->GroupStart());
    size_t off = 0;
    ncclResult_t failed = ncclSuccess;
    for (int r = 0; r < c->world && failed == ncclSuccess; r++) {
        if (r == root) continue;
        const size_t n = floats_of(r);
        if (n) failed = R->Recv(c->d_stage + off, n, ncclFloat, r, c->comm, stream);
        off += n;
    }
    const ncclResult_t ended = R->GroupEnd();
    if (failed != ncclSuccess) { set_error(std::string("ncclRecv: ") + R->GetErrorString(failed)); return FRAYHIP_E_HIP; }
    if (ended != ncclSuccess) { set_error(std::string("ncclGroupEnd: ") + R->GetErrorString(ended)); return FRAYHIP_E_HIP; }
    off = 0;
    for (int r = 0; r < c->world; r++) {
        if (r == root) continue;
        const size_t n = floats_of(r);
        if (n) {
            F.bucketFirst = r; F.nBuckets = (int)(n / 2304 / channels);
            const int items = F.nBuckets * 2304;
            hipLaunchKernelGGL(k_pack, dim3(grid_for_items(items)), dim3(256), 0, stream, F, items, channels, d_frame, c->d_stage + off, 1);
        }
        off += n;
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(c->stageFree, stream));
    return FRAYHIP_OK;
}

// How many ranks RCCL itself sees in this communicator (ncclCommCount); a world-of-one communicator made without RCCL reports 1.
int frayhip_comm_ranks(frayhip_comm* c)
{
    if (!c) { set_error("frayhip_comm_ranks: null argument"); return FRAYHIP_E_ARG; }
    if (!c->comm) return c->world;
    Rccl* R = rccl();
    if (!R) { set_error("frayhip_comm_ranks: RCCL is not available on this host"); return FRAYHIP_E_UNSUPPORTED; }
    int n = 0;
    NCCL_TRY(R->CommCount(c->comm, &n));
    return n;
}

// The file the RCCL entry points were bound from (dladdr of ncclCommInitRank): what a host logs to say WHICH RCCL carried its frames --
// the copy PyTorch mapped, the system's, or the one FRAYHIP_RCCL_LIBRARY named.  "" when RCCL cannot be bound.
const char* frayhip_comm_library(void)
{
    Rccl* R = rccl();
    return R ? R->path.c_str() : "";
}

// Can this host exchange through RCCL at all?  Every rank asks before any rank enters ncclCommInitRank (which blocks until all
// ranks have arrived): a rank without RCCL then makes all of them choose the other transport instead of leaving the rest waiting.
int frayhip_comm_available(void)
{
    return rccl() ? 1 : 0;
}

}  // extern "C"
