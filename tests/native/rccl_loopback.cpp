// TEST INFRASTRUCTURE -- a loopback stand-in for the ten RCCL entry points fray_amd/csrc/capi_comm.hip binds at run time
// (ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy, ncclGroupStart, ncclGroupEnd, ncclSend, ncclRecv, ncclGetErrorString,
// ncclCommCount, ncclCommUserRank), so that frayhip_gather_buckets' world > 1 branch -- pack, send; grouped receives at
// accumulated offsets, per-peer unpack -- can EXECUTE on a one-GPU box, where RCCL itself refuses two ranks on one device
// ("Duplicate GPU detected").  Never part of the product: the library only binds it when FRAYHIP_RCCL_LIBRARY names it
// (tests/test_gpu_gather_loopback.py), and nothing under fray_amd/ refers to it.
//
// Ranks may be threads of one process or processes of one host, all driving the same GPU.  A message travels
//   sender:   hipStreamSynchronize(stream) -> hipMemcpy D2H into a file under /dev/shm -> rename() publishes it
//   receiver: waits for the file -> hipMemcpy H2D on `stream` -> unlink
// keyed by (communicator id, source, destination, sequence number per pair), so messages of one pair arrive in the order
// they were sent, as RCCL's do.  Inside a group calls are queued and run at ncclGroupEnd, sends first.  Every wait gives up
// after FRAY_LOOPBACK_TIMEOUT_S (default 120) seconds with ncclSystemError: a test fails, it does not hang.
#define __HIP_PLATFORM_AMD__
#include <fcntl.h>
#include <hip/hip_runtime_api.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

extern "C" {

typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4, ncclInvalidUsage = 5 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5, ncclFloat16 = 6, ncclFloat32 = 7, ncclFloat64 = 8, ncclBfloat16 = 9 } ncclDataType_t;
typedef struct { char internal[128]; } ncclUniqueId;
struct ncclComm {
    char id[40];
    int rank, world;
    std::vector<unsigned> sent, received;      // per peer: messages so far
};
typedef struct ncclComm* ncclComm_t;

}  // extern "C"

namespace {

double timeout_s()
{
    const char* e = getenv("FRAY_LOOPBACK_TIMEOUT_S");
    const double v = e ? atof(e) : 0;
    return v > 0 ? v : 120.0;
}

size_t size_of(ncclDataType_t t)
{
    switch (t) {
        case ncclInt8: case ncclUint8: return 1;
        case ncclFloat16: case ncclBfloat16: return 2;
        case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
        default: return 8;
    }
}

std::string path_of(const ncclComm* c, const char* what, int a, int b, unsigned seq)
{
    char buf[256];
    snprintf(buf, sizeof buf, "/dev/shm/frayloop_%s_%s_%d_%d_%u", c->id, what, a, b, seq);
    return buf;
}

bool wait_for(const std::string& p)
{
    const auto t0 = std::chrono::steady_clock::now();
    struct stat st;
    while (stat(p.c_str(), &st) != 0) {
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s()) return false;
        usleep(100);
    }
    return true;
}

struct Op { bool send; void* buf; size_t bytes; int peer; ncclComm* comm; hipStream_t stream; };
thread_local int g_depth = 0;
thread_local std::vector<Op> g_queue;

ncclResult_t run(const Op& o)
{
    ncclComm* c = o.comm;
    if (o.send) {
        const unsigned seq = c->sent[o.peer]++;
        const std::string fin = path_of(c, "msg", c->rank, o.peer, seq), tmp = fin + ".part";
        const int fd = open(tmp.c_str(), O_CREAT | O_RDWR | O_TRUNC, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)o.bytes) != 0) { if (fd >= 0) close(fd); return ncclSystemError; }
        void* m = o.bytes ? mmap(nullptr, o.bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0) : nullptr;
        close(fd);
        if (o.bytes && m == MAP_FAILED) return ncclSystemError;
        // stream order: everything queued on `stream` before the send (the pack kernel) has run before the bytes leave
        if (hipStreamSynchronize(o.stream) != hipSuccess) return ncclUnhandledCudaError;
        if (o.bytes && hipMemcpy(m, o.buf, o.bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
        if (m) munmap(m, o.bytes);
        if (rename(tmp.c_str(), fin.c_str()) != 0) return ncclSystemError;
        return ncclSuccess;
    }
    const unsigned seq = c->received[o.peer]++;
    const std::string fin = path_of(c, "msg", o.peer, c->rank, seq);
    if (!wait_for(fin)) return ncclSystemError;
    const int fd = open(fin.c_str(), O_RDONLY);
    struct stat st;
    if (fd < 0 || fstat(fd, &st) != 0) { if (fd >= 0) close(fd); return ncclSystemError; }
    if ((size_t)st.st_size != o.bytes) { close(fd); unlink(fin.c_str()); return ncclInvalidArgument; }     // send and receive counts must match
    void* m = o.bytes ? mmap(nullptr, o.bytes, PROT_READ, MAP_SHARED, fd, 0) : nullptr;
    close(fd);
    if (o.bytes && m == MAP_FAILED) return ncclSystemError;
    ncclResult_t r = ncclSuccess;
    // stream order: the bytes land after whatever `stream` held before the receive and before whatever is queued after it
    if (o.bytes && (hipMemcpyAsync(o.buf, m, o.bytes, hipMemcpyHostToDevice, o.stream) != hipSuccess || hipStreamSynchronize(o.stream) != hipSuccess)) r = ncclUnhandledCudaError;
    if (m) munmap(m, o.bytes);
    unlink(fin.c_str());
    return r;
}

ncclResult_t issue(const Op& o)
{
    if (!o.comm || o.peer < 0 || o.peer >= o.comm->world || o.peer == o.comm->rank) return ncclInvalidArgument;
    if (g_depth > 0) { g_queue.push_back(o); return ncclSuccess; }
    return run(o);
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id)
{
    if (!id) return ncclInvalidArgument;
    static std::atomic<unsigned> counter{0};
    memset(id, 0, sizeof *id);
    unsigned long long r = 0;
    if (FILE* f = fopen("/dev/urandom", "rb")) { if (fread(&r, sizeof r, 1, f) != 1) r = 0; fclose(f); }
    snprintf(id->internal, sizeof id->internal, "%016llx%08x%04x", r, (unsigned)getpid(), counter++ & 0xffffu);
    return ncclSuccess;
}

// every rank announces itself under the id and waits for the others: like the real call, it returns when all have arrived
ncclResult_t ncclCommInitRank(ncclComm_t* out, int world, ncclUniqueId id, int rank)
{
    if (!out || world < 1 || rank < 0 || rank >= world) return ncclInvalidArgument;
    ncclComm* c = new ncclComm();
    memcpy(c->id, id.internal, sizeof c->id - 1);
    c->id[sizeof c->id - 1] = 0;
    for (char* p = c->id; *p; p++) if (!((*p >= '0' && *p <= '9') || (*p >= 'a' && *p <= 'f'))) *p = 'x';     // the id becomes part of a file name
    c->rank = rank; c->world = world;
    c->sent.assign(world, 0); c->received.assign(world, 0);
    const std::string mine = path_of(c, "here", rank, world, 0);
    const int fd = open(mine.c_str(), O_CREAT | O_EXCL | O_WRONLY, 0600);
    if (fd < 0) { delete c; return ncclInvalidUsage; }            // two ranks with the same number under one id
    close(fd);
    for (int r = 0; r < world; r++)
        if (!wait_for(path_of(c, "here", r, world, 0))) { unlink(mine.c_str()); delete c; return ncclSystemError; }
    *out = c;
    return ncclSuccess;
}

// A rank may leave while another has not yet seen it arrive (a sender is done after its one send): the "here" files stay until
// EVERY rank has left; the last one to go removes them.
ncclResult_t ncclCommDestroy(ncclComm_t c)
{
    if (!c) return ncclInvalidArgument;
    const int fd = open(path_of(c, "gone", c->rank, c->world, 0).c_str(), O_CREAT | O_WRONLY, 0600);
    if (fd >= 0) close(fd);
    bool all = true;
    struct stat st;
    for (int r = 0; r < c->world && all; r++) all = stat(path_of(c, "gone", r, c->world, 0).c_str(), &st) == 0;
    if (all)
        for (int r = 0; r < c->world; r++) { unlink(path_of(c, "here", r, c->world, 0).c_str()); unlink(path_of(c, "gone", r, c->world, 0).c_str()); }
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclGroupStart() { g_depth++; return ncclSuccess; }

ncclResult_t ncclGroupEnd()
{
    if (g_depth <= 0) return ncclInvalidUsage;
    if (--g_depth > 0) return ncclSuccess;
    std::vector<Op> q;
    q.swap(g_queue);
    ncclResult_t first = ncclSuccess;
    for (int pass = 0; pass < 2; pass++)             // sends first: a group that sends and receives must not wait on itself
        for (const Op& o : q)
            if (o.send == (pass == 0)) { const ncclResult_t r = run(o); if (first == ncclSuccess) first = r; }
    return first;
}

ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t s)
{
    return issue(Op{true, const_cast<void*>(buf), count * size_of(t), peer, c, s});
}

ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t s)
{
    return issue(Op{false, buf, count * size_of(t), peer, c, s});
}

const char* ncclGetErrorString(ncclResult_t r)
{
    switch (r) {
        case ncclSuccess: return "no error (loopback stand-in)";
        case ncclUnhandledCudaError: return "unhandled HIP error (loopback stand-in)";
        case ncclSystemError: return "system error or timeout (loopback stand-in)";
        case ncclInvalidArgument: return "invalid argument (loopback stand-in)";
        case ncclInvalidUsage: return "invalid usage (loopback stand-in)";
        default: return "internal error (loopback stand-in)";
    }
}

ncclResult_t ncclCommCount(const ncclComm_t c, int* n) { if (!c || !n) return ncclInvalidArgument; *n = c->world; return ncclSuccess; }
ncclResult_t ncclCommUserRank(const ncclComm_t c, int* r) { if (!c || !r) return ncclInvalidArgument; *r = c->rank; return ncclSuccess; }

// what the tests ask to make sure the library bound THIS file and not the RCCL PyTorch maps
int fray_loopback_marker(void) { return 0x10095ac; }

}  // extern "C"
