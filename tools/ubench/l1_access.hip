// What one wave-wide vector memory instruction costs the L1 (TCP) in "accesses", by access width and by how the lanes' addresses spread: the
// calibration of the per-instruction figures tools/pmc_finish.py derives (l1_reads_per_read_instruction, l1_writes_per_write_instruction).
// Run under rocprofv3 --pmc TCP_TOTAL_READ_sum TCP_TOTAL_WRITE_sum TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum (tools/ubench/run_l1_access.sh);
// every kernel issues exactly `iters` loads (or stores) per wave, so counter / (waves x iters) is the cost of one instruction.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/l1_access.hip -o /tmp/l1_access
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef unsigned U2 __attribute__((ext_vector_type(2)));
typedef unsigned U4 __attribute__((ext_vector_type(4)));
// stride in bytes between neighbouring lanes' addresses; T = the access type (4, 8, 16 bytes per lane)
template <class T, int STRIDE> __global__ void k_load(const char* __restrict__ base, T* __restrict__ out, int iters, size_t span)
{
    const size_t lane = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    size_t off = (lane * STRIDE) % span;
    T acc = 0;
    for (int i = 0; i < iters; i++) {
        const T v = *(const volatile T*)(base + off);
        acc ^= v;
        off = (off + 64 * 1024 * 17) % span;          // a new line every iteration, same spread
    }
    out[lane] = acc;
}
template <class T, int STRIDE> __global__ void k_store(char* __restrict__ base, int iters, size_t span)
{
    const size_t lane = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    size_t off = (lane * STRIDE) % span;
    T v = (unsigned)lane;
    for (int i = 0; i < iters; i++) {
        *(volatile T*)(base + off) = v;
        off = (off + 64 * 1024 * 17) % span;
    }
}

int main()
{
    const size_t span = (size_t)1 << 30;
    char* buf; void* out;
    hipMalloc(&buf, span + 4096); hipMalloc(&out, 2048 * 256 * 16);
    hipMemset(buf, 1, span + 4096);
    const int iters = 64, grid = 2048;
#define RUN(K, ...) hipLaunchKernelGGL((K), dim3(grid), dim3(256), 0, 0, __VA_ARGS__); hipDeviceSynchronize()
    RUN((k_load<unsigned, 4>), buf, (unsigned*)out, iters, span);       // coalesced dword: 256 B per wave
    RUN((k_load<U2, 8>), buf, (U2*)out, iters, span);                   // coalesced 8 B: 512 B per wave
    RUN((k_load<U4, 16>), buf, (U4*)out, iters, span);                  // coalesced 16 B: 1 KB per wave
    RUN((k_load<unsigned, 128>), buf, (unsigned*)out, iters, span);     // every lane its own 128-byte line
    RUN((k_load<U2, 128>), buf, (U2*)out, iters, span);
    RUN((k_load<U4, 128>), buf, (U4*)out, iters, span);
    RUN((k_store<unsigned, 4>), buf, iters, span);
    RUN((k_store<U2, 8>), buf, iters, span);
    RUN((k_store<U4, 16>), buf, iters, span);
    RUN((k_store<unsigned, 128>), buf, iters, span);
    RUN((k_store<U2, 128>), buf, iters, span);
    printf("waves per kernel %d, instructions per wave %d\n", grid * 4, iters);
    return 0;
}
