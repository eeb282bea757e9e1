// Attempt at a small reproducer of round 3's wrong pictures in the Cube / CSG kernel variants (k_whitted<2>, k_pt_shadow<2>): those were the only
// variants with OUT-OF-LINE device calls (sixteen CsgOp levels), they spilled 100-350 SGPRs, and they rendered correctly with
// -mllvm -amdgpu-spill-sgpr-to-vgpr=false (SGPR spills to scratch memory instead of VGPR lanes).  The suspected mechanism: the caller keeps spilled
// SGPRs in lanes of a VGPR (written with v_writelane, which ignores EXEC); it calls a noinline function under a PARTIAL exec mask; the callee uses that
// VGPR as an ordinary callee-saved register, i.e. saves and restores it under the CURRENT exec mask -- the lanes that are inactive during the call
// come back with whatever the callee left in them, and the caller's v_readlane of such a lane reads garbage.
//
// This program builds that situation synthetically: a caller with far more live wave-uniform values than SGPRs (so that they are spilled), a noinline
// callee with high VGPR pressure (so that it clobbers many callee-saved VGPRs), called by HALF of the lanes; the caller's uniform values are used
// after the call and summed into a checksum that the host recomputes.
//   hipcc --offload-arch=gfx950 -O3 tools/repro/sgpr_spill_call.hip -o /tmp/repro && /tmp/repro
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-spill-sgpr-to-vgpr=false ... (the control)
// Exit code 1 and a line "MISMATCH" if the device checksum differs from the host's.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define NU 160          // wave-uniform values the caller keeps alive across the call

__device__ __attribute__((noinline)) double heavy(const double* __restrict__ p, double x, int n)
{
    // many independent accumulators: VGPR pressure well into the callee-saved range
    double a[48];
#pragma unroll
    for (int i = 0; i < 48; i++) a[i] = p[i] * x + (double)i;
    for (int k = 0; k < n; k++) {
#pragma unroll
        for (int i = 0; i < 48; i++) a[i] = a[i] * 1.0000001 + a[(i + 7) % 48] * 1e-9;
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 48; i++) s += a[i];
    return s;
}

__global__ void k(const __attribute__((address_space(4))) uint32_t* __restrict__ u, const double* __restrict__ tab, double* __restrict__ out, int n, int rounds)
{
    // NU wave-uniform values, loaded through the scalar cache into SGPRs: more than the 100-odd the register file holds
    uint32_t v[NU];
#pragma unroll
    for (int i = 0; i < NU; i++) v[i] = u[i];
    const int lane = threadIdx.x & 63;
    double acc = 0;
    for (int r = 0; r < rounds; r++) {
        // a call under a partial exec mask: lanes whose bit of a changing pattern is set
        if (((0x9E3779B97F4A7C15ull >> ((lane + r) & 63)) & 1ull) != 0) acc += heavy(tab + (r & 7), (double)(lane + 1), n);
        // the uniform values are needed again after the call, by ALL lanes
        uint32_t h = 0;
#pragma unroll
        for (int i = 0; i < NU; i++) h = h * 31u + (v[i] ^ (uint32_t)(r + i));
        acc += (double)(h & 0xffffu);
#pragma unroll
        for (int i = 0; i < NU; i++) v[i] = v[i] * 1664525u + 1013904223u;      // keep them live and changing (still uniform)
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

static double heavy_host(const double* p, double x, int n)
{
    double a[48];
    for (int i = 0; i < 48; i++) a[i] = p[i] * x + (double)i;
    for (int k = 0; k < n; k++)
        for (int i = 0; i < 48; i++) a[i] = a[i] * 1.0000001 + a[(i + 7) % 48] * 1e-9;
    double s = 0;
    for (int i = 0; i < 48; i++) s += a[i];
    return s;
}

int main()
{
    const int n = 3, rounds = 9, blocks = 64, threads = 256;
    std::vector<uint32_t> u(NU);
    for (int i = 0; i < NU; i++) u[i] = 0x12345u * (i + 1) + 77u;
    std::vector<double> tab(64);
    for (int i = 0; i < 64; i++) tab[i] = 1.0 + i * 0.03125;
    uint32_t* du; double *dt, *dout;
    (void)hipMalloc(&du, NU * 4); (void)hipMalloc(&dt, 64 * 8); (void)hipMalloc(&dout, blocks * threads * 8);
    (void)hipMemcpy(du, u.data(), NU * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dt, tab.data(), 64 * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, (const __attribute__((address_space(4))) uint32_t*)du, dt, dout, n, rounds);
    std::vector<double> out(blocks * threads);
    if (hipMemcpy(out.data(), dout, out.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) { printf("kernel failed\n"); return 2; }
    long bad = 0;
    for (int t = 0; t < blocks * threads; t++) {
        const int lane = t & 63;
        std::vector<uint32_t> v(u);
        double acc = 0;
        for (int r = 0; r < rounds; r++) {
            if (((0x9E3779B97F4A7C15ull >> ((lane + r) & 63)) & 1ull) != 0) acc += heavy_host(tab.data() + (r & 7), (double)(lane + 1), n);
            uint32_t h = 0;
            for (int i = 0; i < NU; i++) h = h * 31u + (v[i] ^ (uint32_t)(r + i));
            acc += (double)(h & 0xffffu);
            for (int i = 0; i < NU; i++) v[i] = v[i] * 1664525u + 1013904223u;
        }
        if (acc != out[t]) { if (bad < 5) printf("thread %d: device %.17g host %.17g\n", t, out[t], acc); bad++; }
    }
    printf("%s: %ld of %d threads differ\n", bad ? "MISMATCH" : "match", bad, blocks * threads);
    return bad ? 1 : 0;
}
