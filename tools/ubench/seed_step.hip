#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ uint32_t lcg_a(uint32_t x, uint32_t i) { return 1812433253u * (x ^ (x >> 30)) + i; }
__device__ __forceinline__ uint32_t lcg_b(uint32_t x, uint32_t i)
{
    uint32_t y = x ^ (x >> 30);
    unsigned long long r;
    asm("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(r) : "v"(y), "s"(1812433253u), "v"((unsigned long long)i) : "vcc");
    return (uint32_t)r;
}
template <int V> __global__ void k(uint32_t* out, uint32_t seed, int n)
{
    uint32_t b[4];
    for (int k = 0; k < 4; k++) b[k] = seed + threadIdx.x * 4 + k + blockIdx.x * 1024;
    for (int rep = 0; rep < n; rep++)
        for (uint32_t i = 1; i <= 397; i++)
#pragma unroll
            for (int k = 0; k < 4; k++) b[k] = V ? lcg_b(b[k], i) : lcg_a(b[k], i);
    out[blockIdx.x * blockDim.x + threadIdx.x] = b[0] ^ b[1] ^ b[2] ^ b[3];
}
int main()
{
    uint32_t* o; hipMalloc(&o, 256 * 8 * 256 * 4);
    uint32_t h0[4], h1[4];
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int v = 0; v < 2; v++) {
        for (int w = 0; w < 2; w++) {
            hipEventRecord(e0);
            if (v) hipLaunchKernelGGL(k<1>, dim3(2048), dim3(256), 0, 0, o, 7u, 20); else hipLaunchKernelGGL(k<0>, dim3(2048), dim3(256), 0, 0, o, 7u, 20);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (w) printf("variant %d: %.3f ms\n", v, ms);
        }
        hipMemcpy(v ? h1 : h0, o, 16, hipMemcpyDeviceToHost);
    }
    printf("same results: %d\n", h0[0] == h1[0] && h0[1] == h1[1] && h0[3] == h1[3]);
    return 0;
}
