// Microbenchmark (development aid): issue cost of the vector instructions the trace kernels are made of, on gfx950.
// Each kernel runs ITER x 32 independent instances of one instruction per wave (8 accumulators round-robin, so the dependent-issue latency
// is covered); timed with 8 waves/SIMD resident (throughput) and with 1 wave/SIMD (a lone wave's issue rate).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rates.hip -o /tmp/valu_rates && /tmp/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define ITER 2048

#define BODY32(ASM)                                                                                   \
    for (int i = 0; i < ITER; i++) {                                                                  \
        _Pragma("unroll") for (int k = 0; k < 4; k++) {                                               \
            asm volatile(ASM : "+v"(a0) : "v"(b)); asm volatile(ASM : "+v"(a1) : "v"(b));             \
            asm volatile(ASM : "+v"(a2) : "v"(b)); asm volatile(ASM : "+v"(a3) : "v"(b));             \
            asm volatile(ASM : "+v"(a4) : "v"(b)); asm volatile(ASM : "+v"(a5) : "v"(b));             \
            asm volatile(ASM : "+v"(a6) : "v"(b)); asm volatile(ASM : "+v"(a7) : "v"(b));             \
        }                                                                                             \
    }

#define K32(name, ASM)                                                                                \
    __global__ void name(unsigned* out, unsigned seed)                                                \
    {                                                                                                 \
        unsigned a0 = threadIdx.x + seed, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7, b = seed | 3; \
        BODY32(ASM)                                                                                   \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;           \
    }
#define K64(name, ASM)                                                                                \
    __global__ void name(unsigned* out, unsigned seed)                                                \
    {                                                                                                 \
        double a0 = threadIdx.x + seed, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7, b = 1.0000001; \
        BODY32(ASM)                                                                                   \
        out[blockIdx.x * blockDim.x + threadIdx.x] = (unsigned)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7); \
    }

K32(k_mul_lo_u32, "v_mul_lo_u32 %0, %0, %1")
K32(k_mul_u32_u24, "v_mul_u32_u24 %0, %0, %1")
K32(k_mad_u32_u24, "v_mad_u32_u24 %0, %0, %1, %0")
K32(k_add_u32, "v_add_u32 %0, %0, %1")
K32(k_xor_b32, "v_xor_b32 %0, %0, %1")
K32(k_lshl_add_u32, "v_lshl_add_u32 %0, %0, 3, %1")
K32(k_xad_u32, "v_xad_u32 %0, %0, %1, %0")
K32(k_cndmask_b32, "v_cndmask_b32 %0, %0, %1, vcc")
K32(k_fma_f32, "v_fma_f32 %0, %0, %1, %0")
K32(k_cndmask_e64, "v_cndmask_b32_e64 %0, %0, %1, s[10:11]")
K32(k_cndmask_src, "v_cndmask_b32 %0, %1, %1, vcc")
K32(k_cndmask_e64_vcc, "v_cndmask_b32_e64 %0, %0, %1, vcc")
K32(k_cmp_cndmask_vcc, "v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc")
K32(k_cmp_cndmask_sgpr, "v_cmp_lt_u32 s[10:11], %0, %1\n\tv_cndmask_b32_e64 %0, %0, %1, s[10:11]")
K32(k_addc, "v_addc_co_u32 %0, vcc, %0, %1, vcc")
K32(k_mov_b32, "v_mov_b32 %0, %1")
K32(k_and_b32, "v_and_b32 %0, %0, %1")
K32(k_bfe_u32, "v_bfe_u32 %0, %0, 3, 5")
K32(k_perm_b32, "v_perm_b32 %0, %0, %1, %1")
K32(k_readlane, "v_readfirstlane_b32 s12, %0")
K32(k_max3_f32, "v_max3_f32 %0, %0, %1, %1")

K64(k_mul_f64, "v_mul_f64 %0, %0, %1")
K64(k_add_f64, "v_add_f64 %0, %0, %1")
K64(k_fma_f64, "v_fma_f64 %0, %0, %1, %0")
K64(k_min_f64, "v_min_f64 %0, %0, %1")
K64(k_rcp_f64, "v_rcp_f64 %0, %0")
K64(k_cmp_f64, "v_cmp_lt_f64 vcc, %0, %1")

int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    unsigned* out; hipMalloc(&out, (size_t)cus * 8 * 256 * 4 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    struct { const char* n; void (*k)(unsigned*, unsigned); } ks[] = {
        {"v_mul_lo_u32", k_mul_lo_u32}, {"v_mul_u32_u24", k_mul_u32_u24}, {"v_mad_u32_u24", k_mad_u32_u24}, {"v_add_u32", k_add_u32}, {"v_xor_b32", k_xor_b32},
        {"v_lshl_add_u32", k_lshl_add_u32}, {"v_xad_u32", k_xad_u32}, {"v_cndmask_b32", k_cndmask_b32}, {"v_fma_f32", k_fma_f32}, {"v_cndmask_e64 sgpr", k_cndmask_e64}, {"v_cndmask dst!=src", k_cndmask_src}, {"cndmask_e64 vcc", k_cndmask_e64_vcc}, {"cmp+cndmask vcc (2)", k_cmp_cndmask_vcc}, {"cmp+cndmask sgpr (2)", k_cmp_cndmask_sgpr}, {"v_addc_co_u32 vcc", k_addc}, {"v_mov_b32", k_mov_b32}, {"v_and_b32", k_and_b32},
        {"v_bfe_u32", k_bfe_u32}, {"v_perm_b32", k_perm_b32}, {"v_readfirstlane", k_readlane}, {"v_max3_f32", k_max3_f32}, {"v_mul_f64", k_mul_f64},
        {"v_add_f64", k_add_f64}, {"v_fma_f64", k_fma_f64}, {"v_min_f64", k_min_f64}, {"v_rcp_f64", k_rcp_f64}, {"v_cmp_lt_f64", k_cmp_f64}};
    printf("%d CUs, clock %d MHz (nominal)\n%-20s %22s %22s\n", cus, p.clockRate / 1000, "instruction", "cycles/wave-instr @8 waves/SIMD", "@1 wave/SIMD");
    for (auto& k : ks) {
        double cyc[2];
        for (int mode = 0; mode < 2; mode++) {
            const int wavesPerSimd = mode == 0 ? 8 : 1;
            const int blocks = cus * wavesPerSimd;            // 256-thread blocks = 4 waves = one per SIMD
            hipLaunchKernelGGL(k.k, dim3(blocks), dim3(256), 0, 0, out, 1u);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(k.k, dim3(blocks), dim3(256), 0, 0, out, 1u);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double instrPerSimd = (double)ITER * 32 * wavesPerSimd;
            cyc[mode] = ms * 1e-3 * 2.4e9 / instrPerSimd;     // at 2.4 GHz: an upper estimate if the chip clocks lower
        }
        printf("%-20s %22.2f %22.2f\n", k.n, cyc[0], cyc[1]);
    }
    return 0;
}
