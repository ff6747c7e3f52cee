// Micro-benchmark: VALU issue rate per SIMD for the instruction kinds the ray loop uses.
// 256 CUs x 16 waves (4 per SIMD, like k_rays_skip), N independent chains per lane.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int ITERS = 4096;
template <int KIND>
__global__ __launch_bounds__(1024) void k(uint32_t *out, uint32_t seed)
{
    uint32_t a[8];
    float f[8];
    double d[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = seed + threadIdx.x * 7 + i; f[i] = (float)a[i]; }
#pragma unroll
    for (int i = 0; i < 4; ++i) d[i] = (double)a[i];
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (KIND == 0) a[i] = a[i] + (a[(i + 1) & 7] ^ 0x55);                 // v_xor + v_add (2 ops) -> count 2
            if (KIND == 1) a[i] = (a[i] >> 3) + 1;                                 // v_lshrrev + v_add  -> 2  (maybe lshl_add fused)
            if (KIND == 2) a[i] = __mul24((int)a[i], 77) + (int)a[(i + 1) & 7];    // v_mad_i32_i24 -> 1
            if (KIND == 3) f[i] = __builtin_fmaf(f[i], 1.0001f, f[(i + 1) & 7]);   // v_fma_f32 -> 1
            if (KIND == 4) a[i] = min(a[i], min(a[(i + 1) & 7], a[(i + 2) & 7] + it)); // v_add + v_min3 -> 2
            if (KIND == 6) a[i] = a[i] * 77u + 3u;                                 // v_mul_lo_u32 + add (mad_u64?) 
            if (KIND == 7) a[i] = (a[i] & 0x1fc) | (a[(i+1)&7] << 2);              // v_and, v_lshl_or -> 2
        }
        if (KIND == 5) {
#pragma unroll
            for (int i = 0; i < 4; ++i) d[i] = __builtin_fma(d[i], 1.0000001, d[(i + 1) & 3]);   // v_fma_f64 -> 1 (x4)
        }
        if (KIND == 8) {
#pragma unroll
            for (int i = 0; i < 8; i += 2) {   // packed fp32 fma: 2 flops-lanes per op
                typedef float float2v __attribute__((ext_vector_type(2)));
                float2v x = {f[i], f[i + 1]}, y = {1.0001f, 1.0002f}, z = {f[(i + 2) & 7], f[(i + 3) & 7]};
                x = __builtin_elementwise_fma(x, y, z);
                f[i] = x[0]; f[i + 1] = x[1];
            }
        }
    }
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) r += a[i] + (uint32_t)f[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) r += (uint32_t)d[i];
    out[blockIdx.x * 1024 + threadIdx.x] = r;
}
template <int KIND>
int run(const char *name, double ops_per_iter, uint32_t *out)
{
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(1024), 0, 0, out, 1u);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(1024), 0, 0, out, 2u);
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    // per SIMD: 4 waves x ITERS x ops_per_iter wave-instructions
    double instr = 4.0 * ITERS * ops_per_iter;
    double cyc = ms * 1e-3 * 2.35e9;
    printf("%-28s %8.3f ms  %6.2f cycles per wave-instruction per SIMD (at 2.35 GHz)\n", name, ms, cyc / instr);
    return 0;
}
int main()
{
    uint32_t *out; CHK(hipMalloc(&out, 256 * 1024 * 4));
    run<0>("xor+add (2 ops x8)", 16, out);
    run<1>("lshr+add (2 ops x8)", 16, out);
    run<2>("mad_i32_i24 (x8)", 8, out);
    run<3>("fma_f32 (x8)", 8, out);
    run<4>("add+min3 (2 ops x8)", 16, out);
    run<5>("fma_f64 (x4)", 4, out);
    run<6>("mul_lo_u32+add (x8)", 8, out);
    run<7>("and + lshl_or (2 ops x8)", 16, out);
    run<8>("pk_fma_f32 (x4)", 4, out);
    return 0;
}
