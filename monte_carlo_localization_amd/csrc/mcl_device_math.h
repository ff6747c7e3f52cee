// mcl_device_math.h — small deterministic math shared by the engine's kernels (gfx950 only).
//
// Everything here is written with explicit fma() so that, compiled with -ffp-contract=off, the
// rounding sequence is fixed: tests compare it bit-for-bit against the scalar restatement in
// oracle/mcl_oracle.c (orc_eng_*), which is test infrastructure and shares no code with this file.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mcl {

constexpr int kWeightFracBits = 36;                 // q = floor(w * 2^36), w in [0,1]
constexpr double kWeightScale = 68719476736.0;      // 2^36

// exp(x) for x <= 0; 0 below -745.  Cody-Waite reduction + degree-13 Horner, all fma.
__device__ __forceinline__ double det_exp(double x)
{
    if (!(x > -745.0)) return (x != x) ? x : 0.0;
    if (x > 0.0) x = 0.0;
    const double LOG2E = 1.4426950408889634074;
    const double LN2_HI = 6.93147180369123816490e-01;
    const double LN2_LO = 1.90821492927058770002e-10;
    double kf = rint(x * LOG2E);
    double r = __builtin_fma(-kf, LN2_HI, x);
    r = __builtin_fma(-kf, LN2_LO, r);
    double p = 1.0 / 6227020800.0;
    p = __builtin_fma(p, r, 1.0 / 479001600.0);
    p = __builtin_fma(p, r, 1.0 / 39916800.0);
    p = __builtin_fma(p, r, 1.0 / 3628800.0);
    p = __builtin_fma(p, r, 1.0 / 362880.0);
    p = __builtin_fma(p, r, 1.0 / 40320.0);
    p = __builtin_fma(p, r, 1.0 / 5040.0);
    p = __builtin_fma(p, r, 1.0 / 720.0);
    p = __builtin_fma(p, r, 1.0 / 120.0);
    p = __builtin_fma(p, r, 1.0 / 24.0);
    p = __builtin_fma(p, r, 1.0 / 6.0);
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    int k = (int)kf;
    if (k < -1000) { p = p * 0x1p-1000; k += 1000; }
    double s = __longlong_as_double((long long)(k + 1023) << 52);
    return p * s;
}

// Philox4x32-10 (Salmon, Moraes, Dror, Shaw 2011).
struct u32x4 { uint32_t v[4]; };
__device__ __forceinline__ u32x4 philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                            uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    u32x4 o; o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3;
    return o;
}
__device__ __forceinline__ uint64_t bits53(uint32_t a, uint32_t b) { return (((uint64_t)a << 32) | b) >> 11; }

// a*b > c*d for 64-bit unsigned operands, in exact 128-bit arithmetic.
__device__ __forceinline__ bool mul_gt(uint64_t a, uint64_t b, uint64_t c, uint64_t d)
{
    uint64_t lh = __umul64hi(a, b), ll = a * b;
    uint64_t rh = __umul64hi(c, d), rl = c * d;
    return (lh > rh) || (lh == rh && ll > rl);
}

// utils.cpp:43-48 with a bounded trip count (a non-finite or absurd angle must not hang a wave).
__device__ __forceinline__ double normalize_angle(double a)
{
    const double PI = 3.14159265358979323846;
    int it = 0;
    while (a > PI && it < 64) { a -= 2.0 * PI; ++it; }
    while (a < -PI && it < 128) { a += 2.0 * PI; ++it; }
    if (it >= 64 && (a > PI || a < -PI)) a = remainder(a, 2.0 * PI);
    return a;
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_max(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += (uint64_t)__shfl_xor((long long)v, o, 64);
    return v;
}

// Maximum of an int over the 64 lanes of the wave, wave-uniform, through the DPP cross-lane paths (quad permutes, row mirrors,
// row broadcasts: VALU only).  __shfl_xor compiles to ds_bpermute_b32 -- an LDS round trip and a wait per step -- which is what a
// reduction costs in a kernel whose waves are short of latency to hide (k_rays_sweep's per-chunk set-up made eighteen of them).
__device__ __forceinline__ int wave_max_i32(int v)
{
    int x = v;
    x = max(x, __builtin_amdgcn_update_dpp(x, x, 0xB1, 0xF, 0xF, false));     // quad_perm [1, 0, 3, 2]
    x = max(x, __builtin_amdgcn_update_dpp(x, x, 0x4E, 0xF, 0xF, false));     // quad_perm [2, 3, 0, 1]
    x = max(x, __builtin_amdgcn_update_dpp(x, x, 0x141, 0xF, 0xF, false));    // row_half_mirror
    x = max(x, __builtin_amdgcn_update_dpp(x, x, 0x140, 0xF, 0xF, false));    // row_mirror: every lane of a row holds the row's maximum
    x = max(x, __builtin_amdgcn_update_dpp(x, x, 0x142, 0xA, 0xF, false));    // row_bcast:15 into rows 1 and 3
    x = max(x, __builtin_amdgcn_update_dpp(x, x, 0x143, 0xC, 0xF, false));    // row_bcast:31 into rows 2 and 3: lane 63 holds the wave's
    return __builtin_amdgcn_readlane(x, 63);
}
__device__ __forceinline__ int wave_min_i32(int v) { return -wave_max_i32(-v); }      // (callers stay away from INT_MIN)
// Two maxima at once, the DPP operand folded into v_max_i32 (the builtin above costs a v_mov_dpp, the v_max and a copy per step):
// twelve VALU instructions for both; the two chains alternate, which leaves one of the two wait states a DPP read needs after the
// VALU write of its source to an s_nop.
__device__ __forceinline__ void wave_max2_i32(int &a, int &b)
{
    int x = a, y = b;
    asm volatile(
        "s_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_max_i32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_max_i32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_max_i32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_max_i32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_max_i32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_max_i32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_max_i32_dpp %1, %1, %1 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_max_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "v_max_i32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_max_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "v_max_i32_dpp %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(x), "+v"(y));
    a = __builtin_amdgcn_readlane(x, 63);
    b = __builtin_amdgcn_readlane(y, 63);
}

}  // namespace mcl
