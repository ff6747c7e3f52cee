// Micro-benchmark: how many cycles one wave64 VALU instruction occupies a SIMD of gfx950, by instruction kind and by
// waves per SIMD.  Settles whether the VALU issue limit of k_rays_cell is one instruction per 4 cycles (16 lanes per
// clock) or per 2 cycles (32 lanes per clock, as MI355X_MICROARCH.md's cycle table says for v_fma_f32).
//
// Method: every wave runs `iters` iterations of 32 asm-pinned instructions (8 independent chains x 4), all CUs busy,
// W waves on every SIMD (block = 256*min(W,4) threads, W = 8: two such blocks per CU).  Per wave the shader-clock
// cycles (s_memtime) and the 100 MHz reference ticks (s_memrealtime) around the loop are recorded, so the clock the chip
// actually held is measured, not assumed.  cycles per instruction per SIMD = median(d_memtime) / (W * iters * 32).
// HW_ID is recorded to check that the waves really were spread W per SIMD.
//
//   hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip && ./valu_rates
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <map>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

enum Kind { FMA_F32, PK_FMA_F32, MAD_I24, FMA_F64, ADD_F64, MUL_F64, ADD_U32, LSHL_ADD, MIN3_U32, SUB_CO, CVT_F64_F32, PROBE_VALU, PROBE_LDS, SWEEP_TRIP, SWEEP_BEAM, NKIND };
static const char *kind_name[NKIND] = {"v_fma_f32", "v_pk_fma_f32", "v_mad_i32_i24", "v_fma_f64", "v_add_f64", "v_mul_f64", "v_add_u32",
                                       "v_lshl_add_u32", "v_min3_u32", "v_sub_co_u32", "v_cvt_f64_f32",
                                       "k_rays_cell probe trip, 9 VALU (no LDS read)", "k_rays_cell probe trip, 9 VALU + ds_read_i8 + wait",
                                       "k_rays_sweep probe trip, 7 VALU (no LDS read)",
                                       "k_rays_sweep whole beam: 11 per-ray VALU + 5 trips (no memory)"};
// VALU instructions per "unit" of 32 asm statements (the probe bodies are 9 VALU, repeated 4x per iteration = 36)
static const int kind_insts[NKIND] = {32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 36, 36, 28, 46};

struct Stamp { unsigned long long cyc, ref; unsigned hwid, xcc, sink, pad; };

template <int KIND>
__global__ __launch_bounds__(1024) void k(Stamp *out, int iters, unsigned seed)
{
    __shared__ unsigned char lds[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = (unsigned char)(1 + (i & 3));
    unsigned a0 = seed + threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3, a4 = a0 ^ 0x1234, a5 = a0 + 77, a6 = a0 * 9, a7 = a0 + 5;
    float f0 = (float)a0, f1 = (float)a1, f2 = (float)a2, f3 = (float)a3, f4 = (float)a4, f5 = (float)a5, f6 = (float)a6, f7 = (float)a7;
    double d0 = (double)a0, d1 = (double)a1, d2 = (double)a2, d3 = (double)a3, d4 = (double)a4, d5 = (double)a5, d6 = (double)a6, d7 = (double)a7;
    typedef float f2v __attribute__((ext_vector_type(2)));
    f2v p0 = {f0, f1}, p1 = {f2, f3}, p2 = {f4, f5}, p3 = {f6, f7}, p4 = {f1, f0}, p5 = {f3, f2}, p6 = {f5, f4}, p7 = {f7, f6};
    const float cf = 1.0001f;
    const double cd = 1.0000001;
    const f2v cp = {1.0001f, 0.9999f};
    unsigned c1 = 77u, stride = 280u, gb = 109u << 10;
    asm volatile("" : "+v"(c1), "+v"(stride), "+v"(gb));
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int it = 0; it < iters; ++it) {
#define REP8(STMT) STMT(0) STMT(1) STMT(2) STMT(3) STMT(4) STMT(5) STMT(6) STMT(7)
#define REP32(STMT) REP8(STMT) REP8(STMT) REP8(STMT) REP8(STMT)
        if constexpr (KIND == FMA_F32) {
#define S(i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f##i) : "v"(cf));
            REP32(S)
#undef S
        } else if constexpr (KIND == PK_FMA_F32) {
#define S(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p##i) : "v"(cp));
            REP32(S)
#undef S
        } else if constexpr (KIND == MAD_I24) {
#define S(i) asm volatile("v_mad_i32_i24 %0, %0, %1, %0" : "+v"(a##i) : "v"(c1));
            REP32(S)
#undef S
        } else if constexpr (KIND == FMA_F64) {
#define S(i) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d##i) : "v"(cd));
            REP32(S)
#undef S
        } else if constexpr (KIND == ADD_F64) {
#define S(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d##i) : "v"(cd));
            REP32(S)
#undef S
        } else if constexpr (KIND == MUL_F64) {
#define S(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d##i) : "v"(cd));
            REP32(S)
#undef S
        } else if constexpr (KIND == ADD_U32) {
#define S(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a##i) : "v"(c1));
            REP32(S)
#undef S
        } else if constexpr (KIND == LSHL_ADD) {
#define S(i) asm volatile("v_lshl_add_u32 %0, %0, 10, %1" : "+v"(a##i) : "v"(gb));
            REP32(S)
#undef S
        } else if constexpr (KIND == MIN3_U32) {
#define S(i) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(a##i) : "v"(c1), "v"(gb));
            REP32(S)
#undef S
        } else if constexpr (KIND == SUB_CO) {
#define S(i) asm volatile("v_sub_co_u32 %0, vcc, %0, %1" : "+v"(a##i) : "v"(c1) : "vcc");
            REP32(S)
#undef S
        } else if constexpr (KIND == CVT_F64_F32) {
#define S(i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d##i) : "v"(f##i));
            REP32(S)
#undef S
        } else if constexpr (KIND == SWEEP_TRIP || KIND == SWEEP_BEAM) {
            // k_rays_sweep (mcl_rays_sweep.h): MCL_SW_TRIP without the LDS read, and the whole per-beam instruction stream
            // (direction rotation + rounding, next-beam bookkeeping, five trips, guard test, table offset, accumulate)
#define SW_TRIP                                                \
    "v_mad_u32_u24 %[tx], %[ad], %[nux], %[tx]\n\t"            \
    "v_mad_u32_u24 %[ty], %[ad], %[nuy], %[ty]\n\t"            \
    "v_perm_b32 %[ad], %[ty], %[tx], %[str]\n\t"               \
    "v_and_b32 %[t0], %[gb], %[tx]\n\t"                         \
    "v_and_b32 %[t1], %[gb], %[ty]\n\t"                         \
    "v_min3_u32 %[g], %[g], %[t0], %[t1]\n\t"                  \
    "v_sub_co_u32 %[rem], vcc, %[rem], %[ad]\n\t"
            unsigned tx = a4, ty = a5, t0v, t1v, ad = a6;
            if constexpr (KIND == SWEEP_TRIP) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    asm volatile(SW_TRIP
                                 : [tx] "+v"(tx), [ty] "+v"(ty), [t0] "=&v"(t0v), [t1] "=&v"(t1v), [ad] "+v"(ad), [g] "+v"(a1), [rem] "+v"(a0)
                                 : [nux] "v"(a2), [nuy] "v"(a3), [str] "v"(stride), [gb] "v"(gb)
                                 : "vcc");
                a4 = tx; a5 = ty; a6 = ad;
            } else {
                asm volatile(
                    "v_mul_f64 %[p], %[sd], %[dy]\n\t"
                    "v_fma_f64 %[ax], %[cd], %[dx], -%[p]\n\t"
                    "v_mul_f64 %[p], %[cd], %[dy]\n\t"
                    "v_fma_f64 %[ay], %[sd], %[dx], %[p]\n\t"
                    "v_add_f64 %[ax], %[ax], %[mg]\n\t"
                    "v_add_f64 %[ay], %[ay], %[mg]\n\t"
                    "v_add_u32 %[j16], %[j16], %[str]\n\t"
                    SW_TRIP SW_TRIP SW_TRIP SW_TRIP SW_TRIP
                    "v_cmp_gt_u32 vcc, %[gb], %[g]\n\t"
                    "v_add_f64 %[acc], %[acc], %[dx]\n\t"
                    "v_mad_i32_i24 %[ad], %[rem], %[str], %[j8]\n\t"
                    "v_add_u32 %[j8], %[j8], %[str]\n\t"
                    : [tx] "+v"(tx), [ty] "+v"(ty), [t0] "=&v"(t0v), [t1] "=&v"(t1v), [ad] "+v"(ad), [g] "+v"(a1), [rem] "+v"(a0),
                      [p] "=&v"(d0), [ax] "=&v"(d1), [ay] "=&v"(d2), [acc] "+v"(d3), [j16] "+v"(a6), [j8] "+v"(a7)
                    : [nux] "v"(a2), [nuy] "v"(a3), [str] "v"(stride), [gb] "v"(gb), [sd] "v"(d4), [cd] "v"(d5), [dx] "v"(d6), [dy] "v"(d7), [mg] "v"(cd)
                    : "vcc");
                a4 = tx; a5 = ty;
            }
        } else {
            // the probe loop body of k_rays_cell (mcl_kernels.h): rem in a0, g in a1; 4 trips per iteration
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                unsigned tx, ty, t0v, t1v, ad, by;
                if constexpr (KIND == PROBE_VALU) {
                    asm volatile(
                        "v_mad_i32_i24 %[tx], %[rem], %[nux], %[pex]\n\t"
                        "v_mad_i32_i24 %[ty], %[rem], %[nuy], %[pey]\n\t"
                        "v_lshrrev_b32 %[t0], 22, %[tx]\n\t"
                        "v_lshrrev_b32 %[t1], 22, %[ty]\n\t"
                        "v_mad_u32_u24 %[ad], %[t1], %[str], %[t0]\n\t"
                        "v_lshl_add_u32 %[t0], %[tx], 10, %[gb]\n\t"
                        "v_lshl_add_u32 %[t1], %[ty], 10, %[gb]\n\t"
                        "v_min3_u32 %[g], %[g], %[t0], %[t1]\n\t"
                        "v_sub_co_u32 %[rem], vcc, %[rem], %[ad]\n\t"
                        : [tx] "=&v"(tx), [ty] "=&v"(ty), [t0] "=&v"(t0v), [t1] "=&v"(t1v), [ad] "=&v"(ad), [g] "+v"(a1), [rem] "+v"(a0)
                        : [nux] "v"(a2), [nuy] "v"(a3), [pex] "v"(a4), [pey] "v"(a5), [str] "v"(stride), [gb] "v"(gb)
                        : "vcc");
                } else {
                    asm volatile(
                        "v_mad_i32_i24 %[tx], %[rem], %[nux], %[pex]\n\t"
                        "v_mad_i32_i24 %[ty], %[rem], %[nuy], %[pey]\n\t"
                        "v_lshrrev_b32 %[t0], 29, %[tx]\n\t"
                        "v_lshrrev_b32 %[t1], 29, %[ty]\n\t"
                        "v_mad_u32_u24 %[ad], %[t1], %[str], %[t0]\n\t"
                        "ds_read_i8 %[by], %[ad]\n\t"
                        "v_lshl_add_u32 %[t0], %[tx], 10, %[gb]\n\t"
                        "v_lshl_add_u32 %[t1], %[ty], 10, %[gb]\n\t"
                        "v_min3_u32 %[g], %[g], %[t0], %[t1]\n\t"
                        "s_waitcnt lgkmcnt(0)\n\t"
                        "v_sub_co_u32 %[rem], vcc, %[rem], %[by]\n\t"
                        : [tx] "=&v"(tx), [ty] "=&v"(ty), [t0] "=&v"(t0v), [t1] "=&v"(t1v), [ad] "=&v"(ad), [by] "=&v"(by), [g] "+v"(a1), [rem] "+v"(a0)
                        : [nux] "v"(a2), [nuy] "v"(a3), [pex] "v"(a4), [pey] "v"(a5), [str] "v"(stride), [gb] "v"(gb)
                        : "vcc", "memory");
                }
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    unsigned r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (unsigned)(f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7) +
                 (unsigned)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7) + (unsigned)(p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y);
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if ((threadIdx.x & 63) == 0) {
        Stamp s;
        s.cyc = t1 - t0; s.ref = r1 - r0;
        s.hwid = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID, 32 bits
        s.xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 15;    // HW_REG_XCC_ID
        s.sink = r; s.pad = 0;
        out[wave] = s;
    }
}

template <int KIND>
int run(Stamp *d_out, int W, int iters)
{
    const int tpb = 256 * std::min(W, 4), blocks = 256 * std::max(1, W / 4);
    const int nwaves = blocks * tpb / 64;
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(tpb), 0, 0, d_out, iters / 8, 1u);     // warm-up
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(tpb), 0, 0, d_out, iters, 2u);
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<Stamp> s(nwaves);
    CHK(hipMemcpy(s.data(), d_out, nwaves * sizeof(Stamp), hipMemcpyDeviceToHost));
    std::vector<double> cyc(nwaves), ghz(nwaves);
    std::map<unsigned, int> per_simd;      // (se, cu, simd) -> waves
    for (int i = 0; i < nwaves; ++i) {
        cyc[i] = (double)s[i].cyc;
        ghz[i] = (double)s[i].cyc / ((double)s[i].ref * 10.0);      // ref ticks are 10 ns
    }
    // HW_ID (gfx9): wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13]; pad = XCC_ID
    for (int i = 0; i < nwaves; ++i) per_simd[(s[i].xcc << 16) | ((s[i].hwid >> 4) & 0xFFF & ~0xCu)] += 1;   // drop pipe_id
    int wmin = 1 << 30, wmax = 0;
    for (auto &kv : per_simd) { wmin = std::min(wmin, kv.second); wmax = std::max(wmax, kv.second); }
    std::sort(cyc.begin(), cyc.end()); std::sort(ghz.begin(), ghz.end());
    const double med_cyc = cyc[nwaves / 2], med_ghz = ghz[nwaves / 2];
    const double insts = (double)W * iters * kind_insts[KIND];
    printf("%-46s W=%d  wall %8.3f ms  clock %.3f GHz  %6.2f cycles/VALU inst/SIMD (in-kernel)  %6.2f (wall x clock)  waves per SIMD seen %d..%d\n",
           kind_name[KIND], W, ms, med_ghz, med_cyc / insts, ms * 1e-3 * med_ghz * 1e9 / insts, wmin, wmax);
    return 0;
}

template <int KIND>
int sweep(Stamp *d_out, int iters)
{
    for (int W : {1, 2, 4, 8})
        if (run<KIND>(d_out, W, iters / W)) return 1;
    return 0;
}

int main()
{
    Stamp *d_out;
    CHK(hipMalloc(&d_out, 512 * 16 * sizeof(Stamp)));
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    printf("# %s, %d CUs, clockRate %d kHz; 32 asm-pinned instructions per iteration, 8 independent chains\n", prop.gcnArchName,
           prop.multiProcessorCount, prop.clockRate);
    const int iters = 1 << 17;
    if (sweep<FMA_F32>(d_out, iters) || sweep<PK_FMA_F32>(d_out, iters) || sweep<MAD_I24>(d_out, iters) || sweep<FMA_F64>(d_out, iters) ||
        sweep<ADD_F64>(d_out, iters) || sweep<MUL_F64>(d_out, iters) || sweep<ADD_U32>(d_out, iters) || sweep<LSHL_ADD>(d_out, iters) ||
        sweep<MIN3_U32>(d_out, iters) || sweep<SUB_CO>(d_out, iters) || sweep<CVT_F64_F32>(d_out, iters) || sweep<PROBE_VALU>(d_out, iters) ||
        sweep<PROBE_LDS>(d_out, iters / 4) || sweep<SWEEP_TRIP>(d_out, iters) || sweep<SWEEP_BEAM>(d_out, iters / 2))
        return 1;
    return 0;
}
