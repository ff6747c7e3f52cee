// Micro-benchmark: SIMD cycles per probe trip of k_rays_sweep's walk IN ITS LOOP STRUCTURE (exec masking, s_cbranch_execz, the
// countdown's taken branch, the LDS read and its wait), by trip variant and waves per SIMD.  tools/ubench/valu_rates.hip prices
// straight-line instruction streams; this one answers what the walk is bound by once the VALU count per trip drops: issue
// cycles, or the dependent chain  mad -> address -> ds_read -> wait -> sub_co -> exec -> branch  of one wave (8 waves per SIMD
// cover a chain of at most 8 x the issue cycles of a trip).
//
// Every lane runs `rays` rays of exactly K trips (the LDS window holds skip 1 everywhere, samples left = K - 1).
// cycles per trip per SIMD = median over waves of (s_memtime span) / (W * rays * K).
//
//   hipcc --offload-arch=gfx950 -O3 -o trip_rates trip_rates.hip && ./trip_rates
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

enum Kind { OLD7, NEW5, NEW5_UNROLL, NEW5_NOLDS, NEW5_CMPX, NEW5_X2, NEW5_UNROLL_X2, OLD7_UNROLL, NKIND };
static const char *kind_name[NKIND] = {
    "round-4 trip: 7 VALU (mad24 x2, perm, and x2, min3, sub_co), loop with countdown",
    "64-bit trip: 5 VALU (mad_u64 x2, lshl_or, min3, sub_co), loop with countdown",
    "64-bit trip, 8 trips unrolled (not-taken execz branches only)",
    "64-bit trip, loop with countdown, NO LDS read (byte = constant)",
    "64-bit trip, v_sub + v_cmpx (exec written by the VALU), loop with countdown",
    "64-bit trip, TWO rays per lane interleaved, loop with countdown",
    "64-bit trip, TWO rays per lane interleaved, 8 trips unrolled",
    "round-4 trip, 8 trips unrolled",
};

struct Stamp { unsigned long long cyc, ref; unsigned sink, pad; };

#define OLD_TRIP(REM, BYIN, TXIN, TYIN)                      \
    "v_mad_u32_u24 v54, " BYIN ", %[xx], " TXIN "\n\t"       \
    "v_mad_u32_u24 v55, " BYIN ", %[xy], " TYIN "\n\t"       \
    "v_perm_b32 v48, v55, v54, %[sel]\n\t"                   \
    "ds_read_i8 v49, v48\n\t"                                \
    "v_and_b32 v45, %[mask], v54\n\t"                        \
    "v_and_b32 v47, %[mask], v55\n\t"                        \
    "v_min3_u32 %[g], %[g], v45, v47\n\t"                    \
    "s_waitcnt lgkmcnt(0)\n\t"                               \
    "v_sub_co_u32 v57, vcc, " REM ", v49\n\t"                \
    "s_andn2_b64 exec, exec, vcc\n\t"

#define NEW_TRIP(REM, BYIN, TXIN, TYIN)                      \
    "v_mad_u64_u32 v[52:53], vcc, " BYIN ", %[xx], " TXIN "\n\t" \
    "v_mad_u64_u32 v[54:55], vcc, " BYIN ", %[xy], " TYIN "\n\t" \
    "v_lshl_or_b32 v48, v55, 8, v53\n\t"                     \
    "ds_read_i8 v49, v48\n\t"                                \
    "v_min3_u32 %[g], %[g], v52, v54\n\t"                    \
    "s_waitcnt lgkmcnt(0)\n\t"                               \
    "v_sub_co_u32 v57, vcc, " REM ", v49\n\t"                \
    "s_andn2_b64 exec, exec, vcc\n\t"

#define NEW_TRIP_NOLDS(REM, BYIN, TXIN, TYIN)                \
    "v_mad_u64_u32 v[52:53], vcc, " BYIN ", %[xx], " TXIN "\n\t" \
    "v_mad_u64_u32 v[54:55], vcc, " BYIN ", %[xy], " TYIN "\n\t" \
    "v_lshl_or_b32 v48, v55, 8, v53\n\t"                     \
    "v_mov_b32 v49, 1\n\t"                                   \
    "v_min3_u32 %[g], %[g], v52, v54\n\t"                    \
    "v_sub_co_u32 v57, vcc, " REM ", v49\n\t"                \
    "s_andn2_b64 exec, exec, vcc\n\t"

#define NEW_TRIP_CMPX(REM, BYIN, TXIN, TYIN)                 \
    "v_mad_u64_u32 v[52:53], vcc, " BYIN ", %[xx], " TXIN "\n\t" \
    "v_mad_u64_u32 v[54:55], vcc, " BYIN ", %[xy], " TYIN "\n\t" \
    "v_lshl_or_b32 v48, v55, 8, v53\n\t"                     \
    "ds_read_i8 v49, v48\n\t"                                \
    "v_min3_u32 %[g], %[g], v52, v54\n\t"                    \
    "s_waitcnt lgkmcnt(0)\n\t"                               \
    "v_cmpx_ge_u32 vcc, " REM ", v49\n\t"                   \
    "v_sub_u32 v57, " REM ", v49\n\t"

// two rays per lane: ray B in v[58:59] v[60:61] (T), v62 address, v63 byte, v56 samples left; exit masks in s[20:21] / s[22:23]
#define NEW_TRIP_X2(REMA, REMB, BYA, BYB, TXA, TYA, TXB, TYB)                \
    "v_mad_u64_u32 v[52:53], vcc, " BYA ", %[xx], " TXA "\n\t"               \
    "v_mad_u64_u32 v[54:55], vcc, " BYA ", %[xy], " TYA "\n\t"               \
    "v_mad_u64_u32 v[58:59], vcc, " BYB ", %[xx2], " TXB "\n\t"              \
    "v_mad_u64_u32 v[60:61], vcc, " BYB ", %[xy2], " TYB "\n\t"              \
    "v_lshl_or_b32 v48, v55, 8, v53\n\t"                                     \
    "ds_read_i8 v49, v48\n\t"                                                \
    "v_lshl_or_b32 v62, v61, 8, v59\n\t"                                     \
    "ds_read_i8 v63, v62\n\t"                                                \
    "v_min3_u32 %[g], %[g], v52, v54\n\t"                                    \
    "v_min3_u32 %[g], %[g], v58, v60\n\t"                                    \
    "s_waitcnt lgkmcnt(1)\n\t"                                               \
    "v_sub_co_u32 v57, vcc, " REMA ", v49\n\t"                               \
    "s_waitcnt lgkmcnt(0)\n\t"                                               \
    "v_sub_co_u32 v56, s[20:21], " REMB ", v63\n\t"                          \
    "s_and_b64 vcc, vcc, s[20:21]\n\t"                                       \
    "s_andn2_b64 exec, exec, vcc\n\t"

template <int KIND>
__global__ __launch_bounds__(1024) void k(Stamp *out, int rays, int K, unsigned seed)
{
    __shared__ unsigned char lds[65536];
    for (int i = threadIdx.x; i < 65536; i += blockDim.x) lds[i] = 1;
    // positions: cell (3 + lane % 32, 5 + wave) of the window, small direction components so that K trips stay in a few cells
    const unsigned lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned xx = 0x30000000u + seed + lane * 977u, xy = 0x20000000u + lane * 131u;      // 64-bit form: 32 fractional bits
    unsigned xx24 = xx >> 8, xy24 = xy >> 8;                                                 // 24-bit form
    unsigned long long p0x = ((unsigned long long)(3 + (lane & 31)) << 32) | 0x12345678u, p0y = ((unsigned long long)(5 + wv) << 32) | 0x9abcdef0u;
    unsigned p0x24 = (unsigned)(p0x >> 8), p0y24 = (unsigned)(p0y >> 8);
    unsigned rem0 = (unsigned)K - 1u, s0 = 1u, g = 0xFFFFFFFFu, mask = 0xFFFFFFu, sel = 0x0C0C0703u;
    unsigned cd;
    asm volatile("" : "+v"(xx), "+v"(xy), "+v"(xx24), "+v"(xy24), "+v"(p0x), "+v"(p0y), "+v"(p0x24), "+v"(p0y24), "+v"(rem0), "+v"(s0));
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int it = 0; it < rays; ++it) {
        if constexpr (KIND == OLD7) {
            asm volatile(
                "s_movk_i32 %[cd], 300\n\t"
                OLD_TRIP("%[rem0]", "%[s0]", "%[p0x]", "%[p0y]")
                "s_cbranch_execz 2f\n"
                "1:\n\t"
                OLD_TRIP("v57", "v49", "v54", "v55")
                "s_cbranch_execz 2f\n\t"
                "s_sub_u32 %[cd], %[cd], 1\n\t"
                "s_cbranch_scc0 1b\n"
                "2:\n\t"
                "s_mov_b64 exec, -1\n\t"
                : [g] "+v"(g), [cd] "=&s"(cd)
                : [xx] "v"(xx24), [xy] "v"(xy24), [p0x] "v"(p0x24), [p0y] "v"(p0y24), [rem0] "v"(rem0), [s0] "v"(s0), [mask] "s"(mask), [sel] "s"(sel)
                : "memory", "vcc", "scc", "v45", "v47", "v48", "v49", "v54", "v55", "v57");
        } else if constexpr (KIND == OLD7_UNROLL) {
            asm volatile(
                OLD_TRIP("%[rem0]", "%[s0]", "%[p0x]", "%[p0y]")
                "s_cbranch_execz 2f\n"
                "1:\n\t"
                OLD_TRIP("v57", "v49", "v54", "v55") "s_cbranch_execz 2f\n\t"
                OLD_TRIP("v57", "v49", "v54", "v55") "s_cbranch_execz 2f\n\t"
                OLD_TRIP("v57", "v49", "v54", "v55") "s_cbranch_execz 2f\n\t"
                OLD_TRIP("v57", "v49", "v54", "v55") "s_cbranch_execz 2f\n\t"
                OLD_TRIP("v57", "v49", "v54", "v55") "s_cbranch_execz 2f\n\t"
                OLD_TRIP("v57", "v49", "v54", "v55") "s_cbranch_execz 2f\n\t"
                OLD_TRIP("v57", "v49", "v54", "v55") "s_cbranch_execz 2f\n\t"
                OLD_TRIP("v57", "v49", "v54", "v55") "s_cbranch_execz 2f\n\t"
                "s_branch 1b\n"
                "2:\n\t"
                "s_mov_b64 exec, -1\n\t"
                : [g] "+v"(g)
                : [xx] "v"(xx24), [xy] "v"(xy24), [p0x] "v"(p0x24), [p0y] "v"(p0y24), [rem0] "v"(rem0), [s0] "v"(s0), [mask] "s"(mask), [sel] "s"(sel)
                : "memory", "vcc", "scc", "v45", "v47", "v48", "v49", "v54", "v55", "v57");
        } else if constexpr (KIND == NEW5 || KIND == NEW5_NOLDS || KIND == NEW5_CMPX) {
#define LOOPED(TRIP)                                                                                                  \
            asm volatile(                                                                                             \
                "s_movk_i32 %[cd], 300\n\t"                                                                           \
                TRIP("%[rem0]", "%[s0]", "%[p0x]", "%[p0y]")                                                          \
                "s_cbranch_execz 2f\n"                                                                                \
                "1:\n\t"                                                                                              \
                TRIP("v57", "v49", "v[52:53]", "v[54:55]")                                                            \
                "s_cbranch_execz 2f\n\t"                                                                              \
                "s_sub_u32 %[cd], %[cd], 1\n\t"                                                                       \
                "s_cbranch_scc0 1b\n"                                                                                 \
                "2:\n\t"                                                                                              \
                "s_mov_b64 exec, -1\n\t"                                                                              \
                : [g] "+v"(g), [cd] "=&s"(cd)                                                                         \
                : [xx] "v"(xx), [xy] "v"(xy), [p0x] "v"(p0x), [p0y] "v"(p0y), [rem0] "v"(rem0), [s0] "v"(s0)              \
                : "memory", "vcc", "scc", "v48", "v49", "v52", "v53", "v54", "v55", "v57");
            if constexpr (KIND == NEW5) { LOOPED(NEW_TRIP) }
            else if constexpr (KIND == NEW5_NOLDS) { LOOPED(NEW_TRIP_NOLDS) }
            else { LOOPED(NEW_TRIP_CMPX) }
        } else if constexpr (KIND == NEW5_UNROLL) {
            asm volatile(
                NEW_TRIP("%[rem0]", "%[s0]", "%[p0x]", "%[p0y]")
                "s_cbranch_execz 2f\n"
                "1:\n\t"
                NEW_TRIP("v57", "v49", "v[52:53]", "v[54:55]") "s_cbranch_execz 2f\n\t"
                NEW_TRIP("v57", "v49", "v[52:53]", "v[54:55]") "s_cbranch_execz 2f\n\t"
                NEW_TRIP("v57", "v49", "v[52:53]", "v[54:55]") "s_cbranch_execz 2f\n\t"
                NEW_TRIP("v57", "v49", "v[52:53]", "v[54:55]") "s_cbranch_execz 2f\n\t"
                NEW_TRIP("v57", "v49", "v[52:53]", "v[54:55]") "s_cbranch_execz 2f\n\t"
                NEW_TRIP("v57", "v49", "v[52:53]", "v[54:55]") "s_cbranch_execz 2f\n\t"
                NEW_TRIP("v57", "v49", "v[52:53]", "v[54:55]") "s_cbranch_execz 2f\n\t"
                NEW_TRIP("v57", "v49", "v[52:53]", "v[54:55]") "s_cbranch_execz 2f\n\t"
                "s_branch 1b\n"
                "2:\n\t"
                "s_mov_b64 exec, -1\n\t"
                : [g] "+v"(g)
                : [xx] "v"(xx), [xy] "v"(xy), [p0x] "v"(p0x), [p0y] "v"(p0y), [rem0] "v"(rem0), [s0] "v"(s0)
                : "memory", "vcc", "scc", "v48", "v49", "v52", "v53", "v54", "v55", "v57");
        } else if constexpr (KIND == NEW5_X2) {
            unsigned xx2 = xx + 12345u, xy2 = xy + 777u;
            asm volatile(
                "s_movk_i32 %[cd], 300\n\t"
                NEW_TRIP_X2("%[rem0]", "%[rem0]", "%[s0]", "%[s0]", "%[p0x]", "%[p0y]", "%[p0x]", "%[p0y]")
                "s_cbranch_execz 2f\n"
                "1:\n\t"
                NEW_TRIP_X2("v57", "v56", "v49", "v63", "v[52:53]", "v[54:55]", "v[58:59]", "v[60:61]")
                "s_cbranch_execz 2f\n\t"
                "s_sub_u32 %[cd], %[cd], 1\n\t"
                "s_cbranch_scc0 1b\n"
                "2:\n\t"
                "s_mov_b64 exec, -1\n\t"
                : [g] "+v"(g), [cd] "=&s"(cd)
                : [xx] "v"(xx), [xy] "v"(xy), [xx2] "v"(xx2), [xy2] "v"(xy2), [p0x] "v"(p0x), [p0y] "v"(p0y), [rem0] "v"(rem0), [s0] "v"(s0)
                : "memory", "vcc", "scc", "s20", "s21", "v48", "v49", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63");
        } else if constexpr (KIND == NEW5_UNROLL_X2) {
            unsigned xx2 = xx + 12345u, xy2 = xy + 777u;
#define X2L NEW_TRIP_X2("v57", "v56", "v49", "v63", "v[52:53]", "v[54:55]", "v[58:59]", "v[60:61]") "s_cbranch_execz 2f\n\t"
            asm volatile(
                NEW_TRIP_X2("%[rem0]", "%[rem0]", "%[s0]", "%[s0]", "%[p0x]", "%[p0y]", "%[p0x]", "%[p0y]")
                "s_cbranch_execz 2f\n"
                "1:\n\t"
                X2L X2L X2L X2L X2L X2L X2L X2L
                "s_branch 1b\n"
                "2:\n\t"
                "s_mov_b64 exec, -1\n\t"
                : [g] "+v"(g)
                : [xx] "v"(xx), [xy] "v"(xy), [xx2] "v"(xx2), [xy2] "v"(xy2), [p0x] "v"(p0x), [p0y] "v"(p0y), [rem0] "v"(rem0), [s0] "v"(s0)
                : "memory", "vcc", "scc", "s20", "s21", "v48", "v49", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63");
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) {
        Stamp s;
        s.cyc = t1 - t0; s.ref = r1 - r0; s.sink = g; s.pad = 0;
        out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = s;
    }
}

template <int KIND>
static int run(int W, int K, int cus)
{
    // W waves per SIMD: block = 1024 threads (4 waves per SIMD), 64 KB of LDS each: W = 4 -> one block per CU, W = 8 -> two
    const int blocks = cus * (W / 4), waves = blocks * 16, rays = 4000;
    Stamp *d;
    CHK(hipMalloc(&d, sizeof(Stamp) * waves));
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(1024), 0, 0, d, 200, K, 1u);      // warm-up
    CHK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(1024), 0, 0, d, rays, K, 1u);
    CHK(hipDeviceSynchronize());
    std::vector<Stamp> h(waves);
    CHK(hipMemcpy(h.data(), d, sizeof(Stamp) * waves, hipMemcpyDeviceToHost));
    std::vector<double> cyc, clk;
    for (auto &s : h) { cyc.push_back((double)s.cyc); clk.push_back((double)s.cyc / ((double)s.ref * 10.0)); }     // ref ticks are 10 ns
    std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
    const int per_trip_rays = (KIND == NEW5_X2 || KIND == NEW5_UNROLL_X2) ? 2 : 1;
    const double c = cyc[cyc.size() / 2] / ((double)W * rays * K * per_trip_rays);
    printf("%-82s W=%d K=%d  %6.2f SIMD cycles per ray-trip  (one wave: %6.1f cycles per trip)  clock %.3f GHz\n", kind_name[KIND], W, K, c,
           cyc[cyc.size() / 2] / ((double)rays * K), clk[clk.size() / 2]);
    CHK(hipFree(d));
    return 0;
}

int main()
{
    hipDeviceProp_t p;
    CHK(hipGetDeviceProperties(&p, 0));
    printf("# %s, %d CUs\n", p.gcnArchName, p.multiProcessorCount);
    const int cus = p.multiProcessorCount;
    for (int W : {4, 8})
        for (int K : {4, 6}) {
            if (run<OLD7>(W, K, cus)) return 1;
            if (run<OLD7_UNROLL>(W, K, cus)) return 1;
            if (run<NEW5>(W, K, cus)) return 1;
            if (run<NEW5_UNROLL>(W, K, cus)) return 1;
            if (run<NEW5_NOLDS>(W, K, cus)) return 1;
            if (run<NEW5_CMPX>(W, K, cus)) return 1;
            if (run<NEW5_X2>(W, K, cus)) return 1;
            if (run<NEW5_UNROLL_X2>(W, K, cus)) return 1;
        }
    return 0;
}
