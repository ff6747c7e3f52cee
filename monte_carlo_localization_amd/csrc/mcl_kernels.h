// mcl_kernels.h — the HIP kernels of the particle-filter update (gfx950 / CDNA4, wave64).
//
//   K5  k_scan_*            exact uint64 inclusive scan of fixed-point weights (resample CDF)
//   K6+K1+K2 k_resample_motion   threshold -> CDF search -> parent gather -> motion + noise
//   K3  k_rays<MODE>        fused ray cast + beam-model log-likelihood  (the dominant kernel)
//   K4  k_reduce_max / k_weights / k_final_*   max-subtracted weights, fixed-point weights, sums
//   K7  (fused into k_weights) weighted pose sums
//
// Reference rows: R = cpp:656-665, M = cpp:449-503, Q/C/E = cpp:506-650,
// N = cpp:678-686, P = cpp:696-716.
#pragma once
#include "mcl_types.h"
#include <type_traits>
#include "mcl_device_math.h"
#include "mcl_wedge.h"

namespace mcl {

// ------------------------------------------------------------------------------------------------
// K5: inclusive scan of uint64.  2048 items per 256-thread block (8 per thread), three launches:
// per-block totals -> single-block exclusive scan of the totals -> block-local scan + offset.
// Integer addition is associative, so the result does not depend on N, the block size or the
// number of GPUs the particle set is sharded over.
// ------------------------------------------------------------------------------------------------
constexpr int kRedBlocks = 1024;              // K4 / K7 reductions (below): fixed grid, fixed-order final pass
constexpr int kRedThreads = 256;
__device__ __forceinline__ void final_sums_of(const double *__restrict__ part, int nb, double *__restrict__ scalars, double (*sm)[7]);


__device__ __forceinline__ uint64_t wave_incl_scan_u64(uint64_t v, int lane)
{
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint64_t t = (uint64_t)__shfl_up((long long)v, o, 64);
        if (lane >= o) v += t;
    }
    return v;
}

// Order-preserving list of the particles with a non-zero fixed-point weight ("alive"), produced by the CDF scan itself.
// After an update with a thousand beams a few per cent of the particles carry all the weight (3.5 % at 4M x 1081 on
// Spielberg_map): the next update's resampling searches and gathers from this list -- 1 MB of CDF and 5 MB of records,
// cache-resident -- instead of the 33 MB CDF and the 134 MB of parent records, and a sharded set exchanges the lists
// instead of every weight.  A particle with q = 0 can never be selected (E6: the first i with
// C_i * lmul > rhs has q_i > 0), so the draw from the list equals the draw from the full CDF.
struct CompactOut {
    uint32_t *block_cnt;        // [tiles]: alive per scan tile -> exclusive prefix (k_scan_spine); null = no list wanted
    uint64_t *ccdf;             // CDF value at every alive particle (strictly increasing)
    uint32_t *cidx;             // its index
    double4 *crec;              // its record (x, y, theta, -)
    uint64_t *ctop;             // ccdf[64 g + 63] for every complete group of 64 entries: the search's first level
    const double *x, *y, *th;
    uint32_t cap;               // room in the arrays above
    unsigned long long *total;  // alive particles of this scan (may exceed cap: the list is then unusable)
};
constexpr int kCompactGroupShift = 6;

__global__ __launch_bounds__(kScanThreads) void k_scan_partials(const uint64_t *__restrict__ q, int64_t n,
                                                               uint64_t *__restrict__ block_tot, uint32_t *__restrict__ block_cnt)
{
    __shared__ uint64_t sm[kScanThreads / 64];
    __shared__ uint32_t smc[kScanThreads / 64];
    int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
    uint64_t s = 0;
    uint32_t c = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k)
        if (base + k < n) { const uint64_t v = q[base + k]; s += v; c += v != 0ull; }
    s = wave_sum_u64(s);
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (block_cnt) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    }
    if (lane == 0) { sm[w] = s; smc[w] = c; }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t t = 0;
        uint32_t tc = 0;
        for (int i = 0; i < kScanThreads / 64; ++i) { t += sm[i]; tc += smc[i]; }
        block_tot[blockIdx.x] = t;
        if (block_cnt) block_cnt[blockIdx.x] = tc;
    }
}

// exclusive scan of nb block totals in place (single block of 1024 threads), adds `offset`; the alive counts likewise.
// sum_part (optional): the per-workgroup partial sums k_weights has just left -> scalars[1..7], what a k_final_sums launch in
// between would do (this kernel is one workgroup anyway and runs after k_weights).
__global__ __launch_bounds__(1024) void k_scan_spine(uint64_t *__restrict__ block_tot, int nb, uint64_t offset,
                                                     uint64_t *__restrict__ grand_total, uint32_t *__restrict__ block_cnt,
                                                     unsigned long long *__restrict__ alive_total,
                                                     const double *__restrict__ sum_part = nullptr, int n_sum_part = 0, double *__restrict__ scalars = nullptr)
{
    __shared__ uint64_t sm[16];
    if (sum_part) {
        __shared__ double sums_sm[kRedThreads / 64][7];
        final_sums_of(sum_part, n_sum_part, scalars, sums_sm);
    }
    __shared__ uint32_t smc[16];
    __shared__ uint64_t carry_s;
    __shared__ uint32_t carry_c;
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (threadIdx.x == 0) { carry_s = offset; carry_c = 0u; }
    __syncthreads();
    for (int base = 0; base < nb; base += 1024) {
        int i = base + threadIdx.x;
        uint64_t v = (i < nb) ? block_tot[i] : 0;
        uint32_t vc = (block_cnt && i < nb) ? block_cnt[i] : 0u;
        uint64_t inc = wave_incl_scan_u64(v, lane);
        uint32_t incc = vc;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(incc, o, 64); if (lane >= o) incc += t; }
        if (lane == 63) { sm[w] = inc; smc[w] = incc; }
        __syncthreads();
        uint64_t woff = 0;
        uint32_t woffc = 0;
        for (int k = 0; k < w; ++k) { woff += sm[k]; woffc += smc[k]; }
        uint64_t carry = carry_s;
        uint32_t carryc = carry_c;
        if (i < nb) {
            block_tot[i] = carry + woff + inc - v;
            if (block_cnt) block_cnt[i] = carryc + woffc + incc - vc;
        }
        __syncthreads();
        if (threadIdx.x == 1023) { carry_s = carry + woff + inc; carry_c = carryc + woffc + incc; }
        __syncthreads();
    }
    if (threadIdx.x == 0 && grand_total) *grand_total = carry_s;
    if (threadIdx.x == 0 && alive_total) *alive_total = (unsigned long long)carry_c;
}

// `leaders` (optional) receives the last CDF entry of every 16-entry (128-byte) group: a 16x smaller copy the
// resampling search walks instead of the CDF itself, so that it fetches one CDF line per child, not seven
constexpr int kLeaderShift = 4;
__global__ __launch_bounds__(kScanThreads) void k_scan_final(const uint64_t *__restrict__ q, int64_t n,
                                                            const uint64_t *__restrict__ block_off,
                                                            uint64_t *__restrict__ cdf, uint64_t *__restrict__ leaders, CompactOut co)
{
    __shared__ uint64_t sm[kScanThreads / 64];
    __shared__ uint32_t smc[kScanThreads / 64];
    int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
    uint64_t v[kScanItems];
    uint64_t s = 0;
    uint32_t alive = 0;                      // bit k: item k has a non-zero weight
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        uint64_t x = (base + k < n) ? q[base + k] : 0;
        s += x;
        v[k] = s;
        alive |= (x != 0ull ? 1u : 0u) << k;
    }
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint64_t inc = wave_incl_scan_u64(s, lane);
    const uint32_t cnt = (uint32_t)__popc(alive);
    uint32_t incc = cnt;
    if (co.block_cnt) {
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(incc, o, 64); if (lane >= o) incc += t; }
    }
    if (lane == 63) { sm[w] = inc; smc[w] = incc; }
    __syncthreads();
    uint64_t off = block_off[blockIdx.x] + inc - s;
    for (int k = 0; k < w; ++k) off += sm[k];
#pragma unroll
    for (int k = 0; k < kScanItems; ++k)
        if (base + k < n) cdf[base + k] = off + v[k];
    if (leaders) {
        static_assert(kScanItems == 8 && (1 << kLeaderShift) == 16, "a leader is the last item of every second thread");
        const int64_t last = base + kScanItems - 1;
        if ((last & 15) == 15 && last < n) leaders[last >> kLeaderShift] = off + v[kScanItems - 1];
        else if (base < n && n - 1 <= last && ((n - 1) & 15) != 15) leaders[(n - 1) >> kLeaderShift] = off + v[(n - 1) - base];   // ragged last group
    }
    if (co.block_cnt && alive) {
        uint32_t pos = co.block_cnt[blockIdx.x] + incc - cnt;
        for (int k = 0; k < w; ++k) pos += smc[k];
#pragma unroll
        for (int k = 0; k < kScanItems; ++k)
            if ((alive >> k) & 1u) {
                if (pos < co.cap) {
                    const int64_t i = base + k;
                    const uint64_t c = off + v[k];
                    co.ccdf[pos] = c;
                    co.cidx[pos] = (uint32_t)i;
                    co.crec[pos] = make_double4(co.x[i], co.y[i], co.th[i], 0.0);
                    if ((pos & ((1u << kCompactGroupShift) - 1u)) == (1u << kCompactGroupShift) - 1u) co.ctop[pos >> kCompactGroupShift] = c;
                }
                ++pos;
            }
    }
}

// ------------------------------------------------------------------------------------------------
// K6 + K1 + K2: one thread per child m (global child index child_first + m).
//   threshold:   multinomial  k53_m * Q   vs   C_i * 2^53      (k53 = floor(u*2^53), u Philox or injected)
//                systematic   (m*2^32 + k0) * Q   vs   C_i * (N_children * 2^32)
//   idx = first i with C_i*lmul > rhs  (C inclusive)  == upper_bound; Q == 0 -> idx = 0 (reference:
//   NaN CDF -> every draw returns 0).
//   then row gather (cpp:664) and the motion model (cpp:474-502) with the three normals.
// ------------------------------------------------------------------------------------------------

#ifndef MCL_SORT_SUB
#define MCL_SORT_SUB 1
#endif
constexpr int kSortSub = MCL_SORT_SUB;               // sort cells per grid cell and axis (the ordering kernels below)
// (defined with the ordering kernels below; the resampling kernel makes the sort keys of its children itself when it is given a layout)
__device__ __forceinline__ int cell_of(double g, int hi);
__device__ __forceinline__ uint32_t sort_key(const int *__restrict__ bbox, const int *__restrict__ tilemap, int ntx_abs, int cx, int cy, double th,
                                         int64_t n, double fx = 0.0, double fy = 0.0);

__device__ __forceinline__ void prep_small_clear(const PrepClear &clr, int i, int nthreads)
{
    if (clr.fix_count) for (int k = i; k < clr.fix_words; k += nthreads) clr.fix_count[k] = 0ull;
    if (clr.fix_over && i < 2) clr.fix_over[i] = 0ull;
    if (clr.exact_count && i == 0) clr.exact_count[0] = 0ull;
    if (clr.far_count && i == 0) clr.far_count[0] = 0ull;
    if (clr.bbox && i < 7) clr.bbox[i] = i < 2 ? 0x7fffffff : (i < 4 ? (int)0x80000000 : (i == 6 ? clr.bbox_play : 0));
}

struct ResampleArgs {
    const double *px, *py, *pth;      // parents
    const uint64_t *cdf;              // inclusive CDF over parents
    const uint64_t *tile_excl;        // exclusive prefix before each kScanTile-sized tile of the CDF (the scan's spine)
    const uint64_t *leaders;          // last CDF entry of every 16-entry group (with tile_excl), or null
    const double4 *ppack;             // parents as (x, y, theta, -) records: one fetch per gathered parent, or null
    double4 *cpack;                   // children in the same form for the next update, or null
    int64_t n_parents;
    uint64_t q_total;
    double *cx, *cy, *cth;            // children out
    int32_t *idx_out;                 // may be null
    int64_t n_children;               // children handled by this launch
    int64_t child_first;              // global index of child 0 of this launch
    int64_t n_children_total;         // global number of children (systematic spacing)
    int mode;                         // 0 multinomial, 1 systematic
    const double *uniforms;           // injected uniforms (per local child) or null
    const double *normals;            // injected normals n x 3 row-major or null
    uint32_t seed_lo, seed_hi, update_idx, k0;
    double dt, v, w;                  // motion scalars (cpp:452-471, computed on the host)
    double disp_x, disp_y, disp_th;
    int do_resample;                  // 0: children = parents (identity): a kept update of adaptive resampling (E9), tests
    int64_t idx_out_base;             // ... whose parent index is reported as idx_out_base + own index (a shard's first global index)
    int do_motion;
    // sharded particle sets: parents of other shards are fetched on demand instead of being gathered
    const double4 *ppack_rank[kMaxShards];   // records of shard r (peer pointers within one process), or all null
    int64_t n_per_rank;               // parents per shard (global index = r * n_per_rank + local index)
    int self_rank;
    unsigned long long *remote_count; // += children whose parent lives in another shard (exchange accounting), or null
    const int32_t *idx_in;            // parent of every child decided earlier (index-only pass + exchange), or null
    int index_only;                   // 1: write idx_out and stop (the host fetches the selected parents, then calls again)
    // small updates (one launch less each): the per-particle constants k_particle_prep would compute, the CDF staged in
    // LDS (cdf_lds_entries = n_parents when the launch has n_parents * 8 bytes of dynamic LDS, else 0), counters to zero
    double4 *pc_out;                  // (cos, sin, pixel x, pixel y) per child, or null
    double *clr_logw_acc;             // with pc_out, for the windowed ray kernels: the per-particle scratch k_particle_prep
    uint32_t *clr_far_flags;          //   would zero (the launch then needs k_prep_small only), or null
    double ox, oy, res;
    int cdf_lds_entries;
    unsigned long long *clear_counters;   // 4 words zeroed by the first thread, or null
    // compact parent list (mcl::CompactOut: the particles with a non-zero fixed-point weight, in index order): the search
    // runs over ccdf / ctop, the parent's index and record come from the list.  Sharded sets gather the shards' lists as
    // chunks of cchunk_bytes ([ccdf | crec | cidx], ccap entries each; shard r's at cchunks + r * cchunk_bytes) and search
    // gcdf, the chunks' CDF columns made global (offsets added, padding turned into plateaus) by k_compact_merge.
    const uint64_t *ccdf;             // n_compact entries, non-decreasing, or null (dense search)
    const uint64_t *ctop;             // ccdf[64 g + 63], n_compact >> 6 entries
    int64_t n_compact;
    const uint32_t *cidx;             // single list: particle index of entry p ...
    const double4 *crec;              // ... and its record
    const unsigned char *cchunks;     // gathered lists (cidx / crec null): entry p lives in chunk p / ccap
    int64_t cchunk_bytes, ccap;
    // the ordering of the ray stage: (key, index) of every child from the layout (bounding box, occupied tiles) of the PREVIOUS update's
    // children -- the set moves by a cell or so per update, and a key only decides which rays share a wave -- so that the radix
    // sort can start right after this kernel (k_cell_bbox / k_tile_compact / k_sort_keys off the critical path)
    uint32_t *key_out, *val_out;      // null: keys are made by k_sort_keys / k_sort_hist
    const int *key_bbox, *key_tilemap;
    int key_ntx, key_Wp, key_Hp;
    PrepClear prep;                   // prep_on: the first workgroup also clears the few words of the ray stage that are not per
    int prep_on;                      //   particle (what k_prep_small would do in a launch of its own)
    const float *obs_src;             // this update's ranges (pinned host memory, read once by the first workgroup), or null
    int32_t *obs_idx_out;             // their table rows (obs_index_of), for a ray kernel that reads the static table directly
    int obs_B, obs_P;
};

__device__ __forceinline__ int obs_index_of(float obs, double res, int P);

// per-particle constants of the ray stage: (cos, sin, pixel x, pixel y); a garbage heading gets a NaN pixel position (the
// pair fails every window test and is marched literally)
__device__ __forceinline__ double4 particle_constants(double x, double y, double t, double ox, double oy, double res)
{
    double s, c;
    sincos(t, &s, &c);
    const bool heading_ok = t == t && fabs(t) < 1e6;
    const double nanv = __longlong_as_double(0x7ff8000000000000ll);
    return make_double4(c, s, heading_ok ? (x - ox) / res : nanv, heading_ok ? (y - oy) / res : nanv);
}

__global__ __launch_bounds__(256) void k_resample_motion(ResampleArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char resample_lds[];
    int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (a.clear_counters && m == 0) { a.clear_counters[0] = 0ull; a.clear_counters[1] = 0ull; a.clear_counters[2] = 0ull; a.clear_counters[3] = 0ull; }
    if (a.obs_src && blockIdx.x == 0)
        for (int j = threadIdx.x; j < a.obs_B; j += blockDim.x) a.obs_idx_out[j] = obs_index_of(a.obs_src[j], a.res, a.obs_P);
    if (a.prep_on && blockIdx.x == 0) prep_small_clear(a.prep, (int)threadIdx.x, (int)blockDim.x);
    const uint64_t *cdf = a.cdf;
    if (a.cdf_lds_entries > 0) {
        // a small CDF: one coalesced pass into LDS, then the bisection runs at LDS latency (11 dependent L2 round trips
        // were most of this kernel at the stock 2000 particles)
        uint64_t *c_sh = reinterpret_cast<uint64_t *>(resample_lds);
        for (int i = threadIdx.x; i < a.cdf_lds_entries; i += blockDim.x) c_sh[i] = a.cdf[i];
        __syncthreads();
        cdf = c_sh;
    }
    if (m >= a.n_children) return;
    uint64_t g = (uint64_t)(a.child_first + m);
    int64_t idx = m;
    int64_t cpos = -1;                        // position in the compact parent list, when that is what was searched
    if (a.idx_in) {
        idx = a.idx_in[m];
    } else if (a.do_resample) {
        idx = 0;
        if (a.q_total != 0) {
            uint64_t lmul, r0, r1 = a.q_total;
            if (a.mode == 0) {
                uint64_t k53;
                if (a.uniforms) {
                    double u = a.uniforms[m];
                    u = (u >= 0.0) ? u : 0.0;
                    k53 = (uint64_t)(u * 9007199254740992.0);
                    if (k53 > 9007199254740991ull) k53 = 9007199254740991ull;
                } else {
                    u32x4 o = philox4x32((uint32_t)g, a.update_idx, 2u, (uint32_t)(g >> 32), a.seed_lo, a.seed_hi);
                    k53 = bits53(o.v[0], o.v[1]);
                }
                r0 = k53; lmul = 1ull << 53;
            } else {
                r0 = g * 4294967296ull + a.k0;
                lmul = (uint64_t)a.n_children_total * 4294967296ull;
            }
            // "entry c is above the threshold": c * lmul > r0 * r1 in exact 128-bit arithmetic.  With the multinomial draw's
            // lmul = 2^53 that is c > floor(r0 r1 / 2^53) for the integer c: one product per child instead of one per probe
            // (the search was a sixth of this kernel's instructions); the systematic comb keeps the product form.
            const bool by_thr = a.mode == 0;
            const uint64_t thr = (__umul64hi(r0, r1) << 11) | ((r0 * r1) >> 53);
            auto above = [&](uint64_t c) -> bool { return by_thr ? c > thr : mul_gt(c, lmul, r0, r1); };
            if (a.ccdf) {
                // compact list: first group of 64 whose last entry exceeds the threshold (ctop: dense, a few KB), then
                // inside that group's 512 bytes
                const int64_t ng = a.n_compact >> kCompactGroupShift;
                int64_t glo = 0, glen = ng;
                while (glen > 0) {
                    int64_t half = glen >> 1, mid = glo + half;
                    if (!above(a.ctop[mid])) { glo = mid + 1; glen = glen - half - 1; }
                    else glen = half;
                }
                int64_t lo = glo << kCompactGroupShift;
                int64_t len = (lo + (1 << kCompactGroupShift) <= a.n_compact) ? (1 << kCompactGroupShift) : a.n_compact - lo;
                while (len > 0) {
                    int64_t half = len >> 1, mid = lo + half;
                    if (!above(a.ccdf[mid])) { lo = mid + 1; len = len - half - 1; }
                    else len = half;
                }
                cpos = (lo >= a.n_compact) ? a.n_compact - 1 : lo;
            } else {
            // two-level search: first the tile (its exclusive prefixes are a small, cache-resident array left by the
            // scan), then inside the tile's 16 KB of the CDF
            int64_t lo = 0, len = a.n_parents;
            if (a.tile_excl) {
                const int64_t ntiles = (a.n_parents + kScanTile - 1) / kScanTile;
                int64_t tlo = 0, tlen = ntiles;           // first tile whose exclusive prefix exceeds the threshold
                while (tlen > 0) {
                    int64_t half = tlen >> 1, mid = tlo + half;
                    if (!above(a.tile_excl[mid])) { tlo = mid + 1; tlen = tlen - half - 1; }
                    else tlen = half;
                }
                const int64_t t = tlo > 0 ? tlo - 1 : 0;
                lo = t * kScanTile;
                len = (lo + kScanTile <= a.n_parents) ? kScanTile : a.n_parents - lo;
                if (a.leaders) {
                    // first 16-entry group of the tile whose last entry exceeds the threshold, then only that group
                    const int64_t g0 = lo >> kLeaderShift, ng = (len + 15) >> kLeaderShift;
                    int64_t glo = g0, glen = ng;
                    while (glen > 0) {
                        int64_t half = glen >> 1, mid = glo + half;
                        if (!above(a.leaders[mid])) { glo = mid + 1; glen = glen - half - 1; }
                        else glen = half;
                    }
                    if (glo >= g0 + ng) glo = g0 + ng - 1;          // threshold beyond the tile (only at the very end)
                    const int64_t end = lo + len;
                    lo = glo << kLeaderShift;
                    len = (lo + 16 <= end) ? 16 : end - lo;
                }
            }
            while (len > 0) {
                int64_t half = len >> 1, mid = lo + half;
                if (!above(cdf[mid])) { lo = mid + 1; len = len - half - 1; }
                else len = half;
            }
            idx = (lo >= a.n_parents) ? a.n_parents - 1 : lo;
            }
        }
    }
    double x, y, th;
    bool have_rec = false;
    if (cpos >= 0) {
        // index and record of the selected entry of the compact list
        if (a.cchunks) {
            const int64_t r = cpos / a.ccap, e = cpos - r * a.ccap;
            const unsigned char *chunk = a.cchunks + (size_t)r * (size_t)a.cchunk_bytes;
            const double4 pr = reinterpret_cast<const double4 *>(chunk + (size_t)a.ccap * 8)[e];
            idx = r * a.n_per_rank + (int64_t)reinterpret_cast<const uint32_t *>(chunk + (size_t)a.ccap * 40)[e];
            x = pr.x; y = pr.y; th = pr.z;
            if (a.remote_count) {
                const unsigned long long rem = __ballot(r != a.self_rank);
                if ((threadIdx.x & 63) == __ffsll((long long)__ballot(1)) - 1 && rem) atomicAdd(a.remote_count, (unsigned long long)__popcll(rem));
            }
        } else {
            const double4 pr = a.crec[cpos];
            idx = (int64_t)a.cidx[cpos];
            x = pr.x; y = pr.y; th = pr.z;
        }
        have_rec = true;
    }
    if (a.idx_out) a.idx_out[m] = (int32_t)((a.do_resample || a.idx_in) ? idx : idx + a.idx_out_base);
    if (a.index_only) return;
    if (have_rec) {
    } else if (a.n_per_rank > 0) {
        // the parent's record straight from the shard that owns it (this GPU or a peer over xGMI): only selected
        // parents ever cross a link, and a parent many children share is served from this GPU's L2 after the first fetch
        const int r = (int)(idx / a.n_per_rank);
        const double4 pr = a.ppack_rank[r][idx - (int64_t)r * a.n_per_rank];
        x = pr.x; y = pr.y; th = pr.z;
        if (a.remote_count) {
            const unsigned long long rem = __ballot(r != a.self_rank);
            if ((threadIdx.x & 63) == __ffsll((long long)__ballot(1)) - 1 && rem) atomicAdd(a.remote_count, (unsigned long long)__popcll(rem));
        }
    } else if (a.ppack) { const double4 pr = a.ppack[idx]; x = pr.x; y = pr.y; th = pr.z; }
    else { x = a.px[idx]; y = a.py[idx]; th = a.pth[idx]; }
    if (a.do_motion) {
        double n0, n1, n2;
        if (a.normals) {
            n0 = a.normals[3 * m + 0]; n1 = a.normals[3 * m + 1]; n2 = a.normals[3 * m + 2];
        } else {
            const double TWO_M53 = 1.0 / 9007199254740992.0;
            const double TWO_PI = 2.0 * 3.14159265358979323846;
            u32x4 o = philox4x32((uint32_t)g, a.update_idx, 0u, (uint32_t)(g >> 32), a.seed_lo, a.seed_hi);
            double u1 = (double)(bits53(o.v[0], o.v[1]) + 1) * TWO_M53;
            double u2 = (double)bits53(o.v[2], o.v[3]) * TWO_M53;
            double rad = sqrt(-2.0 * log(u1));
            double sn, cs;
            sincos(TWO_PI * u2, &sn, &cs);            // one argument reduction for the pair; the values are those of sin() and cos()
            n0 = rad * cs;
            n1 = rad * sn;
            o = philox4x32((uint32_t)g, a.update_idx, 1u, (uint32_t)(g >> 32), a.seed_lo, a.seed_hi);
            u1 = (double)(bits53(o.v[0], o.v[1]) + 1) * TWO_M53;
            u2 = (double)bits53(o.v[2], o.v[3]) * TWO_M53;
            rad = sqrt(-2.0 * log(u1));
            n2 = rad * cos(TWO_PI * u2);
        }
        double nx, ny, nth;
        if (fabs(a.w) < 1e-6) {                       // cpp:480-484
            double sn, cs;
            sincos(th, &sn, &cs);
            nx = x + a.v * a.dt * cs;
            ny = y + a.v * a.dt * sn;
            nth = th;
        } else {                                      // cpp:485-493
            double radius = a.v / a.w;
            double dth = a.w * a.dt;
            double s0, c0, s1, c1;
            sincos(th, &s0, &c0);
            sincos(th + dth, &s1, &c1);
            nx = x + radius * (s1 - s0);
            ny = y - radius * (c1 - c0);
            nth = th + dth;
        }
        nx += n0 * a.disp_x;                          // cpp:496-498
        ny += n1 * a.disp_y;
        nth += n2 * a.disp_th;
        nth = normalize_angle(nth);                   // cpp:501
        x = nx; y = ny; th = nth;
    }
    a.cx[m] = x; a.cy[m] = y; a.cth[m] = th;
    if (a.cpack) a.cpack[m] = make_double4(x, y, th, 0.0);
    if (a.pc_out) {
        const double4 c = particle_constants(x, y, th, a.ox, a.oy, a.res);
        a.pc_out[m] = c;
        if (a.key_out) {
            // the same key k_sort_keys would make (same function, same arguments), from the layout handed in
            a.key_out[m] = sort_key(a.key_bbox, a.key_tilemap, a.key_ntx, cell_of(c.z * kSortSub, a.key_Wp * kSortSub - 1),
                                    cell_of(c.w * kSortSub, a.key_Hp * kSortSub - 1), th, a.n_children, c.z * kSortSub - floor(c.z * kSortSub),
                                    c.w * kSortSub - floor(c.w * kSortSub));
            a.val_out[m] = (uint32_t)m;
        }
    }
    if (a.clr_logw_acc) a.clr_logw_acc[m] = 0.0;
    if (a.clr_far_flags) a.clr_far_flags[m] = 0u;
}

// The shards' compact lists, gathered as chunks ([ccdf | crec | cidx], ccap entries each), become ONE searchable CDF: chunk r's
// column plus the fixed-point total of the shards before it, its unused tail turned into a plateau at the shard's end value
// (a plateau entry is never the first one above a threshold, so it is never selected), and the first search level beside it.
struct MergeArgs {
    const unsigned char *chunks;
    int64_t chunk_bytes, ccap;
    int n_shards;
    uint32_t count[kMaxShards];
    uint64_t off[kMaxShards], tot[kMaxShards];
    uint64_t *gcdf, *gtop;
};
__global__ __launch_bounds__(256) void k_compact_merge(MergeArgs a)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= (int64_t)a.n_shards * a.ccap) return;
    const int r = (int)(p / a.ccap);
    const int64_t e = p - (int64_t)r * a.ccap;
    const uint64_t *col = reinterpret_cast<const uint64_t *>(a.chunks + (size_t)r * (size_t)a.chunk_bytes);
    const uint64_t v = a.off[r] + (e < (int64_t)a.count[r] ? col[e] : a.tot[r]);
    a.gcdf[p] = v;
    if ((p & ((1 << kCompactGroupShift) - 1)) == (1 << kCompactGroupShift) - 1) a.gtop[p >> kCompactGroupShift] = v;
}

// ---- distinct parents of a shard's children (sharded resampling): a bitmap over the GLOBAL particle indices, its popcount
//      prefix and two expansions.  Every pass but the marking is over n_total / 32 words. ----
__global__ void k_bm_mark(const int32_t *__restrict__ parent, int64_t n, int64_t n_total, uint32_t *__restrict__ bm)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t p = (uint32_t)parent[i];
    if ((int64_t)p < n_total) atomicOr(&bm[p >> 5], 1u << (p & 31u));
}
__global__ void k_bm_pop(const uint32_t *__restrict__ bm, int64_t nwords, uint64_t *__restrict__ pop)
{
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w < nwords) pop[w] = (uint64_t)__popc(bm[w]);
}
// pref = inclusive prefix of pop: the set bits of word w are distinct parents pref[w] - popc(word) ... pref[w] - 1
__global__ void k_bm_expand(const uint32_t *__restrict__ bm, const uint64_t *__restrict__ pref, int64_t nwords, int64_t *__restrict__ distinct)
{
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= nwords) return;
    uint32_t word = bm[w];
    int64_t at = (int64_t)pref[w] - __popc(word);
    while (word) {
        const int b = __ffs((int)word) - 1;
        distinct[at++] = w * 32 + b;
        word &= word - 1u;
    }
}
__global__ void k_bm_slot(const int32_t *__restrict__ parent, int64_t n, int64_t n_total, const uint32_t *__restrict__ bm,
                          const uint64_t *__restrict__ pref, int32_t *__restrict__ slot)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t p = (uint32_t)parent[i];
    if ((int64_t)p >= n_total) { slot[i] = 0; return; }
    const uint32_t word = bm[p >> 5];
    slot[i] = (int32_t)((int64_t)pref[p >> 5] - __popc(word) + __popc(word & ((1u << (p & 31u)) - 1u)));
}
// records of the listed particles (indices into the packed records of one shard)
__global__ void k_gather_records(const double4 *__restrict__ rec, const int64_t *__restrict__ index, int64_t count, int64_t n,
                                 double4 *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const int64_t k = index[i];
    out[i] = (k >= 0 && k < n) ? rec[k] : make_double4(0.0, 0.0, 0.0, 0.0);
}

// (x, y, theta) columns -> packed records (the form k_resample_motion gathers parents from)
__global__ void k_pack_records(const double *__restrict__ x, const double *__restrict__ y, const double *__restrict__ th, int64_t n,
                               double4 *__restrict__ out)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = make_double4(x[i], y[i], th[i], 0.0);
}

// lidarCB's downsampled range -> table row of a beam, cpp:549-554, 570, 573 (NaN -> 0)
__device__ __forceinline__ int obs_index_of(float obs, double res, int P)
{
    float px = (float)((double)obs / res);
    if (px > (float)P) px = (float)P;
    float r = roundf(px);
    int idx;
    if (r != r) idx = 0;
    else if (r <= -2147483648.0f) idx = 0;
    else idx = (int)r;
    idx = idx > P ? P : idx;
    idx = idx < 0 ? 0 : idx;
    return idx;
}

// ------------------------------------------------------------------------------------------------
// Particle initialisation on the device.  Philox streams 5/6 (pose cloud) and 7 (global).
//   pose cloud, cpp:390-398: x = pose_x + n0*0.5, y = pose_y + n1*0.5, theta = wrap(pose_t + n2*0.4)
//   global, cpp:433-441:     cell = free[floor(u*n_free)], x = col*res + ox, y = row*res + oy, theta = u2*2pi
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_init_pose(double px, double py, double pt, int64_t n, int64_t first, uint32_t seed_lo,
                                                  uint32_t seed_hi, uint32_t init_idx, double *__restrict__ x,
                                                  double *__restrict__ y, double *__restrict__ th)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t g = (uint64_t)(first + i);
    const double TWO_M53 = 1.0 / 9007199254740992.0;
    const double TWO_PI = 2.0 * 3.14159265358979323846;
    u32x4 o = philox4x32((uint32_t)g, init_idx, 5u, (uint32_t)(g >> 32), seed_lo, seed_hi);
    double u1 = (double)(bits53(o.v[0], o.v[1]) + 1) * TWO_M53;
    double u2 = (double)bits53(o.v[2], o.v[3]) * TWO_M53;
    double rad = sqrt(-2.0 * log(u1));
    double n0 = rad * cos(TWO_PI * u2), n1 = rad * sin(TWO_PI * u2);
    o = philox4x32((uint32_t)g, init_idx, 6u, (uint32_t)(g >> 32), seed_lo, seed_hi);
    u1 = (double)(bits53(o.v[0], o.v[1]) + 1) * TWO_M53;
    u2 = (double)bits53(o.v[2], o.v[3]) * TWO_M53;
    double n2 = sqrt(-2.0 * log(u1)) * cos(TWO_PI * u2);
    x[i] = px + n0 * 0.5;
    y[i] = py + n1 * 0.5;
    th[i] = normalize_angle(pt + n2 * 0.4);
}

__global__ __launch_bounds__(256) void k_init_global(const uint32_t *__restrict__ free_cells, uint64_t n_free, int W, double res,
                                                    double ox, double oy, int64_t n, int64_t first, uint32_t seed_lo,
                                                    uint32_t seed_hi, uint32_t init_idx, double *__restrict__ x,
                                                    double *__restrict__ y, double *__restrict__ th)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t g = (uint64_t)(first + i);
    u32x4 o = philox4x32((uint32_t)g, init_idx, 7u, (uint32_t)(g >> 32), seed_lo, seed_hi);
    uint64_t k = bits53(o.v[0], o.v[1]);
    uint64_t pick = __umul64hi(k << 11, n_free);            // floor(k/2^53 * n_free)
    uint32_t cell = free_cells[pick];
    int row = (int)(cell / (uint32_t)W), col = (int)(cell - (uint32_t)row * (uint32_t)W);
    x[i] = col * res + ox;                                   // cpp:438
    y[i] = row * res + oy;                                   // cpp:439
    th[i] = (double)bits53(o.v[2], o.v[3]) * (1.0 / 9007199254740992.0) * (2.0 * 3.14159265358979323846);
}

__global__ void k_fill(double *__restrict__ p, int64_t n, double v)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// per-update transposed log table: Lt[d * bpad + j] = L[obs_idx[j] * (P+1) + d], straight from the raw ranges (every thread
// derives its beam's table row; row 0's threads also publish obs_idx for the kernels that read it).
// Ltr holds the same rows in reverse order (row rho = P - d), the form k_rays_cell indexes with "samples left"
__global__ void k_obs_build_lt(const float *__restrict__ obs, double res, int P, const float *__restrict__ L, int B, int bpad,
                               int32_t *__restrict__ obs_idx, float *__restrict__ Lt, float *__restrict__ Ltr)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int d = blockIdx.y, tw = P + 1;
    if (j >= bpad) return;
    float v = 0.f;
    if (j < B) {
        const int oi = obs_index_of(obs[j], res, P);
        if (d == 0) obs_idx[j] = oi;
        v = L[(size_t)oi * tw + d];
    }
    Lt[(size_t)d * bpad + j] = v;
    Ltr[(size_t)(tw - 1 - d) * bpad + j] = v;
}

// ------------------------------------------------------------------------------------------------
// K3: ray cast + likelihood.
// ------------------------------------------------------------------------------------------------
struct RayArgs {
    const double *x, *y, *th;      // particles (this launch's)
    const double4 *pc;             // per particle (cos th, sin th, (x-ox)/res, (y-oy)/res), k_particle_prep
    const short4 *qr;              // per particle quadrant ranges of its beams (k_rays_quad), k_particle_prep
    const double4 *pcs;            // k_rays_cell: pc in cell-sorted order (k_sort_scatter)
    const uint32_t *perm;          // k_rays_cell: sorted slot -> particle index
    const double *ths;             // k_rays_cell: heading in cell-sorted order
    const double2 *slice_mean;     // k_rays_cell: mean pixel position of every slice of the sorted order (k_slice_means)
    const uint8_t *distw;          // k_rays_cell: kWedges wedge fields (mcl_wedge.h), field k at distw + k * distw_stride
    size_t distw_stride;
    const uint8_t *distg;          // k_rays_sweep<.., GLOBAL>: the same fields mirrored per quadrant, with a two-cell ring of stop bytes around the
    size_t distg_stride;           //   padded grid and a tail of stop rows: mirrored padded cell (y, x) of field k at distg + k * distg_stride +
    int distg_pitch;               //   (y + 2) * distg_pitch + (x + 2); distg is the START of the allocation (every offset the kernel forms is >= 0)
    int qside;                     // k_rays_quad: window side in cells (1 byte per cell)
    int nslices;                   // k_rays_quad: particle slices; grid = 4 * nslices
    unsigned long long *fix_list;  // k_rays_quad -> k_rays_fix: (particle << 16 | beam) of undecided rays
    unsigned long long *fix_count; // per workgroup of k_rays_quad (stride 8 words): entries appended (> fix_cap: overflow)
    unsigned long long fix_cap;    // capacity of one workgroup's segment
    int fix_segments;              // number of segments = workgroups of k_rays_quad
    unsigned long long *exact_list;   // k_rays_fix -> k_rays_exact: rays that need the literal march (level 3)
    unsigned long long *exact_count;  // entries appended (beyond exact_cap: marched inline by k_rays_fix)
    unsigned long long exact_cap;
    uint8_t *far_flags;            // [particle][quadrant]: pair does not fit its quadrant window -> k_rays_far
    unsigned long long *work_counter;  // k_rays_quad: next (slice, quadrant) item
    unsigned long long *dbg;       // optional [workgroup][4]: start, end (s_memrealtime, 100 MHz), HW_ID, XCC_ID
    int64_t n;
    int B, bpad, P;
    const double2 *beam_cs;        // (cos a_j, sin a_j) of (double)angle_f32[j], host fp64
    const double2 *beam_csx;       // k_rays_sweep: the same with beam_margin virtual beams before beam 0 and after beam B - 1 (entry j + beam_margin)
    int beam_pad, beam_margin;     // beams of a full wedge up to which a scan-edge lane is padded with virtual beams (0: never); see k_rays_sweep
    const double2 *beam_csi;       // k_rays_sweep<.., REC>: (cos, sin) of the GRID angle a0 + j inc of every table column (entry j + beam_margin)
    const double *beam_err;        //   and the beam's own offset from it, a_j - (a0 + j inc) (0 for virtual beams and padding): ltd_cols entries
    double rec_k;                  //   2 cos(inc): the three-term recurrence of the turned direction (MCL_SW_STEP_REC)
    const float *beam_angle;       // float angles (MARCH path uses theta + (double)angle)
    double beam_a0, beam_inv_inc;  // first angle and beams per radian (k_rays_cell's guess of a wedge's first beam)
    double beam_alast;             // last angle: (double)beam_angle[B - 1]
    const float *Lt;               // (P+1) x bpad
    const float *Ltr;              // the same with the rows reversed (row P - d)
    const double *Ltd;             // k_rays_sweep: fp64 table indexed by samples left + kSwUnder (mcl_rays_sweep.h), ltd_cols columns
    int ltd_cols;
    int sweep_g;                   // k_rays_sweep: wedges per work item
    int split16;                   // k_rays_sweep: passes of 9..16 chunks are handed out in halves too (small launches)
    const int4 *items;             // k_rays_sweep: work items (first unit, units, wedge group, run), big first (guided schedule)
    const int4 *centres;           // k_rays_sweep: per run of units (window centre as a padded cell x, y; first unit; units), k_sweep_plan
    int nitems;
    const int *nitems_ptr;          // k_rays_sweep: number of work items, written by k_sweep_plan
    const double4 *unit_sums;      // k_rays_sweep: per unit of the sorted order (sum px, sum py, count, -) and its bounding box, k_unit_sums
    const uint32_t *unit_begin;    // k_rays_sweep: first slot of every unit, one entry past the last (k_unit_table)
    uint32_t *far_list;            // k_rays_sweep -> k_rays_far: slots with at least one flagged quadrant (appended once each), or null
    unsigned long long *far_count; // entries in far_list
    const uint32_t *far_sorted;    // k_rays_skip<.., FAR>: the flagged slots in ascending (= spatial) order, k_far_scatter
    const unsigned long long *far_min;   // the windowed far pass runs when *far_count >= *far_min ... (k_rays_far: when below)
    int far_windowed;              // k_rays_far: 1 = a windowed far pass was launched beside it (stand down when it runs)
    int slot_space;                // 1: fix-list entries, far flags and `logw` are indexed by sorted slot (k_rays_sweep), and the
                                   // per-particle constants of k_rays_fix / k_rays_far come from pcs / ths; perm gives the particle
    double *logw;                  // out
    uint8_t *steps;                // out N*B or null
    uint16_t *steps16;             // the same as 16-bit entries, for maps whose range exceeds 255 px (k_rays_march / k_rays_skip)
    // map
    const int8_t *grid; int W, H;
    double res, ox, oy;
    const uint8_t *dist;           // padded distance field Hp x Wps bytes (0 = stop), cap 255
    const float *Ldirect;          // k_rays_skip, small updates: the static table L[row][step] read through obs_idx (no per-update
    const int32_t *obs_idx;        //   transposed copy is built); null = use Lt
    const uint8_t *dist4;          // the same field as nibbles min(d, 15), two cells per byte, Hp x Wps/2 bytes (k_rays_skip's window)
    const uint8_t *distq[4];       // directional fields per quadrant (k_rays_quad, k_rays_far)
    int Wp, Hp, Wps;
    int tw_cells;                  // LDS window side (multiple of 8)
    unsigned long long *counters;  // [0] exact-fallback rays, [1] particles off-window, [2] probes
    int force_exact;
};

// literal restatement of cast_ray (cpp:611-650) on the int8 grid; returns the step index
// (0..P-1) or P for "no hit within MAX_RANGE_PX samples".
__device__ __forceinline__ int march_exact(const RayArgs &a, double x, double y, double angle)
{
    double dx = cos(angle) * a.res;
    double dy = sin(angle) * a.res;
    double cx = x, cy = y;
    for (int step = 0; step < a.P; ++step) {
        cx += dx;
        cy += dy;
        int gx = (int)((cx - a.ox) / a.res);
        int gy = (int)((cy - a.oy) / a.res);
        if (gx < 0 || gx >= a.W || gy < 0 || gy >= a.H) return step;
        if (a.grid[(size_t)gy * a.W + gx] > 50) return step;
    }
    return a.P;
}

// step index of ray (i, j) to whichever output the launch has
__device__ __forceinline__ void store_step(const RayArgs &a, int64_t i, int j, int r)
{
    if (a.steps16) a.steps16[(size_t)i * a.B + j] = (uint16_t)r;
    else if (a.steps) a.steps[(size_t)i * a.B + j] = (uint8_t)r;
}

constexpr int kRayThreads = 1024;
constexpr int kRayWaves = kRayThreads / 64;

// per-particle constants of the skipping march, computed once by one lane instead of by all 64 lanes
// of the wave that owns the particle
// k_rays_quad assigns a ray to the quadrant floor((theta + a_j) / (pi/2)) of its direction.  Beam angles
// increase monotonically over less than a full turn (checked at mcl_set_beam_angles), so the beams of one
// particle fall into at most five contiguous index ranges with quadrants q0, q0+1, ..., q0+4 (mod 4).
// qr = (start of ranges 1..4, B when absent); q0 is packed into the top two bits of .x.
// direction bin (wedge, unwrapped) of beam `angle` for heading th; the quadrant index is derived from it so that
// every kernel classifies a ray the same way
__device__ __forceinline__ int beam_wedge(double th, float angle) { return (int)floor((th + (double)angle) * (kWedges * 0.15915494309189533577)); }
__device__ __forceinline__ int beam_wedge_d(double th, double angle) { return (int)floor((th + angle) * (kWedges * 0.15915494309189533577)); }   // angle = (double) of the float
__device__ __forceinline__ int beam_turns(double th, float angle) { return beam_wedge(th, angle) >> kWedgeShift; }

// quadrant ranges of one particle's beams (see above); a garbage heading gets one range in quadrant 0
__device__ __forceinline__ short4 quadrant_ranges_of(double t, bool heading_ok, const float *__restrict__ beam_angle, int B)
{
    short st[4];
    int q0 = 0;
    if (heading_ok) {
        const int k0 = beam_turns(t, beam_angle[0]);
        q0 = k0 & 3;
        for (int k = 1; k <= 4; ++k) {
            int lo = 0, hi = B;              // first j with turns(j) - turns(0) >= k
            while (lo < hi) {
                int mid = (lo + hi) >> 1;
                if (beam_turns(t, beam_angle[mid]) - k0 >= k) hi = mid; else lo = mid + 1;
            }
            st[k - 1] = (short)lo;
        }
    } else {
        st[0] = st[1] = st[2] = st[3] = (short)B;   // garbage heading: one range, marched literally by k_rays_far
    }
    return make_short4((short)(st[0] | (q0 << 14)), st[1], st[2], st[3]);
}

__global__ __launch_bounds__(256) void k_particle_prep(const double *__restrict__ x, const double *__restrict__ y,
                                                      const double *__restrict__ th, int64_t n, double ox, double oy, double res,
                                                      double4 *__restrict__ pc, const float *__restrict__ beam_angle, int B,
                                                      short4 *__restrict__ qr, PrepClear clr)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (clr.logw_acc) clr.logw_acc[i] = 0.0;
    if (clr.far_flags) clr.far_flags[i] = 0u;
    if (clr.fix_count) for (int64_t k = i; k < clr.fix_words; k += n) clr.fix_count[k] = 0ull;
    if (clr.fix_over) for (int64_t k = i; k < 2; k += n) clr.fix_over[k] = 0ull;
    if (clr.exact_count && i == 0) clr.exact_count[0] = 0ull;
    if (clr.far_count && i == 0) clr.far_count[0] = 0ull;
    if (clr.bbox) for (int64_t k = i; k < 7; k += n) clr.bbox[k] = k < 2 ? 0x7fffffff : (k < 4 ? (int)0x80000000 : (k == 6 ? clr.bbox_play : 0));
    if (clr.hist) for (int64_t k = i; k < clr.hist_n; k += n) clr.hist[k] = 0u;
    const double t = th[i];
    pc[i] = particle_constants(x[i], y[i], t, ox, oy, res);
    if (qr) qr[i] = quadrant_ranges_of(t, t == t && fabs(t) < 1e6, beam_angle, B);
}

// the few words of k_particle_prep's clearing that are not per particle: what is left to do when the resampling kernel has
// already written the constants and zeroed the per-particle scratch
__global__ __launch_bounds__(256) void k_prep_small(PrepClear clr)
{
    prep_small_clear(clr, (int)threadIdx.x, 256);
}

// D = a*b + c on the low 24 bits of a and b (v_mad_i32_i24): the level-1 position update
__device__ __forceinline__ uint32_t mad_i24(int a, int b, uint32_t c)
{
    uint32_t d;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
// D = a*b + c on the low 24 bits (unsigned): LDS byte offset = row * stride + column byte
__device__ __forceinline__ uint32_t mad_u24(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t d;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ uint32_t mad_u24_s(uint32_t a, uint32_t b_uniform, uint32_t c)
{
    uint32_t d;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "s"(b_uniform), "v"(c));
    return d;
}
// the same with a wave-uniform first factor kept in a scalar register
__device__ __forceinline__ uint32_t mad_i24_s(int a, int b, uint32_t c)
{
    uint32_t d;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=&v"(d) : "s"(a), "v"(b), "v"(c));
    return d;
}
// round-to-nearest-even double -> int32 through the 1.5*2^52 magic add (|x| < 2^31)
__device__ __forceinline__ int rint_i32(double x) { return __double2loint(x + 6755399441055744.0); }

// ---- K3a: literal march for every ray (MCL_RAYS_MARCH): the on-device ground truth -------------
template <bool COUNT>
__global__ __launch_bounds__(kRayThreads) void k_rays_march(RayArgs a)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t per = (a.n + gridDim.x - 1) / gridDim.x;
    const int64_t p_begin = (int64_t)blockIdx.x * per;
    const int64_t p_end = (p_begin + per < a.n) ? p_begin + per : a.n;
    unsigned long long cnt_probe = 0;
    for (int64_t i = p_begin + wave; i < p_end; i += kRayWaves) {
        const double x = a.x[i], y = a.y[i], th = a.th[i];
        double acc = 0.0;
        for (int j0 = 0; j0 < a.B; j0 += 64) {
            int j = j0 + lane;
            if (j < a.B) {
                int r = march_exact(a, x, y, th + (double)a.beam_angle[j]);   // cpp:533
                acc += (double)a.Lt[(size_t)r * a.bpad + j];
                store_step(a, i, j, r);
                if (COUNT) cnt_probe += (r < a.P) ? (r + 1) : a.P;
            }
        }
        acc = wave_sum(acc);
        if (lane == 0) a.logw[i] = acc;
    }
    if (COUNT && a.counters) {
        cnt_probe = wave_sum_u64(cnt_probe);
        if (lane == 0 && cnt_probe) atomicAdd(&a.counters[2], cnt_probe);
    }
}

// ---- K3b: empty-space skipping on the same sample lattice (MCL_RAYS_SKIP) ----------------------
//
// A ray's k-th sample is p0 + k*u (pixel units).  D(c) = Chebyshev distance from cell c to the
// nearest "stop" cell (occupied or outside the map).  Successive samples are at most one cell apart
// in either axis, so if sample k lies in cell c the samples k+1 .. k+D(c)-1 cannot be stops: the
// march may jump from k to k+max(D(c),1) and still returns exactly the first stop sample of the
// fixed-step march (cpp:622-647).  Three precision levels keep that claim bit-exact:
//   level 1  32-bit fixed point (22 fractional bits) on the LDS window: T = P0 + s*U with
//            P0 = rint(p0*2^22), U = rint(u*2^22)  =>  |T - exact| <= (1+s)/2 <= 128 units (s <= 255);
//            a sample with fraction in [kG1, 2^22-kG1) units is in the cell the reference computes;
//   level 2  rays with any level-1 sample closer than that to a cell boundary are re-run with fp64
//            positions (error ~1 unit of 2^-32 px, guard 2^-30 px);
//   level 3  what is still ambiguous, and every ray of a particle whose own cell is ambiguous,
//            is re-run by march_exact (the literal restatement).
// The reference's own accumulated rounding (207 sequential fp64 adds at |x| ~ 85 m) stays below
// 3e-11 px, far inside the level-2 guard, which is what makes the equality with cpp:611-650 hold.
constexpr int kFx = 22;
constexpr uint32_t kG1 = 132u;
// fp64 fixed-point extraction: t = p + kMagic puts floor(p)+2^19 in the low 20 bits of the high
// dword and the fraction (2^-32 units, biased by +4) in the low dword; lo < kGuard <=> within 2^-30 px.
constexpr double kMagic = 1572864.0 + 0x1p-30;   // 1.5 * 2^20 + 2^-30
constexpr uint32_t kGuard = 8u;
constexpr int kCellBase = 1 << 19;

// fp64 skipping march of one ray, on the LDS nibble window (LDSWIN) or on the global byte field.
template <bool LDSWIN, bool COUNT>
__device__ __forceinline__ int trace_fp64(const RayArgs &a, const unsigned char *ldsb, int strideB, int base, double p0x, double p0y,
                                          double ux, double uy, int s0, uint32_t &amb, unsigned &np, const uint8_t *field = nullptr,
                                          bool wedge_coded = false)
{
    const uint8_t *gf = field ? field : a.dist;
    int s = s0, r = a.P;
    if (s > a.P) return r;
    while (true) {
        double sd = (double)s;
        double tx = __builtin_fma(sd, ux, p0x);
        double ty = __builtin_fma(sd, uy, p0y);
        uint32_t lox = (uint32_t)__double2loint(tx), loy = (uint32_t)__double2loint(ty);
        int cx = (__double2hiint(tx) & 0xFFFFF) - base;
        int cy = (__double2hiint(ty) & 0xFFFFF) - base;
        uint32_t mlo = lox < loy ? lox : loy;
        amb = amb < mlo ? amb : mlo;
        int d;
        if (LDSWIN) {
            uint32_t byte = ldsb[cy * strideB + (cx >> 1)];
            d = (byte >> ((cx & 1) * 4)) & 15;
        } else {
            d = ((unsigned)cx < (unsigned)a.Wp && (unsigned)cy < (unsigned)a.Hp) ? gf[(size_t)cy * a.Wps + cx] : 0;
            if (wedge_coded) d = d == 255 ? 0 : d;        // a wedge field (mcl_wedge.h) as the windows hold it: 0xFF = stop, skips 1..127
        }
        if (COUNT) ++np;
        if (d == 0) { r = s - 1; break; }
        s += d;
        if (s > a.P) break;
    }
    return r;
}

// R = rays per lane per pass (independent dependency chains that hide the LDS round trip).
// FAR: the windowed pass over the (particle, quadrant) pairs k_rays_sweep could not fit into its 256-cell windows -- the
// uniform cloud of a global re-localisation, where 1024 consecutive sorted particles cover more cells than a window leaves
// room for.  The flagged slots arrive in ascending (= tile-sorted) order; persistent workgroups take slices of kFarSlice
// of them, centre this kernel's 568-cell nibble window on each slice (a 32 x 32-cell tile or two fit its play at any
// supported range) and trace only the beams of the flagged quadrants; the sums are added to the slot's accumulator.
constexpr int kFarSlice = 256;
constexpr unsigned long long kFarWindowedMin = 2048;     // fewer flagged slots than this: k_rays_far (no window loads)
template <int R, bool COUNT, bool FAR = false>
__global__ __launch_bounds__(kRayThreads) void k_rays_skip(RayArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned long long cnt_exact = 0, cnt_off = 0, cnt_probe = 0, cnt_l2 = 0;
    const int64_t n_items = FAR ? (int64_t)min((unsigned long long)a.n, *a.far_count) : a.n;
    if (FAR && (unsigned long long)n_items < kFarWindowedMin) return;
    const uint32_t *flags32 = reinterpret_cast<const uint32_t *>(a.far_flags);
    for (int64_t slice = blockIdx.x; FAR ? slice * kFarSlice < n_items : slice == (int64_t)blockIdx.x; slice += gridDim.x) {
    if (FAR && slice != (int64_t)blockIdx.x) __syncthreads();          // every wave is done with the previous slice's window
    const int64_t per = FAR ? kFarSlice : (a.n + gridDim.x - 1) / gridDim.x;
    const int64_t p_begin = (FAR ? slice : (int64_t)blockIdx.x) * per;
    const int64_t p_end = (p_begin + per < n_items) ? p_begin + per : n_items;

    const int TW = a.tw_cells;
    const int strideB = TW >> 1;
    const int wpr = TW >> 3;                 // 32-bit words (8 cells) per window row
    // requested before the window is loaded, needed after: this wave's first particle and the lane's first beam direction
    // (a small update is one particle per wave: these two round trips would otherwise follow the fill)
    const int64_t i_first = p_begin + wave;
    double4 pci_first = make_double4(0.0, 0.0, 0.0, 0.0);
    if (!FAR && i_first < p_end) pci_first = a.pc[i_first];
    const double2 cs_first = a.beam_cs[lane];          // beam_cs is padded to a multiple of 64 * R entries
    const int tw = a.P + 1;
    int wx0, wy0;
    {
        // ---- window placement: centred on the mean padded-pixel position of this slice (FAR: on the centre of its bounding
        //      box -- a slice that straddles two neighbouring tiles, 64 x 32 cells, then fits the play as a whole) ----
        double *red = reinterpret_cast<double *>(lds_raw);   // scratch, overwritten by the window below
        double sx = 0.0, sy = 0.0, bx0 = INFINITY, bx1 = -INFINITY, by0 = INFINITY, by1 = -INFINITY;
        for (int64_t i = p_begin + threadIdx.x; i < p_end; i += kRayThreads) {
            double4 c = FAR ? a.pcs[a.far_sorted[i]] : a.pc[i];
            double gx = c.z, gy = c.w;
            if (gx == gx && gy == gy && fabs(gx) < 1e9 && fabs(gy) < 1e9) {
                sx += gx; sy += gy;
                if (FAR) { bx0 = fmin(bx0, gx); bx1 = fmax(bx1, gx); by0 = fmin(by0, gy); by1 = fmax(by1, gy); }
            }
        }
        double mx = 0.0, my = 0.0;
        if (FAR) {
            bx0 = -wave_max(-bx0); bx1 = wave_max(bx1); by0 = -wave_max(-by0); by1 = wave_max(by1);
            if (lane == 0) { red[4 * wave] = bx0; red[4 * wave + 1] = bx1; red[4 * wave + 2] = by0; red[4 * wave + 3] = by1; }
            __syncthreads();
            for (int k = 0; k < kRayWaves; ++k) { bx0 = fmin(bx0, red[4 * k]); bx1 = fmax(bx1, red[4 * k + 1]); by0 = fmin(by0, red[4 * k + 2]); by1 = fmax(by1, red[4 * k + 3]); }
            if (bx1 >= bx0) { mx = 0.5 * (bx0 + bx1); my = 0.5 * (by0 + by1); }
        } else {
            sx = wave_sum(sx); sy = wave_sum(sy);
            if (lane == 0) { red[2 * wave] = sx; red[2 * wave + 1] = sy; }
            __syncthreads();
            for (int k = 0; k < kRayWaves; ++k) { mx += red[2 * k]; my += red[2 * k + 1]; }
            int64_t cntp = p_end - p_begin;
            if (cntp > 0) { mx /= (double)cntp; my /= (double)cntp; }
        }
        wx0 = (((int)floor(mx) + 1 - TW / 2)) & ~7;      // padded coordinate = global + 1
        wy0 = (int)floor(my) + 1 - TW / 2;
        __syncthreads();
        // ---- load the window: a straight copy of the nibble field, 8 cells per 32-bit word, twenty independent loads in
        //      flight per thread (at the stock 2000 x 61 the fill is a third of the kernel: dependent round trips to L2) ----
        uint32_t *win = reinterpret_cast<uint32_t *>(lds_raw);
        const uint32_t *src4 = reinterpret_cast<const uint32_t *>(a.dist4);
        const int nwords = wpr * TW, spw = a.Wps >> 3, gxw0 = wx0 >> 3;        // wx0 is a multiple of 8 (possibly negative)
        constexpr int kBatch = 20;
        for (int wi0 = threadIdx.x; wi0 < nwords; wi0 += kRayThreads * kBatch) {
            uint32_t v[kBatch];
#pragma unroll
            for (int k = 0; k < kBatch; ++k) {
                const int wi = wi0 + k * kRayThreads;
                const int row = wi / wpr, cw = wi - row * wpr;
                const int gy = wy0 + row, gxw = gxw0 + cw;
                v[k] = 0u;
                if (wi < nwords && gy >= 0 && gy < a.Hp && gxw >= 0 && gxw < spw) v[k] = src4[(size_t)gy * spw + gxw];
            }
#pragma unroll
            for (int k = 0; k < kBatch; ++k) {
                const int wi = wi0 + k * kRayThreads;
                if (wi < nwords) win[wi] = v[k];
            }
        }
        __syncthreads();
    }
    const unsigned char *ldsb = lds_raw;
    const int ngroups = (a.B + 64 * R - 1) / (64 * R);
    // the level-1 loop addresses the window with raw LDS offsets: the dynamic segment must start at 0, i.e. the kernel has no
    // static LDS -- checked on the host at mcl_create (hipFuncGetAttributes), which keeps the engine off this kernel otherwise
    // level-1 error bound: (1 + s) / 2 units for s <= P samples (+ 4 of slack); kG1 covers the byte ranges
    const uint32_t g1 = (uint32_t)a.P > 255u ? (uint32_t)(a.P + 2) / 2u + 4u : kG1;
    uint32_t strideB_v = (uint32_t)strideB, gbias_v = g1 << (32 - kFx);
    asm volatile("" : "+v"(strideB_v), "+v"(gbias_v));   // keep both in VGPRs across the loop

    for (int64_t ii = p_begin + wave; ii < p_end; ii += kRayWaves) {
        // FAR: list entry ii is a sorted slot; its constants, heading and flags live in slot space, its particle index is perm[slot]
        const int64_t slot = FAR ? (int64_t)a.far_sorted[ii] : ii;
        const int64_t i = FAR ? (int64_t)a.perm[slot] : ii;
        const double4 pci = FAR ? a.pcs[slot] : ((ii == i_first) ? pci_first : a.pc[ii]);
        const uint32_t farfl = FAR ? flags32[slot] : 0u;
        const double hth = FAR ? a.ths[slot] : 0.0;
        const bool hth_ok = hth == hth && fabs(hth) < 1e6;
        double acc = 0.0;
        const double cth = pci.x, sth = pci.y;
        const double gpx = pci.z;                     // global pixel coordinate of the particle
        const double gpy = pci.w;
        const double wpx = gpx - (double)(wx0 - 1);   // window-relative padded coordinate
        const double wpy = gpy - (double)(wy0 - 1);
        const double reach = (double)(a.P + 2);
        const bool inwin = (wpx - reach >= 0.0) && (wpx + reach < (double)TW) && (wpy - reach >= 0.0) && (wpy + reach < (double)TW);
        // outside the window the same algorithm runs on the global byte field, in padded global
        // coordinates shifted by 2^18 so that the magic add sees positive values
        const bool sane = (gpx > -200000.0) && (gpx < 200000.0) && (gpy > -200000.0) && (gpy < 200000.0);
        const double p0x = inwin ? (wpx + kMagic) : ((gpx + 1.0 + 262144.0) + kMagic);
        const double p0y = inwin ? (wpy + kMagic) : ((gpy + 1.0 + 262144.0) + kMagic);
        const int base = inwin ? kCellBase : (kCellBase + 262144);
        if ((FAR || !inwin) && lane == 0) ++cnt_off;      // FAR: every pair here is one k_rays_sweep's windows did not fit
        // the particle's own cell gives a first skip shared by all its beams
        uint32_t amb0 = 0;
        int s0 = 1;
        if (inwin || sane) {
            uint32_t lox = (uint32_t)__double2loint(p0x), loy = (uint32_t)__double2loint(p0y);
            int cx = (__double2hiint(p0x) & 0xFFFFF) - base;
            int cy = (__double2hiint(p0y) & 0xFFFFF) - base;
            amb0 = lox < loy ? lox : loy;
            int d;
            if (inwin) {
                uint32_t byte = ldsb[cy * strideB + (cx >> 1)];
                d = (byte >> ((cx & 1) * 4)) & 15;
            } else {
                d = ((unsigned)cx < (unsigned)a.Wp && (unsigned)cy < (unsigned)a.Hp) ? a.dist[(size_t)cy * a.Wps + cx] : 0;
            }
            s0 = d > 1 ? d : 1;
        }

        if (inwin) {
            const uint32_t P0x = (uint32_t)rint_i32(wpx * 4194304.0 - 2147483648.0) + 0x80000000u;   // rint(wpx*2^22) mod 2^32
            const uint32_t P0y = (uint32_t)rint_i32(wpy * 4194304.0 - 2147483648.0) + 0x80000000u;
            const uint32_t g0 = (amb0 < kGuard) ? 0u : 0xFFFFFFFFu;
            const int rem_start = s0 <= a.P ? a.P - s0 : 0;
            const double ncth22 = -cth * 4194304.0, sth22 = sth * 4194304.0;
            for (int grp = 0; grp < ngroups; ++grp) {
                int NUx[R], NUy[R], rem[R], n[R];
                uint32_t Pex[R], Pey[R], g[R];
                bool want[R];
#pragma unroll
                for (int k = 0; k < R; ++k) {
                    int j = (grp * R + k) * 64 + lane;        // beam_cs is padded: no clamp needed
                    double2 cs = (grp == 0 && k == 0) ? cs_first : a.beam_cs[j];
                    // -U = -rint(2^22 * (cos, sin)(theta + a_j)); T(s) = P0 + s*U = Pe - rem*U with rem = P - s
                    NUx[k] = rint_i32(__builtin_fma(ncth22, cs.x, sth22 * cs.y));
                    NUy[k] = rint_i32(__builtin_fma(ncth22, cs.y, -(sth22 * cs.x)));
                    Pex[k] = mad_i24(-a.P, NUx[k], P0x);
                    Pey[k] = mad_i24(-a.P, NUy[k], P0y);
                    rem[k] = (j < a.B) ? rem_start : 0;       // a padding slot probes its end sample once
                    if (FAR) {
                        // only the beams of a flagged quadrant (the others were traced by k_rays_sweep); the quadrant of a
                        // beam is derived as everywhere else (beam_turns), a garbage heading has all its beams in quadrant 0
                        const int qj = (j < a.B) ? (hth_ok ? (beam_turns(hth, a.beam_angle[j]) & 3) : 0) : 0;
                        want[k] = j < a.B && ((farfl >> (8 * qj)) & 0xFFu) != 0u;
                        if (!want[k]) rem[k] = 0;
                    }
                    g[k] = g0;
                    n[k] = 0;
                }
                // ---- level 1: all R rays of the lane advance together; a finished ray keeps
                //      re-probing its last sample (same state every trip) until the wave is done ----
                bool any;
                do {
                    any = false;
#pragma unroll
                    for (int k = 0; k < R; ++k) {
                        // One hand-scheduled block per probe (13 VALU + 1 LDS read): position update, LDS byte
                        // address = row * stride + (cx >> 1) (the window starts at LDS offset 0), boundary guard
                        // and nibble select overlapped with the LDS round trip.
                        uint32_t Tx, Ty, t0, t1, addr, byte;
                        asm volatile(
                            "v_mad_i32_i24 %[tx], %[rem], %[nux], %[pex]\n\t"     // Tx = rem*NUx + Pex
                            "v_mad_i32_i24 %[ty], %[rem], %[nuy], %[pey]\n\t"     // Ty = rem*NUy + Pey
                            "v_lshrrev_b32 %[t0], 23, %[tx]\n\t"                  // cx >> 1
                            "v_lshrrev_b32 %[t1], 22, %[ty]\n\t"                  // cy
                            "v_mad_u32_u24 %[ad], %[t1], %[str], %[t0]\n\t"       // byte address
                            "ds_read_u8 %[by], %[ad]\n\t"
                            "v_lshl_add_u32 %[t0], %[tx], 10, %[gb]\n\t"          // biased fraction of x in the top 22 bits
                            "v_lshl_add_u32 %[t1], %[ty], 10, %[gb]\n\t"
                            "v_min3_u32 %[g], %[g], %[t0], %[t1]\n\t"             // closest approach to a cell boundary so far
                            "v_lshrrev_b32 %[t0], 20, %[tx]\n\t"
                            "v_and_b32 %[t0], 4, %[t0]\n\t"                       // nibble select: (cx & 1) * 4
                            "s_waitcnt lgkmcnt(0)\n\t"
                            "v_bfe_u32 %[by], %[by], %[t0], 4"                     // skip distance, 0 on a stop
                            : [tx] "=&v"(Tx), [ty] "=&v"(Ty), [t0] "=&v"(t0), [t1] "=&v"(t1), [ad] "=&v"(addr), [by] "=&v"(byte),
                              [g] "+v"(g[k])
                            : [rem] "v"(rem[k]), [nux] "v"(NUx[k]), [nuy] "v"(NUy[k]), [pex] "v"(Pex[k]), [pey] "v"(Pey[k]),
                              [str] "v"(strideB_v), [gb] "v"(gbias_v)
                            : "memory");
                        n[k] = (int)byte;
                        uint32_t nr;
                        bool over = __builtin_usub_overflow((uint32_t)rem[k], (uint32_t)n[k], &nr);   // skip > samples left
                        bool go = !over && n[k] != 0;
                        if (R == 1) {
                            rem[k] = (int)nr;                                                    // unchanged on a stop
                        } else {
                            rem[k] = go ? (int)nr : rem[k];
                        }
                        any |= go;
                        if (COUNT) cnt_probe += go ? 1 : 0;
                    }
                } while (any);
#pragma unroll
                for (int k = 0; k < R; ++k) {
                    int j = (grp * R + k) * 64 + lane;
                    if (j < a.B && (!FAR || want[k])) {
                        // R == 1: rem was decremented once more after an overshoot; only the stop case reads it
                        int r = (n[k] == 0) ? a.P - rem[k] - 1 : a.P;
                        if (COUNT) ++cnt_probe;
                        if (g[k] < ((2u * g1) << (32 - kFx)) || a.force_exact) {
                            // ---- level 2 (and 3): rare, everything recomputed from scratch ----
                            double2 cs = a.beam_cs[j];
                            double ux = cth * cs.x - sth * cs.y;
                            double uy = sth * cs.x + cth * cs.y;
                            uint32_t amb = amb0;
                            unsigned np = 0;
                            r = trace_fp64<true, COUNT>(a, ldsb, strideB, kCellBase, p0x, p0y, ux, uy, s0, amb, np);
                            ++cnt_l2;
                            if (COUNT) cnt_probe += np;
                            if (amb < kGuard || a.force_exact == 1) {
                                r = march_exact(a, a.x[i], a.y[i], a.th[i] + (double)a.beam_angle[j]);
                                ++cnt_exact;
                            }
                        }
                        acc += a.Ldirect ? (double)a.Ldirect[(size_t)a.obs_idx[j] * tw + r] : (double)a.Lt[(size_t)r * a.bpad + j];
                        store_step(a, i, j, r);
                    }
                }
            }
        } else {
            for (int j0 = 0; j0 < a.B; j0 += 64) {
                int j = j0 + lane;
                bool mine = j < a.B;
                if (FAR && mine) {
                    const int qj = hth_ok ? (beam_turns(hth, a.beam_angle[j]) & 3) : 0;
                    mine = ((farfl >> (8 * qj)) & 0xFFu) != 0u;
                }
                if (mine) {
                    int r = a.P;
                    unsigned np = 0;
                    uint32_t amb = amb0;
                    if (sane) {
                        double2 cs = a.beam_cs[j];
                        double ux = cth * cs.x - sth * cs.y;
                        double uy = sth * cs.x + cth * cs.y;
                        r = trace_fp64<false, COUNT>(a, ldsb, strideB, base, p0x, p0y, ux, uy, s0, amb, np);
                    }
                    if (!sane || amb < kGuard || a.force_exact == 1) {
                        r = march_exact(a, a.x[i], a.y[i], a.th[i] + (double)a.beam_angle[j]);
                        ++cnt_exact;
                    }
                    acc += a.Ldirect ? (double)a.Ldirect[(size_t)a.obs_idx[j] * tw + r] : (double)a.Lt[(size_t)r * a.bpad + j];
                    store_step(a, i, j, r);
                    if (COUNT) cnt_probe += np;
                }
            }
        }
        acc = wave_sum(acc);          // exact: fp32 table entries, |sum| < 2^13
        if (lane == 0) {
            if (FAR) atomicAdd(&a.logw[slot], acc);      // beside what k_rays_sweep / k_rays_fix leave for this slot
            else a.logw[i] = acc;
        }
    }
    }   // slices
    if (a.counters) {
        cnt_exact = wave_sum_u64(cnt_exact);
        cnt_off = wave_sum_u64(cnt_off);
        cnt_probe = wave_sum_u64(cnt_probe);
        cnt_l2 = wave_sum_u64(cnt_l2);
        if (lane == 0) {
            if (cnt_exact) atomicAdd(&a.counters[0], cnt_exact);
            if (cnt_off) atomicAdd(&a.counters[1], cnt_off);
            if (cnt_probe) atomicAdd(&a.counters[2], cnt_probe);
            if (cnt_l2) atomicAdd(&a.counters[3], cnt_l2);
        }
    }
}

// raw LDS offset of the windows of k_rays_sweep (and of the legacy k_rays_quad / k_rays_cell): they follow 16 bytes of static LDS
constexpr int kQLdsBase = 16;            // the window follows 16 bytes of static LDS (the work-item slot)

// beams of particle i in quadrant q: range [ja, jb) and, when the first quadrant is visited twice, [ja2, B)
__device__ __forceinline__ void quad_ranges(short4 qr, int q, int B, int &ja, int &jb, int &ja2)
{
    const int st1 = qr.x & 0x3FFF, st2 = qr.y, st3 = qr.z, st4 = qr.w, q0 = ((int)qr.x >> 14) & 3;
    const int r_idx = (q - q0) & 3;
    ja = r_idx == 0 ? 0 : (r_idx == 1 ? st1 : (r_idx == 2 ? st2 : st3));
    jb = r_idx == 0 ? st1 : (r_idx == 1 ? st2 : (r_idx == 2 ? st3 : st4));
    ja2 = r_idx == 0 ? st4 : B;
}


// ---- wedge fields (mcl_wedge.h) built on the device at set_map ------------------------------------------------
// The copy of wedge field k that k_rays_sweep<.., GLOBAL> probes in place (mcl_rays_sweep.h): (Hp + 4) rows of `pitch` bytes holding
// the padded grid MIRRORED in the axes along which the wedge's rays run backwards (sxp / syp = 0), at offset (2, 2); everything
// else -- the two-cell ring around it and the row padding -- stop.  Every ray of the wedge runs towards +x, +y in this frame.
// The tail of stop rows behind the field is left by the memset of the allocation (mcl_set_map).
__global__ __launch_bounds__(256) void k_ring_field(const uint8_t *__restrict__ src, int Wp, int Hp, int Wps, int pitch, int sxp, int syp,
                                                   uint8_t *__restrict__ dst)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;          // destination column / row
    if (x >= pitch) return;
    const int mx = x - 2, my = y - 2;                                             // cell of the mirrored padded grid
    uint8_t v = 0xFF;
    if (mx >= 0 && mx < Wp && my >= 0 && my < Hp) v = src[(size_t)(syp ? my : Hp - 1 - my) * Wps + (sxp ? mx : Wp - 1 - mx)];
    dst[(size_t)y * pitch + x] = v;
}

// one thread per row of the padded grid: next / previous stop cell in the row (dist == 0 marks a stop)
__global__ void k_row_tables(const uint8_t *__restrict__ dist, int Wp, int Hp, int Wps, int32_t *__restrict__ nxt, int32_t *__restrict__ prv)
{
    const int y = blockIdx.x * blockDim.x + threadIdx.x;
    if (y >= Hp) return;
    const uint8_t *row = dist + (size_t)y * Wps;
    int last = -1;
    for (int x = 0; x < Wp; ++x) { if (row[x] == 0) last = x; prv[(size_t)y * Wp + x] = last; }
    int next = Wp;
    for (int x = Wp - 1; x >= 0; --x) { if (row[x] == 0) next = x; nxt[(size_t)y * Wp + x] = next; }
}

__global__ __launch_bounds__(256) void k_wedge_field(const int32_t *__restrict__ nxt, const int32_t *__restrict__ prv, int Wp, int Hp, int Wps,
                                                    const WedgeRow *__restrict__ rows, uint8_t *__restrict__ out)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= Wp || y >= Hp) return;
    // stored in k_rays_cell's LDS encoding: stop -> 0xFF (-1 as a signed byte), skips capped at 127
    const int v = wedge_skip_cell(nxt, prv, Wp, Hp, x, y, rows);
    out[(size_t)y * Wps + x] = (uint8_t)(v == 0 ? 255 : (v > 127 ? 127 : v));
}

// ---- cell sort: particles ordered by (32x32 tile, cell in tile, heading) ---------------------------------
//
// k_rays_cell gives every LANE one particle and walks that particle's beams; the 64 lanes of a wave then trace
// near-parallel rays (same beam slot of the same direction range) and, once the particles are ordered by grid cell
// and heading, from almost the same origin: their trip counts are nearly equal and the wave no longer waits for
// its slowest lane.  The order is a counting sort: bucket = (tile-major cell id << theta_bits) | quantised
// heading over the bounding box of the particle set, as fine as fits kSortKeySpace.
//
// The histogram is kept once per XCD, hist[xcd][bucket]: a workgroup only touches the copy of the XCD it runs on,
// so its atomics can be workgroup-scope and execute in that XCD's L2 instead of at the memory side (device-scope
// atomics from eight XCDs on the same hot counters took 0.5 ms per update, these take 0.1).  The value an atomic
// returns is the particle's rank inside (bucket, xcd); the scan orders the counters bucket-major, xcd-minor.  The
// rank order varies from run to run — harmless: the order only decides which rays share a wave, every log-weight
// is an exact sum.
// (22 key bits and half cells.  24 bits / quarter cells and 26 bits / eighths were built and measured at the end of round 5: the ray
//  kernel of the bench workload issues the same number of instructions launch for launch with all three -- the density gate of
//  sort_layout, not the key space, ends the split there, so the order is the same -- and yet runs 1 % faster or slower: with where
//  the buffers lie (the histogram allocation is 128 MB / 512 MB / 2 GB), which is all that is left to differ:
//  profiles/r05_experiments/sort_key_bits.txt.  Not taken.)
#ifndef MCL_SORT_KEY_LOG2
#define MCL_SORT_KEY_LOG2 22
#endif
constexpr int kSortKeyLog2 = MCL_SORT_KEY_LOG2;
constexpr uint32_t kSortKeySpace = 1u << kSortKeyLog2;
#ifndef MCL_SORT_MAX_SUB
#define MCL_SORT_MAX_SUB 1
#endif
constexpr int kSortXcds = 8;                        // copies of the histogram (XCC_ID & 7)
constexpr uint32_t kSortBuckets = kSortKeySpace * kSortXcds;
constexpr int kHistTile = 4096;                     // entries per workgroup of the bucket scan

__device__ __forceinline__ int cell_of(double g, int hi)
{
    // padded cell coordinate of a pixel coordinate, clamped into the padded grid (NaN -> 0)
    int c = (g > -1.0) ? ((g < 1e9) ? (int)floor(g) + 1 : hi) : 0;
    return c < 0 ? 0 : (c > hi ? hi : c);
}

// bbox[0..3] = min cx, min cy, max cx, max cy over every `stride`-th particle (initialised to +big / -big by the
// host).  A sample is enough: sort_key clamps cells into the box, so a particle outside it merely lands in an
// edge bucket (the order is a performance matter only).
// bbox as PrepClear leaves it: empty box, no tiles numbered, [6] = the window play sort_layout works with
__global__ void k_bbox_init(int *__restrict__ bbox, int play)
{
    const int i = threadIdx.x;
    if (i < 7) bbox[i] = i < 2 ? 0x7fffffff : (i < 4 ? (int)0x80000000 : (i == 6 ? play : 0));
}

__global__ __launch_bounds__(256) void k_cell_bbox(const double4 *__restrict__ pc, int64_t n, int stride, int Wp, int Hp, int *__restrict__ bbox,
                                                  int *__restrict__ tilemark = nullptr, int ntx_abs = 0)
{
    int x0 = 0x7fffffff, y0 = 0x7fffffff, x1 = -1, y1 = -1;
    const int64_t ns = (n + stride - 1) / stride;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < ns; k += (int64_t)gridDim.x * blockDim.x) {
        const double4 c = pc[k * stride];
        const int cx = cell_of(c.z * kSortSub, Wp * kSortSub - 1), cy = cell_of(c.w * kSortSub, Hp * kSortSub - 1);
        x0 = min(x0, cx); y0 = min(y0, cy); x1 = max(x1, cx); y1 = max(y1, cy);
        if (tilemark) tilemark[(cy >> 5) * ntx_abs + (cx >> 5)] = 1;      // occupied tiles of the map (k_tile_compact)
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        x0 = min(x0, __shfl_xor(x0, o, 64)); y0 = min(y0, __shfl_xor(y0, o, 64));
        x1 = max(x1, __shfl_xor(x1, o, 64)); y1 = max(y1, __shfl_xor(y1, o, 64));
    }
    // one set of atomics per workgroup: same-address atomics serialise at ~12 ns each
    __shared__ int red[4][4];
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[w][0] = x0; red[w][1] = y0; red[w][2] = x1; red[w][3] = y1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 4; ++k) { x0 = min(x0, red[k][0]); y0 = min(y0, red[k][1]); x1 = max(x1, red[k][2]); y1 = max(y1, red[k][3]); }
        if (x1 >= 0) { atomicMin(&bbox[0], x0); atomicMin(&bbox[1], y0); atomicMax(&bbox[2], x1); atomicMax(&bbox[3], y1); }
    }
}

// Occupied tiles.  The key of a particle starts with its 32 x 32-cell tile; numbering only the tiles that hold (sampled)
// particles instead of every tile of the bounding box gives the bits of the empty ones -- most of them when the set sits on
// a few far-apart clusters -- back to cells and heading.  bbox[4] = number of occupied tiles T, bbox[5] = 1 when the
// numbering is in use (a bounding box of at most kSortMaxTiles tiles); tilemap[t] = compact id, T for a tile no sampled
// particle fell into (its particles share one overflow tile: the order is a performance matter only).
constexpr int kSortMaxTiles = 8192;
// tilemark[t] (set by k_cell_bbox for the tiles of the MAP its sampled particles fall into) -> tilemap[t] = compact id,
// T for an unmarked tile; the marks are zeroed for the next update.  One workgroup, ntiles_abs <= kSortMaxTiles.
__global__ __launch_bounds__(1024) void k_tile_compact(int *__restrict__ bbox, int *__restrict__ tilemark, int *__restrict__ tilemap, int ntiles_abs)
{
    __shared__ int ws[16];
    __shared__ int carry_sh;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_sh = 0;
    __syncthreads();
    for (int t0 = 0; t0 < ntiles_abs; t0 += 1024) {
        const int t = t0 + (int)threadIdx.x;
        const int m = (t < ntiles_abs && tilemark[t] != 0) ? 1 : 0;
        int inc = m;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(inc, o, 64); if (lane >= o) inc += v; }
        if (lane == 63) ws[wv] = inc;
        __syncthreads();
        int id = carry_sh + inc - m;
        for (int k = 0; k < wv; ++k) id += ws[k];
        if (t < ntiles_abs) { tilemap[t] = m ? id : -1; tilemark[t] = 0; }
        __syncthreads();
        if (threadIdx.x == 1023) carry_sh = id + m;
        __syncthreads();
    }
    const int T = carry_sh;
    for (int t = threadIdx.x; t < ntiles_abs; t += 1024)
        if (tilemap[t] < 0) tilemap[t] = T;           // (written by this thread in the loop above: t = t0 + threadIdx.x)
    if (threadIdx.x == 0) { bbox[4] = T; bbox[5] = 1; }
}

#ifndef MCL_SORT_SUB_GATE
#define MCL_SORT_SUB_GATE 32.0
#endif
constexpr double kSortSubGate = MCL_SORT_SUB_GATE;  // particles a bucket must hold (an eighth of its heading bins taken as occupied) to be split in four
constexpr int kSortMaxSub = MCL_SORT_MAX_SUB;        // sub-cell bits per axis a dense set may get (0: none)
// How the key bits are split for this update's set: a pure function of the bounding box / occupied-tile count and n, so that
// every particle -- and k_unit_table, which needs to know where a tile's keys start -- derives the same layout.
//   key = ((((tile << inner | iy) << inner | ix) << ss | sy) << ss | sx) << tb | heading bin
struct SortLayout {
    int cs, tb, inner, ss;
    uint32_t ntx;
    int tx0, ty0;
    bool compact;
    bool nocut;                       // bbox[6] < 0 (MCL_NO_BUCKET_CUTS): the plain grid of units, sparse sets by whole tiles
    bool sparse;                      // ordered by whole buckets of 2^cs x 2^cs cells (a spread cloud): units are cut at their borders
    uint64_t ntiles;
    __device__ __forceinline__ int tile_shift() const { return 2 * inner + 2 * ss + tb; }
    // Units are cut where the sorted order changes GROUP: a sparse set's bucket, a dense set's 32 x 32 tile (a set that sits on
    // a few far-apart clusters -- the steady state of a global re-localisation -- otherwise has units that end in one cluster
    // and begin in the next, whose minority side no window reaches: 0.85 ms of far pass per update at 4M on ~30 clusters).
    __device__ __forceinline__ int cut_shift() const { return sparse ? 2 * ss + tb : tile_shift(); }                 // key bits below the group
    __device__ __forceinline__ uint64_t cut_groups() const { return sparse ? ntiles << (2 * inner) : ntiles; }       // groups of the key space
    // (k_sort_gather notes the borders, k_unit_table uses them: one rule for both)
    __device__ __forceinline__ bool cuts() const;
};
constexpr int kSwMaxCuts = 65536;     // buckets up to which a sparse set's units are cut at bucket borders (k_unit_table)
__device__ __forceinline__ bool SortLayout::cuts() const
{
    return compact && !nocut && cut_groups() <= (uint64_t)kSwMaxCuts && (cut_groups() << cut_shift()) <= kSortKeySpace;
}
__device__ __forceinline__ SortLayout sort_layout(const int *__restrict__ bbox, int64_t n)
{
    SortLayout L;
    L.tx0 = bbox[0] >> 5; L.ty0 = bbox[1] >> 5;
    L.ntx = (uint32_t)((bbox[2] >> 5) - L.tx0 + 1);
    const uint32_t nty = (uint32_t)((bbox[3] >> 5) - L.ty0 + 1);
    L.compact = bbox[5] != 0;
    L.ntiles = L.compact ? (uint64_t)bbox[4] + 1u : (uint64_t)L.ntx * nty;       // + 1: the overflow tile
    // cells the set can be taken to live on: its bounding box, or its occupied tiles if that is less
    double cells_est = (double)(bbox[2] - bbox[0] + 1) * (double)(bbox[3] - bbox[1] + 1);
    if (L.compact && (double)bbox[4] * 1024.0 < cells_est) cells_est = (double)bbox[4] * 1024.0;
    // Heading first: the lanes of a wave must agree on which wedge their beams are in, so six heading bits (5.6 degrees)
    // are kept while the in-tile resolution is coarsened to make room (a particle set spread over a large bounding box
    // -- several far-apart clusters -- otherwise spends the whole key space on empty cells: 80 % slower ray stage);
    // only a bounding box of more than 2^14 tiles gives heading bits up.  What is left over goes to the heading.
    // A SPARSE set (fewer than 8 particles per cell of the bounding box: the uniform cloud of global localisation, or a
    // few far-apart clusters) is ordered by whole 32 x 32 tiles: 1024 consecutive particles cover hundreds of cells
    // whatever the bucket size, and with one bucket per tile the 64 particles of a wave are neighbours in heading
    // (first update of the global regime: ray kernel 13.8 -> 11.4 ms).  A dense set keeps single cells: there the
    // compactness of a unit is what the windows and the probe loop live on (coarser buckets cost 4-30 %).
    int cs = 0, tb = 6;
    L.nocut = bbox[6] < 0;
    L.sparse = cells_est > 0.0 && (double)n < 8.0 * cells_est;
    if (L.sparse) {
        cs = 5;
        // ... of the size the play of a ray window can hold: a unit is one bucket's particles, and a unit wider than the play
        // loses its particles to the far pass (the levine stand-in's 239-px range leaves 12 cells: whole tiles put 3.5M of the
        // 4M particles of a uniform cloud there, 35 ms).  Smaller buckets only while one still holds a couple of chunks.
        const int play = bbox[6];
        while (play > 0 && cs > 1 && (1 << cs) >= play - 2 && (double)n * (double)(1 << (2 * cs - 2)) >= 128.0 * cells_est) --cs;
    }
    while (cs < 5 && ((L.ntiles << (10 - 2 * cs + tb)) > kSortKeySpace)) ++cs;
    while (tb > 0 && ((L.ntiles << (10 - 2 * cs + tb)) > kSortKeySpace)) --tb;
    const int inner = 5 - cs;                             // log2 of the bucket grid inside one tile
    const uint64_t ncell = L.ntiles << (2 * inner);
    while (tb < 8 && (ncell << (tb + 1)) <= kSortKeySpace) ++tb;
    // Bits left over once the cells have their full resolution and the heading its eight bits go to the position inside
    // the cell (half cells, then quarter cells): a collapsed set is thousands of particles per cell, and rays that start
    // within half a cell of each other keep company longer (levine stand-in: ray kernel -10 %, Spielberg -3 %).
    int ss = 0;
    if (cs == 0 && tb == 8) {
        // ... as long as a bucket still holds a few waves' worth: particles per cell of the bounding box, an eighth of the
        // heading bins taken as occupied (262 144 particles on 25 cells are better off without: 43 per bucket)
        double per_bucket = cells_est > 0.0 ? (double)n / cells_est / 32.0 : 0.0;
        while (ss < kSortMaxSub && (ncell << (tb + 2 * (ss + 1))) <= kSortKeySpace && per_bucket >= kSortSubGate) { ++ss; per_bucket *= 0.25; }
    }
    L.cs = cs; L.tb = tb; L.inner = inner; L.ss = ss;
    return L;
}

__device__ __forceinline__ uint32_t sort_key(const int *__restrict__ bbox, const int *__restrict__ tilemap, int ntx_abs, int cx, int cy, double th,
                                         int64_t n, double fx, double fy)
{
    cx = cx < bbox[0] ? bbox[0] : (cx > bbox[2] ? bbox[2] : cx);
    cy = cy < bbox[1] ? bbox[1] : (cy > bbox[3] ? bbox[3] : cy);
    const SortLayout L = sort_layout(bbox, n);
    const int cs = L.cs, tb = L.tb, inner = L.inner, ss = L.ss;
    uint32_t tile = (uint32_t)((cy >> 5) - L.ty0) * L.ntx + (uint32_t)((cx >> 5) - L.tx0);
    if (L.compact) tile = (uint32_t)tilemap[(cy >> 5) * ntx_abs + (cx >> 5)];
    const uint32_t ix = (uint32_t)(cx & 31) >> cs, iy = (uint32_t)(cy & 31) >> cs;
    uint64_t key = ((((uint64_t)tile << inner) | iy) << inner) | ix;
    if (ss > 0) {
        const uint32_t sx = (fx >= 0.0 && fx < 1.0) ? (uint32_t)(fx * (double)(1 << ss)) : 0u;
        const uint32_t sy = (fy >= 0.0 && fy < 1.0) ? (uint32_t)(fy * (double)(1 << ss)) : 0u;
        key = (((key << ss) | sy) << ss) | sx;
    }
    double f = th * 0.15915494309189533577;               // heading as a fraction of a turn
    f -= floor(f);
    uint32_t tq = (f >= 0.0 && f < 1.0) ? (uint32_t)(f * (double)(1u << tb)) : 0u;
    if (tq >> tb) tq = (1u << tb) - 1u;
    key = (key << tb) | tq;
    return key < kSortKeySpace ? (uint32_t)key : kSortKeySpace - 1u;    // a bounding box beyond the key space: unsorted tail
}

__global__ __launch_bounds__(256) void k_sort_hist(const double4 *__restrict__ pc, const double *__restrict__ th, int64_t n, int Wp, int Hp,
                                                  const int *__restrict__ bbox, uint32_t *__restrict__ hist,
                                                  uint32_t *__restrict__ key_out, uint32_t *__restrict__ rank_out, uint32_t *__restrict__ tile_used,
                                                  const int *__restrict__ tilemap, int ntx_abs)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t xcd = (uint32_t)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & (kSortXcds - 1);   // HW_REG_XCC_ID[3:0]
    const double4 c = pc[i];
    const uint32_t key = sort_key(bbox, tilemap, ntx_abs, cell_of(c.z * kSortSub, Wp * kSortSub - 1), cell_of(c.w * kSortSub, Hp * kSortSub - 1), th[i], n,
                                  c.z * kSortSub - floor(c.z * kSortSub), c.w * kSortSub - floor(c.w * kSortSub));
    key_out[i] = key;
    // this XCD's private copy: workgroup scope keeps the read-modify-write in the local L2
    const uint32_t r = __hip_atomic_fetch_add(&hist[(size_t)xcd * kSortKeySpace + key], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    rank_out[i] = (xcd << 28) | r;
    // the first arrival in a counter marks its tile of kHistTile buckets as used: the scan and the clearing of the
    // histogram then touch the used tiles only (a collapsed set uses a few dozen of the 1024)
    if (r == 0u) tile_used[key / kHistTile] = 1u;
}

// totals of kHistTile buckets (over all XCD copies) per workgroup: part[t], zero for a tile nobody fell into
__global__ __launch_bounds__(256) void k_hist_partials(const uint32_t *__restrict__ hist, uint32_t *__restrict__ part, const uint32_t *__restrict__ tile_used)
{
    __shared__ uint32_t ws[4];
    if (!tile_used[blockIdx.x]) {                 // wave-uniform
        if (threadIdx.x == 0) part[blockIdx.x] = 0u;
        return;
    }
    uint32_t s = 0;
    for (int x = 0; x < kSortXcds; ++x) {
        const uint4 *p = reinterpret_cast<const uint4 *>(hist + (size_t)x * kSortKeySpace + (size_t)blockIdx.x * kHistTile) + threadIdx.x * 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) { uint4 v = p[k]; s += v.x + v.y + v.z + v.w; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

// in place: counter (bucket b, xcd x) -> first slot of that group in the order bucket-major, xcd-minor
__global__ __launch_bounds__(256) void k_hist_final(uint32_t *__restrict__ hist, const uint32_t *__restrict__ part, const uint32_t *__restrict__ tile_used)
{
    __shared__ uint32_t ws[4], ps[4];
    if (!tile_used[blockIdx.x]) return;           // no particle in these buckets: nobody reads their offsets
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    // slots in front of this tile: the totals of the tiles before it, summed here (at most 1023 words out of L2, by the used
    // tiles only) instead of scanned by a one-workgroup launch in between
    uint32_t pre = 0;
    for (int t = (int)threadIdx.x; t < (int)blockIdx.x; t += 256) pre += part[t];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) pre += __shfl_xor(pre, o, 64);
    if (lane == 0) ps[w] = pre;
    const size_t b0 = (size_t)blockIdx.x * kHistTile + (size_t)threadIdx.x * 16;    // 16 consecutive buckets per thread
    uint4 v[kSortXcds][4];
    uint32_t s = 0;
#pragma unroll
    for (int x = 0; x < kSortXcds; ++x) {
        const uint4 *p = reinterpret_cast<const uint4 *>(hist + (size_t)x * kSortKeySpace + b0);
#pragma unroll
        for (int k = 0; k < 4; ++k) { v[x][k] = p[k]; s += v[x][k].x + v[x][k].y + v[x][k].z + v[x][k].w; }
    }
    uint32_t inc = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { uint32_t t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
    if (lane == 63) ws[w] = inc;
    __syncthreads();
    uint32_t run = ps[0] + ps[1] + ps[2] + ps[3] + inc - s;
    for (int k = 0; k < w; ++k) run += ws[k];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        // bucket 4k..4k+3 of this thread: the xcd copies in turn
#pragma unroll
        for (int x = 0; x < kSortXcds; ++x) { uint32_t c = v[x][k].x; v[x][k].x = run; run += c; }
#pragma unroll
        for (int x = 0; x < kSortXcds; ++x) { uint32_t c = v[x][k].y; v[x][k].y = run; run += c; }
#pragma unroll
        for (int x = 0; x < kSortXcds; ++x) { uint32_t c = v[x][k].z; v[x][k].z = run; run += c; }
#pragma unroll
        for (int x = 0; x < kSortXcds; ++x) { uint32_t c = v[x][k].w; v[x][k].w = run; run += c; }
    }
#pragma unroll
    for (int x = 0; x < kSortXcds; ++x) {
        uint4 *p = reinterpret_cast<uint4 *>(hist + (size_t)x * kSortKeySpace + b0);
#pragma unroll
        for (int k = 0; k < 4; ++k) p[k] = v[x][k];
    }
}

__global__ __launch_bounds__(256) void k_sort_scatter(const double4 *__restrict__ pc, const double *__restrict__ th, int64_t n,
                                                     const uint32_t *__restrict__ key, const uint32_t *__restrict__ rank,
                                                     const uint32_t *__restrict__ start, double4 *__restrict__ pcs,
                                                     double *__restrict__ ths, uint32_t *__restrict__ perm)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t rk = rank[i];
    const uint32_t slot = start[(size_t)(rk >> 28) * kSortKeySpace + key[i]] + (rk & 0x0FFFFFFFu);
    if (slot >= (uint64_t)n) return;                       // cannot happen (the counts sum to n); never write out of bounds
    pcs[slot] = pc[i];
    ths[slot] = th[i];
    perm[slot] = (uint32_t)i;
}

// ---- the same ordering through a radix sort of (key, index) pairs (MCL_SORT=radix): keys only, no atomics; the sort itself is
//      rocPRIM's (a plain library sort, like a library GEMM would be), the gather puts the records in sorted order ----
__global__ __launch_bounds__(256) void k_sort_keys(const double4 *__restrict__ pc, const double *__restrict__ th, int64_t n, int Wp, int Hp,
                                                  const int *__restrict__ bbox, uint32_t *__restrict__ key_out, uint32_t *__restrict__ val_out,
                                                  const int *__restrict__ tilemap, int ntx_abs)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double4 c = pc[i];
    key_out[i] = sort_key(bbox, tilemap, ntx_abs, cell_of(c.z * kSortSub, Wp * kSortSub - 1), cell_of(c.w * kSortSub, Hp * kSortSub - 1), th[i], n,
                          c.z * kSortSub - floor(c.z * kSortSub), c.w * kSortSub - floor(c.w * kSortSub));
    val_out[i] = (uint32_t)i;
}
// cut_start[g] + 1 / cut_end[g]: first slot + 1 and end of bucket g in the sorted order (0 / anything: nobody there), written where
// the sorted keys change bucket -- what k_unit_table needs to cut a sparse set's units at bucket borders (it zeroes the
// entries it reads).  sorted_keys null: no borders wanted.
__global__ __launch_bounds__(256) void k_sort_gather(const double4 *__restrict__ pc, const double *__restrict__ th, int64_t n,
                                                    const uint32_t *__restrict__ order, double4 *__restrict__ pcs, double *__restrict__ ths,
                                                    uint32_t *__restrict__ perm, const uint32_t *__restrict__ sorted_keys = nullptr,
                                                    const int *__restrict__ bbox = nullptr, uint32_t *__restrict__ cut_start = nullptr,
                                                    uint32_t *__restrict__ cut_end = nullptr)
{
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const uint32_t i = order[s];
    pcs[s] = pc[i];
    ths[s] = th[i];
    perm[s] = i;
    if (sorted_keys) {
        const SortLayout L = sort_layout(bbox, n);
        if (L.cuts()) {
            const int sh = L.cut_shift();
            const uint32_t g = sorted_keys[s] >> sh;
            const uint32_t gp = s > 0 ? sorted_keys[s - 1] >> sh : 0xFFFFFFFFu;
            if (g != gp && g < (uint32_t)kSwMaxCuts) {
                cut_start[g] = (uint32_t)s + 1u;
                if (s > 0 && gp < (uint32_t)kSwMaxCuts) cut_end[gp] = (uint32_t)s;
            }
            if (s == n - 1 && g < (uint32_t)kSwMaxCuts) cut_end[g] = (uint32_t)n;
        }
    }
}

// after the scatter: the used tiles of the histogram (all XCD copies) and their marks back to zero, so that the next
// update starts from an all-zero histogram without a pass over its 128 MB
__global__ __launch_bounds__(256) void k_hist_clear(uint32_t *__restrict__ hist, uint32_t *__restrict__ tile_used)
{
    if (!tile_used[blockIdx.x]) return;
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
    for (int x = 0; x < kSortXcds; ++x) {
        uint4 *p = reinterpret_cast<uint4 *>(hist + (size_t)x * kSortKeySpace + (size_t)blockIdx.x * kHistTile) + threadIdx.x * 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) p[k] = z;
    }
    __syncthreads();
    if (threadIdx.x == 0) tile_used[blockIdx.x] = 0u;
}

// mean pixel position of every slice of `per` sorted particles (non-finite positions excluded): k_rays_cell centres
// the sixteen wedge windows of a slice on it
__global__ __launch_bounds__(256) void k_slice_means(const double4 *__restrict__ pcs, int64_t n, int64_t per, double2 *__restrict__ out)
{
    __shared__ double sm[4][3];
    const int64_t p_begin = (int64_t)blockIdx.x * per;
    const int64_t p_end = (p_begin + per < n) ? p_begin + per : n;
    double sx = 0.0, sy = 0.0, cnt = 0.0;
    for (int64_t s = p_begin + threadIdx.x; s < p_end; s += blockDim.x) {
        const double4 c = pcs[s];
        if (c.z == c.z && c.w == c.w && fabs(c.z) < 1e9 && fabs(c.w) < 1e9) { sx += c.z; sy += c.w; cnt += 1.0; }
    }
    sx = wave_sum(sx); sy = wave_sum(sy); cnt = wave_sum(cnt);
    if ((threadIdx.x & 63) == 0) { sm[threadIdx.x >> 6][0] = sx; sm[threadIdx.x >> 6][1] = sy; sm[threadIdx.x >> 6][2] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        sx = sm[0][0] + sm[1][0] + sm[2][0] + sm[3][0];
        sy = sm[0][1] + sm[1][1] + sm[2][1] + sm[3][1];
        cnt = sm[0][2] + sm[1][2] + sm[2][2] + sm[3][2];
        out[blockIdx.x] = cnt > 0.0 ? make_double2(sx / cnt, sy / cnt) : make_double2(0.0, 0.0);
    }
}

// first beam j with beam_wedge(th, angle[j]) >= m (beam angles increase, so the wedge index is monotone in j).
// For the usual evenly spaced scan the answer is within a beam or two of (m * 2pi/K - th - a0) / increment: the
// bisection starts from a five-beam bracket around that guess when the bracket holds and from [0, B] otherwise.
// GRID: the scan is evenly spaced to within 2e-6 rad (mcl_set_beam_angles: the REC condition, a0 / inv_inc the grid through the
// first and the last beam).  The guess is then the answer unless x lies within 4e-6 rad (twice the bound on a beam's offset; ten
// orders above the rounding of this arithmetic) of a whole beam -- no load, no second evaluation of beam_wedge in all but one
// visit in a thousand at a quarter of a degree per beam.
template <bool GRID = false>
__device__ __forceinline__ int first_beam_in_wedge(double th, const float *__restrict__ angle, int B, int m, double a0, double inv_inc)
{
    int lo = 0, hi = B;
    const double x = ((double)m * (6.283185307179586476925286766559 / kWedges) - th - a0) * inv_inc;
    if (x > -4.0 && x < (double)B + 4.0) {
        const int jg = min(max((int)ceil(x), 0), B);
        if (GRID && fabs(x - rint(x)) > 4e-6 * inv_inc + 1e-6) return jg;
        const bool below = jg == 0 || beam_wedge(th, angle[jg - 1]) < m;      // beam jg-1 is still before wedge m
        const bool at = jg == B || beam_wedge(th, angle[jg]) >= m;           // beam jg is in wedge m or later
        if (below && at) return jg;                                           // the usual case: two loads
        // off by a beam or two (float rounding of the angles): a five-beam bracket when it holds, else [0, B]
        const int l = max(jg - 2, 0), h = min(jg + 2, B);
        if ((l == 0 || beam_wedge(th, angle[l - 1]) < m) && (h == B || beam_wedge(th, angle[h]) >= m)) { lo = l; hi = h; }
    }
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (beam_wedge(th, angle[mid]) >= m) hi = mid; else lo = mid + 1;
    }
    return lo;
}

#ifdef MCL_LEGACY_RAY_KERNELS
}  // namespace mcl
#include "mcl_rays_legacy.h"        // k_rays_quad, k_rays_cell: the predecessors of k_rays_sweep (a build flag: AUTO never picks them)
namespace mcl {
#endif

// Levels 2 and 3 for the rays k_rays_quad could not decide: one thread per listed ray, fp64 positions on the
// global byte field (padded global coordinates shifted by 2^18), then the literal march if still ambiguous.
template <bool COUNT>
__global__ __launch_bounds__(256) void k_rays_fix(RayArgs a)
{
    unsigned long long cnt_exact = 0, cnt_probe = 0, cnt_l2 = 0;
    // gridDim.x = split * fix_segments: `split` workgroups share one segment (the kernel is bound by the latency of
    // its dependent global loads, so it wants every CU full of waves rather than one workgroup per segment)
    const int split = max(1, (int)gridDim.x / a.fix_segments);
    const int part = blockIdx.x % split;
    for (int seg = blockIdx.x / split; seg < a.fix_segments; seg += gridDim.x / split) {
    // the counters were updated by memory-side atomics: read them the same way (a cached copy may be stale)
    // (k_rays_quad / k_rays_cell append with memory-side atomics and are read back the same way; k_rays_sweep's lists and counts
    //  are plain stores of an earlier kernel -- slot_space -- which plain loads see)
    unsigned long long n = 0;
    if (threadIdx.x == 0) n = a.slot_space ? a.fix_count[(size_t)seg * 8] : atomicAdd(&a.fix_count[(size_t)seg * 8], 0ull);
    n = __shfl((long long)n, 0, 64);
    {
        __shared__ unsigned long long n_sh;
        if (threadIdx.x == 0) n_sh = n;
        __syncthreads();
        n = n_sh;
        __syncthreads();
    }
    if (n > a.fix_cap) n = 0;                              // overflow (or a ray kernel that stood down): the host re-runs the stage with
                                                           // k_rays_skip and discards this one, so nothing of the segment is traced
    cnt_l2 += (threadIdx.x == 0 && part == 0) ? n : 0;
    const unsigned long long *list = a.fix_list + (size_t)seg * a.fix_cap;
    for (unsigned long long k = (unsigned long long)part * blockDim.x + threadIdx.x; k < n; k += (unsigned long long)blockDim.x * split) {
        const unsigned long long e = a.slot_space ? list[k] : atomicAdd(const_cast<unsigned long long *>(&list[k]), 0ull);
        const int64_t p = (int64_t)(e >> 16);                     // particle index, or sorted slot (slot_space)
        const int j = (int)(e & 0xFFFF);
        const double4 pci = a.slot_space ? a.pcs[p] : a.pc[p];
        // k_rays_sweep's rays are re-traced on the field of their own wedge (half the probes of the isotropic field)
        const uint8_t *wfield = nullptr;
        if (a.slot_space && a.distw) {
            const double hth = a.ths[p];
            if (hth == hth && fabs(hth) < 1e6) wfield = a.distw + (size_t)(beam_wedge(hth, a.beam_angle[j]) & (kWedges - 1)) * a.distw_stride;
        }
        const double2 cs = a.beam_cs[j];
        const double ux = pci.x * cs.x - pci.y * cs.y, uy = pci.y * cs.x + pci.x * cs.y;
        const bool sane = (pci.z > -200000.0) && (pci.z < 200000.0) && (pci.w > -200000.0) && (pci.w < 200000.0);
        const double p0x = (pci.z + 1.0 + 262144.0) + kMagic, p0y = (pci.w + 1.0 + 262144.0) + kMagic;
        const int base = kCellBase + 262144;
        int r = a.P;
        uint32_t amb = 0;
        unsigned np = 0;
        if (sane) {
            const uint32_t lox = (uint32_t)__double2loint(p0x), loy = (uint32_t)__double2loint(p0y);
            const int cx = (__double2hiint(p0x) & 0xFFFFF) - base, cy = (__double2hiint(p0y) & 0xFFFFF) - base;
            amb = lox < loy ? lox : loy;
            int d = ((unsigned)cx < (unsigned)a.Wp && (unsigned)cy < (unsigned)a.Hp) ? (wfield ? wfield : a.dist)[(size_t)cy * a.Wps + cx] : 0;
            if (wfield && d == 255) d = 0;
            r = trace_fp64<false, COUNT>(a, nullptr, 0, base, p0x, p0y, ux, uy, d > 1 ? d : 1, amb, np, wfield, wfield != nullptr);
        }
        if (!sane || amb < kGuard || a.force_exact == 1) {
            // level 3, the literal march: 207 dependent steps in one thread would set the duration of this kernel, so the
            // (few) rays that need it go to k_rays_exact, which gives each of them a whole wave
            if (a.exact_list) {
                const unsigned long long slot = atomicAdd(a.exact_count, 1ull);
                if (slot < a.exact_cap) { atomicExch(&a.exact_list[slot], e); continue; }
            }
            const int64_t i = a.slot_space ? (int64_t)a.perm[p] : p;
            r = march_exact(a, a.x[i], a.y[i], a.th[i] + (double)a.beam_angle[j]);
            ++cnt_exact;
        }
        atomicAdd(&a.logw[p], (double)a.Lt[(size_t)r * a.bpad + j]);
        if (a.steps || a.steps16) store_step(a, a.slot_space ? (int64_t)a.perm[p] : p, j, r);
        if (COUNT) cnt_probe += np;
    }
    }
    if (a.counters) {
        if (cnt_exact) atomicAdd(&a.counters[0], cnt_exact);
        if (COUNT && cnt_probe) atomicAdd(&a.counters[2], cnt_probe);
        if (cnt_l2) atomicAdd(&a.counters[3], cnt_l2);
    }
}

// Level 3 for the rays k_rays_fix handed over: the literal march of cast_ray (cpp:611-650), one WAVE per ray.  Lane l
// accumulates `current += d` l+1 times exactly as the reference's single accumulator does (same additions in the same
// order, so the same bits), then the 64 lanes test 64 consecutive samples at once; the first stop wins.
template <bool COUNT>
__global__ __launch_bounds__(256) void k_rays_exact(RayArgs a)
{
    const int lane = threadIdx.x & 63;
    const unsigned long long wave_id = ((unsigned long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const unsigned long long nwaves = ((unsigned long long)gridDim.x * blockDim.x) >> 6;
    unsigned long long n = 0;
    if (lane == 0) n = atomicAdd(a.exact_count, 0ull);
    n = (unsigned long long)__shfl((long long)n, 0, 64);
    if (n > a.exact_cap) n = a.exact_cap;
    unsigned long long done = 0;
    for (unsigned long long k = wave_id; k < n; k += nwaves) {
        unsigned long long e = 0;
        if (lane == 0) e = atomicAdd(&a.exact_list[k], 0ull);
        e = (unsigned long long)__shfl((long long)e, 0, 64);
        const int64_t p = (int64_t)(e >> 16);
        const int64_t i = a.slot_space ? (int64_t)a.perm[p] : p;
        const int j = (int)(e & 0xFFFF);
        const double angle = a.th[i] + (double)a.beam_angle[j];
        const double dx = cos(angle) * a.res, dy = sin(angle) * a.res;
        double cx = a.x[i], cy = a.y[i];
        for (int t = 0; t <= lane; ++t) { cx += dx; cy += dy; }        // sample lane + 1 of the sequential accumulation
        int r = a.P;
        for (int base = 0; base < a.P; base += 64) {
            const int step = base + lane;
            bool hit = false;
            if (step < a.P) {
                const int gx = (int)((cx - a.ox) / a.res), gy = (int)((cy - a.oy) / a.res);
                hit = gx < 0 || gx >= a.W || gy < 0 || gy >= a.H || a.grid[(size_t)gy * a.W + gx] > 50;
            }
            const unsigned long long m = __ballot(hit);
            if (m) { r = base + (__ffsll((long long)m) - 1); break; }
            for (int t = 0; t < 64; ++t) { cx += dx; cy += dy; }        // 64 samples further
        }
        if (lane == 0) {
            atomicAdd(&a.logw[p], (double)a.Lt[(size_t)r * a.bpad + j]);
            if (a.steps || a.steps16) store_step(a, i, j, r);
            ++done;
        }
    }
    if (a.counters && lane == 0 && done) atomicAdd(&a.counters[0], done);
}

// acc (fp64 atomics of the ray kernels of this launch sequence) -> plain array for the rest of the pipeline.  Plain loads
// in a LATER kernel see the result of memory-side atomics (tools/ubench/coherence.hip, profiles/r02_coherence.txt: 0 stale
// entries in 8 x 84M reads, whether the zeroing was a plain or a write-through store); round 1 read these back with atomics
// because of wrong sums that went away with the stream fix of the same day (memsets on the null stream racing the engine's
// stream), not because of the cache hierarchy.
__global__ void k_gather_logw(const double *__restrict__ acc, int64_t n, double *__restrict__ out)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = acc[i];
}

// overflow flag of the fix-up list: *over = number of segments whose append count exceeded the capacity
__global__ void k_fix_overflow(const unsigned long long *__restrict__ counts, int nseg, unsigned long long cap, unsigned long long *over)
{
    __shared__ unsigned int s;
    if (threadIdx.x == 0) s = 0;
    __syncthreads();
    unsigned int c = 0;
    for (int k = threadIdx.x; k < nseg; k += blockDim.x)
        c += atomicAdd(const_cast<unsigned long long *>(&counts[(size_t)k * 8]), 0ull) > cap ? 1u : 0u;
    if (c) atomicAdd(&s, c);
    __syncthreads();
    if (threadIdx.x == 0) *over = s;
}

// The flagged slots in ascending order (k_rays_sweep appends them in arrival order): count per 2048 slots, exclusive scan of the
// counts by one workgroup, scatter.  All three stand down below kFarWindowedMin flagged slots (k_rays_far handles those).
constexpr int kFarTile = 2048;
__global__ __launch_bounds__(256) void k_far_count(const uint32_t *__restrict__ flags32, int64_t n, const unsigned long long *__restrict__ far_count,
                                                  uint32_t *__restrict__ cnt)
{
    if (*far_count < kFarWindowedMin) return;
    __shared__ uint32_t ws[4];
    const int64_t base = (int64_t)blockIdx.x * kFarTile + (int64_t)threadIdx.x * 8;
    uint32_t c = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) c += (base + k < n && flags32[base + k] != 0u) ? 1u : 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) cnt[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}
__global__ __launch_bounds__(1024) void k_far_spine(uint32_t *__restrict__ cnt, int nb, const unsigned long long *__restrict__ far_count)
{
    if (*far_count < kFarWindowedMin) return;
    __shared__ uint32_t ws[16];
    __shared__ uint32_t carry_sh;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_sh = 0u;
    __syncthreads();
    for (int b0 = 0; b0 < nb; b0 += 1024) {
        const int i = b0 + (int)threadIdx.x;
        const uint32_t v = i < nb ? cnt[i] : 0u;
        uint32_t inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
        if (lane == 63) ws[wv] = inc;
        __syncthreads();
        uint32_t at = carry_sh + inc - v;
        for (int k = 0; k < wv; ++k) at += ws[k];
        if (i < nb) cnt[i] = at;
        __syncthreads();
        if (threadIdx.x == 1023) carry_sh = at + v;
        __syncthreads();
    }
}
__global__ __launch_bounds__(256) void k_far_scatter(const uint32_t *__restrict__ flags32, int64_t n, const unsigned long long *__restrict__ far_count,
                                                    const uint32_t *__restrict__ cnt_excl, uint32_t *__restrict__ out)
{
    if (*far_count < kFarWindowedMin) return;
    __shared__ uint32_t ws[4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t base = (int64_t)blockIdx.x * kFarTile + (int64_t)threadIdx.x * 8;
    uint32_t m = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) m |= ((base + k < n && flags32[base + k] != 0u) ? 1u : 0u) << k;
    const uint32_t c = (uint32_t)__popc(m);
    uint32_t inc = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
    if (lane == 63) ws[wv] = inc;
    __syncthreads();
    uint32_t at = cnt_excl[blockIdx.x] + inc - c;
    for (int k = 0; k < wv; ++k) at += ws[k];
#pragma unroll
    for (int k = 0; k < 8; ++k)
        if ((m >> k) & 1u) out[at++] = (uint32_t)(base + k);
}

// (particle, quadrant) pairs outside their quadrant window: the global-field path, one wave per particle.
template <bool COUNT>
__global__ __launch_bounds__(kRayThreads, 8) void k_rays_far(RayArgs a)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t p_end = a.n;
    unsigned long long cnt_exact = 0, cnt_off = 0, cnt_probe = 0;
    const uint32_t *flags32 = reinterpret_cast<const uint32_t *>(a.far_flags);
    if (a.far_windowed && a.far_count && *a.far_count >= kFarWindowedMin) return;     // k_rays_skip<.., FAR> takes this launch
    // each wave scans 64 particles' flags with one coalesced load and visits only the flagged ones.  The 64-particle
    // chunks are dealt round-robin over ALL waves of the grid: in sorted-slot order (k_rays_sweep) the flagged particles
    // sit in a few long runs, which a contiguous range per workgroup would hand to a few workgroups
    // With a list of the flagged slots (k_rays_sweep appends a slot when it sets its first flag) every wave takes list entries
    // round-robin: in sorted-slot order the flagged particles sit in a few long runs, which any partition of the slot range
    // would hand to a few waves.  Without a list the 64-particle chunks of the flag array are dealt round-robin.
    const int64_t n_list = a.far_list ? (int64_t)min((unsigned long long)a.n, *a.far_count) : 0;
    const int64_t wave_global = (int64_t)wave * gridDim.x + blockIdx.x, waves_total = (int64_t)kRayWaves * gridDim.x;
    for (int64_t i0 = a.far_list ? wave_global : wave_global * 64; i0 < (a.far_list ? n_list : p_end); i0 += a.far_list ? waves_total : waves_total * 64) {
      uint32_t myfl;
      unsigned long long todo;
      int64_t listed = 0;
      if (a.far_list) {
          listed = (int64_t)a.far_list[i0];
          myfl = flags32[listed];
          todo = 1ull;
      } else {
          myfl = (i0 + lane < p_end) ? flags32[i0 + lane] : 0u;
          todo = __ballot(myfl != 0u);
      }
      while (todo) {
        const int src = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const int64_t p = a.far_list ? listed : i0 + src;           // particle index, or sorted slot (slot_space)
        const int64_t i = a.slot_space ? (int64_t)a.perm[p] : p;
        const uint32_t fl = a.far_list ? myfl : (uint32_t)__shfl((int)myfl, src, 64);
        if (lane == 0) ++cnt_off;
        const double4 pci = a.slot_space ? a.pcs[p] : a.pc[p];
        const double cth = pci.x, sth = pci.y, gpx = pci.z, gpy = pci.w;
        const bool sane = (gpx > -200000.0) && (gpx < 200000.0) && (gpy > -200000.0) && (gpy < 200000.0);
        const double p0x = (gpx + 1.0 + 262144.0) + kMagic, p0y = (gpy + 1.0 + 262144.0) + kMagic;
        const int base = kCellBase + 262144;
        uint32_t amb0 = 0;
        int cx0 = 0, cy0 = 0;
        bool in0 = false;
        if (sane) {
            uint32_t lox = (uint32_t)__double2loint(p0x), loy = (uint32_t)__double2loint(p0y);
            cx0 = (__double2hiint(p0x) & 0xFFFFF) - base; cy0 = (__double2hiint(p0y) & 0xFFFFF) - base;
            amb0 = lox < loy ? lox : loy;
            in0 = (unsigned)cx0 < (unsigned)a.Wp && (unsigned)cy0 < (unsigned)a.Hp;
        }
        // k_rays_cell does not materialise the ranges: recompute them for the (few) flagged particles
        short4 qr;
        if (a.qr) qr = a.qr[i];
        else { const double t = a.th[i]; qr = quadrant_ranges_of(t, t == t && fabs(t) < 1e6, a.beam_angle, a.B); }
        double acc = 0.0;
        for (int q = 0; q < 4; ++q) {
            if (((fl >> (8 * q)) & 0xFFu) == 0) continue;
            const uint8_t *field = a.distq[q];               // directional field of this quadrant
            const int d0 = in0 ? field[(size_t)cy0 * a.Wps + cx0] : 0;
            const int s0 = d0 > 1 ? d0 : 1;
            int ja, jb, ja2;
            quad_ranges(qr, q, a.B, ja, jb, ja2);
            for (int seg = 0; seg < 2; ++seg)
            for (int j0 = seg ? ja2 : ja, je = seg ? a.B : jb; j0 < je; j0 += 64) {
                int j = j0 + lane;
                if (j < je) {
                    int r = a.P;
                    unsigned np = 0;
                    uint32_t amb = amb0;
                    if (sane) {
                        double2 cs = a.beam_cs[j];
                        double ux = cth * cs.x - sth * cs.y;
                        double uy = sth * cs.x + cth * cs.y;
                        r = trace_fp64<false, COUNT>(a, nullptr, 0, base, p0x, p0y, ux, uy, s0, amb, np, field);
                    }
                    if (!sane || amb < kGuard || a.force_exact == 1) {
                        r = march_exact(a, a.x[i], a.y[i], a.th[i] + (double)a.beam_angle[j]);
                        ++cnt_exact;
                    }
                    acc += (double)a.Lt[(size_t)r * a.bpad + j];
                    if (a.steps || a.steps16) store_step(a, i, j, r);
                    if (COUNT) cnt_probe += np;
                }
            }
        }
        acc = wave_sum(acc);
        if (lane == 0) atomicAdd(&a.logw[p], acc);
      }
    }
    if (a.counters) {
        cnt_exact = wave_sum_u64(cnt_exact);
        cnt_off = wave_sum_u64(cnt_off);
        cnt_probe = wave_sum_u64(cnt_probe);
        if (lane == 0) {
            if (cnt_exact) atomicAdd(&a.counters[0], cnt_exact);
            if (cnt_off) atomicAdd(&a.counters[1], cnt_off);
            if (COUNT && cnt_probe) atomicAdd(&a.counters[2], cnt_probe);
        }
    }
}

// product-as-reference weights (cpp:566-578) from materialised steps: one thread per particle,
// sequential double product in beam order, then pow(.,inv_squash).
__global__ void k_product_weights(const uint8_t *__restrict__ steps, const uint16_t *__restrict__ steps16, const int32_t *__restrict__ obs_idx, int64_t n, int B,
                                  const double *__restrict__ table /* col-major (P+1)^2 */, int tw, double inv_squash,
                                  double *__restrict__ w_out)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double w = 1.0;
    for (int j = 0; j < B; ++j) w *= table[(size_t)(steps16 ? (int)steps16[(size_t)i * B + j] : (int)steps[(size_t)i * B + j]) * tw + obs_idx[j]];
    w_out[i] = pow(w, inv_squash);
}

// ------------------------------------------------------------------------------------------------
// K4 / K7: reductions.  Fixed grid (kRedBlocks x 256) with grid-stride loops and a fixed-order
// final pass: deterministic for a given N.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kRedThreads) void k_reduce_max(const double *__restrict__ v, int64_t n, double *__restrict__ part)
{
    __shared__ double sm[kRedThreads / 64];
    double m = -INFINITY;
    for (int64_t i = (int64_t)blockIdx.x * kRedThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kRedThreads)
        m = fmax(m, v[i]);
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < kRedThreads / 64; ++k) m = fmax(m, sm[k]);
        part[blockIdx.x] = m;
    }
}
// (out2: a second place for the same value -- the caller's exchange buffer in the device-ordered staged flow -- or null)
__global__ __launch_bounds__(kRedThreads) void k_final_max(const double *__restrict__ part, int nb, double *__restrict__ out,
                                                           double *__restrict__ out2)
{
    __shared__ double sm[kRedThreads / 64];
    double m = -INFINITY;
    for (int i = threadIdx.x; i < nb; i += kRedThreads) m = fmax(m, part[i]);
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < kRedThreads / 64; ++k) m = fmax(m, sm[k]);
        out[0] = m;
        if (out2) out2[0] = m;
    }
}

__global__ void k_copy_double(const double *__restrict__ src, double *__restrict__ dst) { dst[0] = src[0]; }
__global__ void k_set_double(double *__restrict__ dst, double v) { dst[0] = v; }
// keeps a stream busy for `ms` milliseconds (at most two seconds): the test switch of the sharded update's bounded wait
__global__ void k_spin_ms(double ms)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();           // 100 MHz
    const unsigned long long ticks = (unsigned long long)((ms < 2000.0 ? ms : 2000.0) * 1e5);
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}

// mcl_group_update: the maximum over the shards' maxima, read where they live (peer pointers)
__global__ void k_group_max(GroupMaxArgs a)
{
    double m = -INFINITY;
    for (int i = 0; i < a.n; ++i) m = fmax(m, a.src[i][0]);
    a.out[0] = m;
}

// Device-ordered staged flow (mcl_stage_weights_async): this shard's contribution to the one SUM exchange of an update, written
// where the collective reads it.  vec = [sum w, sum w x, sum w y, sum w sin, sum w cos | per shard: list length + 1 (0: no list),
// low and high half of the fixed-point weight total | 1 when the ray stage's fix-up lists overflowed | sum w^2]; every other shard's
// slots are zeroed here, so that the SUM over the shards fills them in.  res = the engine's result block.
__global__ void k_stage_pack(const unsigned long long *__restrict__ res, double *__restrict__ vec, int n_shards, int self, int listed,
                             unsigned long long list_cap)
{
    const int len = 5 + 3 * n_shards + 2;
    const double *sc = reinterpret_cast<const double *>(res);
    for (int i = threadIdx.x; i < len; i += blockDim.x) {
        double v = 0.0;
        if (i == 0) v = sc[1];
        else if (i < 5) v = sc[2 + i];
        else if (i == 5 + 3 * self) v = (listed && res[16] <= list_cap) ? (double)(res[16] + 1ull) : 0.0;
        else if (i == 6 + 3 * self) v = (double)(res[2] & 0xFFFFFFFFull);
        else if (i == 7 + 3 * self) v = (double)(res[2] >> 32);
        else if (i == len - 2) v = res[12] != 0ull ? 1.0 : 0.0;
        else if (i == len - 1) v = sc[7];                       // sum w^2 (adaptive resampling: the effective sample size of the whole set)
        vec[i] = v;
    }
}

// from_log: w = det_exp(logw - *maxp) (max element -> exactly 1); else w given, scaled by 1/ *maxp
// for the fixed-point value only.  Writes w, q and per-block partials
// [sum w, sum q (bits), sum w x, sum w y, sum w sin, sum w cos].
__global__ __launch_bounds__(kRedThreads) void k_weights(const double *__restrict__ logw_or_w, int from_log,
                                                        const double *__restrict__ maxp, const double *__restrict__ x,
                                                        const double *__restrict__ y, const double *__restrict__ th, int64_t n,
                                                        double *__restrict__ w_out, uint64_t *__restrict__ q_out,
                                                        double *__restrict__ part /* gridDim.x * 8 */,
                                                        double *__restrict__ carry_out /* logw - max, or null */,
                                                        const double *__restrict__ max_parts = nullptr, int n_max_parts = 0)
{
    __shared__ double sm[kRedThreads / 64][7];
    // max_parts: the maximum is still in pieces (per-workgroup maxima of an earlier kernel; a separate array from `part`): every
    // workgroup reduces the few KB itself -- a maximum does not depend on the order -- instead of a one-workgroup launch in
    // between, and the first leaves it in *maxp for the host
    double mx;
    if (max_parts) {
        __shared__ double mxs[kRedThreads / 64];
        double m = -INFINITY;
        for (int i = threadIdx.x; i < n_max_parts; i += kRedThreads) m = fmax(m, max_parts[i]);
        m = wave_max(m);
        if ((threadIdx.x & 63) == 0) mxs[threadIdx.x >> 6] = m;
        __syncthreads();
        mx = mxs[0];
        for (int k = 1; k < kRedThreads / 64; ++k) mx = fmax(mx, mxs[k]);
        if (blockIdx.x == 0 && threadIdx.x == 0) const_cast<double *>(maxp)[0] = mx;
    } else {
        mx = *maxp;
    }
    double sw = 0, swx = 0, swy = 0, sws = 0, swc = 0, sww = 0;
    uint64_t sq = 0;
    for (int64_t i = (int64_t)blockIdx.x * kRedThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kRedThreads) {
        double w, wq;
        if (from_log) { const double d = logw_or_w[i] - mx; w = det_exp(d); wq = w; if (carry_out) carry_out[i] = d; }
        else {
            w = logw_or_w[i];
            wq = (mx > 0.0 && w > 0.0) ? (w / mx) : 0.0;
        }
        uint64_t q = (uint64_t)(wq * kWeightScale);
        w_out[i] = w;
        q_out[i] = q;
        double s, c;
        sincos(th[i], &s, &c);
        sw += w; sq += q; swx += w * x[i]; swy += w * y[i]; sws += w * s; swc += w * c; sww += w * w;
    }
    sw = wave_sum(sw); swx = wave_sum(swx); swy = wave_sum(swy); sws = wave_sum(sws); swc = wave_sum(swc); sww = wave_sum(sww);
    sq = wave_sum_u64(sq);
    int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        sm[wv][0] = sw; sm[wv][1] = __longlong_as_double((long long)sq); sm[wv][2] = swx; sm[wv][3] = swy; sm[wv][4] = sws; sm[wv][5] = swc; sm[wv][6] = sww;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t tq = 0;
        double t[7] = {0, 0, 0, 0, 0, 0, 0};
        for (int k = 0; k < kRedThreads / 64; ++k) {
            t[0] += sm[k][0]; tq += (uint64_t)__double_as_longlong(sm[k][1]);
            t[2] += sm[k][2]; t[3] += sm[k][3]; t[4] += sm[k][4]; t[5] += sm[k][5]; t[6] += sm[k][6];
        }
        double *p = part + (size_t)blockIdx.x * 8;
        p[0] = t[0]; p[1] = __longlong_as_double((long long)tq); p[2] = t[2]; p[3] = t[3]; p[4] = t[4]; p[5] = t[5]; p[6] = t[6];
    }
}
// scalars[1..6] = fixed-order sums of the partials (scalars[0] = max stays): the first kRedThreads threads of the workgroup,
// thread t over partials t, t + kRedThreads, ..., wave butterflies, waves in order.  sm: [kRedThreads / 64][7] doubles of LDS.
// Every thread of the workgroup must call it (one barrier inside).
__device__ __forceinline__ void final_sums_of(const double *__restrict__ part, int nb, double *__restrict__ scalars, double (*sm)[7])
{
    const bool mine = threadIdx.x < kRedThreads;
    double t[7] = {0, 0, 0, 0, 0, 0, 0};
    uint64_t tq = 0;
    if (mine)
        for (int i = threadIdx.x; i < nb; i += kRedThreads) {
            const double *p = part + (size_t)i * 8;
            t[0] += p[0]; tq += (uint64_t)__double_as_longlong(p[1]); t[2] += p[2]; t[3] += p[3]; t[4] += p[4]; t[5] += p[5]; t[6] += p[6];
        }
    t[0] = wave_sum(t[0]); t[2] = wave_sum(t[2]); t[3] = wave_sum(t[3]); t[4] = wave_sum(t[4]); t[5] = wave_sum(t[5]); t[6] = wave_sum(t[6]);
    tq = wave_sum_u64(tq);
    int wv = threadIdx.x >> 6;
    if (mine && (threadIdx.x & 63) == 0) {
        sm[wv][0] = t[0]; sm[wv][1] = __longlong_as_double((long long)tq); sm[wv][2] = t[2]; sm[wv][3] = t[3]; sm[wv][4] = t[4]; sm[wv][5] = t[5]; sm[wv][6] = t[6];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double r[7] = {0, 0, 0, 0, 0, 0, 0};
        uint64_t rq = 0;
        for (int k = 0; k < kRedThreads / 64; ++k) {
            r[0] += sm[k][0]; rq += (uint64_t)__double_as_longlong(sm[k][1]);
            r[2] += sm[k][2]; r[3] += sm[k][3]; r[4] += sm[k][4]; r[5] += sm[k][5]; r[6] += sm[k][6];
        }
        scalars[1] = r[0]; scalars[2] = __longlong_as_double((long long)rq);
        scalars[3] = r[2]; scalars[4] = r[3]; scalars[5] = r[4]; scalars[6] = r[5];
        scalars[7] = r[6];                      // sum w^2: effective sample size = (sum w)^2 / sum w^2
    }
}
__global__ __launch_bounds__(kRedThreads) void k_final_sums(const double *__restrict__ part, int nb, double *__restrict__ scalars)
{
    __shared__ double sm[kRedThreads / 64][7];
    final_sums_of(part, nb, scalars, sm);
}

// The whole tail of a SMALL update (N <= kTinyTailMax) by ONE workgroup: maximum, max-subtracted weights, fixed-point
// weights, the seven sums and the inclusive CDF -- seven launches of ~4 us each otherwise, and no cross-workgroup
// reduction is needed at this size.  Same formulas as k_reduce_max / k_weights / k_scan_*.  The per-particle work (the
// deterministic exp, the weighted sums) is dealt thread-strided so that all 1024 threads work at the stock 2000-4000
// particles; the fixed-point weights wait in LDS for the scan, which needs consecutive runs per thread.  The sums are
// reduced in this kernel's own fixed order (thread-strided, wave butterflies, waves in order): deterministic for a given N.
// pc: (cos, sin, ., .) of every heading as k_particle_prep left them (the same sincos call), or null.
constexpr int64_t kTinyTailMax = 8192;
__global__ __launch_bounds__(1024) void k_tiny_tail(const double *__restrict__ logw, const double *__restrict__ x, const double *__restrict__ y,
                                                   const double *__restrict__ th, const double4 *__restrict__ pc, int64_t n,
                                                   double *__restrict__ w_out, uint64_t *__restrict__ q_out, uint64_t *__restrict__ cdf_out,
                                                   double *__restrict__ scalars, unsigned long long *__restrict__ host_out,
                                                   unsigned long long host_seq)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char tail_lds[];
    uint64_t *q_sh = reinterpret_cast<uint64_t *>(tail_lds);       // n entries
    __shared__ double sm[16][7];
    __shared__ uint64_t wave_tot[16];
    __shared__ double mx_sh;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double m = -INFINITY;
    for (int64_t i = threadIdx.x; i < n; i += 1024) m = fmax(m, logw[i]);
    m = wave_max(m);
    if (lane == 0) sm[wv][0] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = sm[0][0];
        for (int k = 1; k < 16; ++k) t = fmax(t, sm[k][0]);
        mx_sh = t;
    }
    __syncthreads();
    const double mx = mx_sh;
    double sw = 0, swx = 0, swy = 0, sws = 0, swc = 0, sww = 0;
    for (int64_t i = threadIdx.x; i < n; i += 1024) {
        const double w = det_exp(logw[i] - mx);
        const uint64_t q = (uint64_t)(w * kWeightScale);
        w_out[i] = w;
        q_out[i] = q;
        q_sh[i] = q;
        double s, c;
        if (pc) { const double4 r = pc[i]; c = r.x; s = r.y; }
        else sincos(th[i], &s, &c);
        sw += w; swx += w * x[i]; swy += w * y[i]; sws += w * s; swc += w * c; sww += w * w;
    }
    sw = wave_sum(sw); swx = wave_sum(swx); swy = wave_sum(swy); sws = wave_sum(sws); swc = wave_sum(swc); sww = wave_sum(sww);
    __syncthreads();                            // every q_sh entry written; sm[.][0] (the maxima) read by everyone
    if (lane == 0) { sm[wv][0] = sw; sm[wv][2] = swx; sm[wv][3] = swy; sm[wv][4] = sws; sm[wv][5] = swc; sm[wv][6] = sww; }
    // inclusive scan: thread t owns the `per` consecutive entries from t * per
    const int per = (int)((n + 1023) / 1024);
    const int64_t c0 = (int64_t)threadIdx.x * per;
    uint64_t tot = 0;
    for (int k = 0; k < per; ++k) tot += (c0 + k < n) ? q_sh[c0 + k] : 0ull;
    const uint64_t inc = wave_incl_scan_u64(tot, lane);
    if (lane == 63) wave_tot[wv] = inc;
    __syncthreads();
    uint64_t run = inc - tot;
    for (int k = 0; k < wv; ++k) run += wave_tot[k];
    for (int k = 0; k < per; ++k)
        if (c0 + k < n) { run += q_sh[c0 + k]; cdf_out[c0 + k] = run; }
    if (host_out) __syncthreads();              // every store of this kernel is out before the host is told it may look
    if (threadIdx.x == 0) {
        double r[7] = {0, 0, 0, 0, 0, 0, 0};
        uint64_t rq = 0;
        for (int k = 0; k < 16; ++k) { r[0] += sm[k][0]; r[2] += sm[k][2]; r[3] += sm[k][3]; r[4] += sm[k][4]; r[5] += sm[k][5]; r[6] += sm[k][6]; rq += wave_tot[k]; }
        scalars[0] = mx;
        scalars[1] = r[0]; scalars[2] = __longlong_as_double((long long)rq);
        scalars[3] = r[2]; scalars[4] = r[3]; scalars[5] = r[4]; scalars[6] = r[5];
        scalars[7] = r[6];
        if (host_out) {
            // the result block (8 scalars, then the counters and flags the ray kernels left behind them in the same
            // allocation) straight into pinned host memory: no copy node, no hand-over to a DMA engine
            const unsigned long long *blk = reinterpret_cast<const unsigned long long *>(scalars);
            for (int k = 0; k < 8; ++k) host_out[k] = (unsigned long long)__double_as_longlong(scalars[k]);
            for (int k = 8; k < 14; ++k) host_out[k] = blk[k];
            // the host may be polling word 32 (kResultStamp) instead of waiting for the stream's completion signal
            __threadfence_system();
            __hip_atomic_store(&host_out[32], host_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// out[i] = w[i] / sum (sum > 0) else w[i]   (cpp:680-686)
// log-weights of an update that kept its particles (no resampling): add what they carried
__global__ void k_add_carry(double *__restrict__ logw, const double *__restrict__ carry, int64_t n)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) logw[i] += carry[i];
}

__global__ void k_normalized(const double *__restrict__ w, int64_t n, double sum, double *__restrict__ out)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (sum > 0.0) ? (w[i] / sum) : w[i];
}

// column sums for particles_.colwise().mean() (cpp:904); part = gridDim.x * 4 doubles
__global__ __launch_bounds__(kRedThreads) void k_colsum(const double *__restrict__ x, const double *__restrict__ y,
                                                       const double *__restrict__ th, int64_t n, double *__restrict__ part)
{
    __shared__ double sm[kRedThreads / 64][3];
    double a = 0, b = 0, c = 0;
    for (int64_t i = (int64_t)blockIdx.x * kRedThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kRedThreads) {
        a += x[i]; b += y[i]; c += th[i];
    }
    a = wave_sum(a); b = wave_sum(b); c = wave_sum(c);
    int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sm[wv][0] = a; sm[wv][1] = b; sm[wv][2] = c; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t0 = 0, t1 = 0, t2 = 0;
        for (int k = 0; k < kRedThreads / 64; ++k) { t0 += sm[k][0]; t1 += sm[k][1]; t2 += sm[k][2]; }
        part[blockIdx.x * 4 + 0] = t0; part[blockIdx.x * 4 + 1] = t1; part[blockIdx.x * 4 + 2] = t2;
    }
}

// k draws from the current CDF (visualize, cpp:949-956): out k x 3 column-major
__global__ void k_sample(const double *__restrict__ x, const double *__restrict__ y, const double *__restrict__ th,
                         const uint64_t *__restrict__ cdf, int64_t n, uint64_t q_total, const double *__restrict__ uniforms,
                         uint32_t seed_lo, uint32_t seed_hi, uint32_t ctr, int k, double *__restrict__ out)
{
    int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= k) return;
    uint64_t k53;
    if (uniforms) {
        double u = uniforms[m];
        u = (u >= 0.0) ? u : 0.0;
        k53 = (uint64_t)(u * 9007199254740992.0);
        if (k53 > 9007199254740991ull) k53 = 9007199254740991ull;
    } else {
        u32x4 o = philox4x32((uint32_t)m, ctr, 4u, 0u, seed_lo, seed_hi);
        k53 = bits53(o.v[0], o.v[1]);
    }
    int64_t idx = 0;
    if (q_total) {
        int64_t lo = 0, len = n;
        while (len > 0) {
            int64_t half = len >> 1, mid = lo + half;
            if (!mul_gt(cdf[mid], 1ull << 53, k53, q_total)) { lo = mid + 1; len = len - half - 1; }
            else len = half;
        }
        idx = lo >= n ? n - 1 : lo;
    }
    out[m] = x[idx]; out[k + m] = y[idx]; out[2 * k + m] = th[idx];
}

}  // namespace mcl

#include "mcl_rays_sweep.h"
