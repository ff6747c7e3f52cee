// mcl_rays_sweep.h — K3e: k_rays_sweep, the default ray-cast + likelihood kernel from 65 536 particles (MCL_RAYS_SWEEP).
//
// Cell-sorted particles, one particle per lane, 16 wedge fields, fixed point with a boundary guard, fix-up list, far flags;
// reference rows Q, C, E = cpp:524-650.  What shapes the code (numbers: DESIGN.md 4.5 / 8, HISTORY.md round 5):
//
//   * on gfx950 a wave64 VALU instruction occupies its SIMD for four cycles whatever its kind, and a wave's probe trip is a
//     dependent chain (advance -> address -> LDS read -> wait -> borrow -> exec -> branch) of ~150 cycles that eight waves per SIMD
//     only just cover.  So the walk counts instructions AND keeps its chain short:
//       - positions are 64-bit per axis, cell in the high dword, a 32-bit fraction in the low one, in a 256 x 256-cell window
//         stored MIRRORED per quadrant (every ray of a wedge runs towards +x, +y): both direction components are unsigned 32-bit
//         operands, the position advances by T += skip * X with one v_mad_u64_u32 (no end point), the LDS address of a cell is
//         row * pitch + column of the two high dwords (one v_mad_u32_u24; pitch 260: no row starts in its neighbour's LDS bank);
//       - the guard bias is folded into the origin, so the fraction test is one v_min3_u32 over the two low dwords, and the guard
//         (2^-22 px) is narrow enough to be looked at once per walk, not per ray;
//       - the beam direction is TURNED from beam to beam (an evenly spaced scan: MCL_SW_INTS_REC / MCL_SW_STEP_REC), not fetched;
//       - the table is read as fp64 (no v_cvt_f64_f32), has 127 extra rows below "no hit" (no clamp of the samples left) and one
//         all-zero row; the entry of a ray is requested when the ray ends and added when the next one has ended;
//       - the beam walk of the slots every live lane of the wave has is ONE asm block with its own waitcnt counting, trips unrolled
//         (not-taken exit branches), induction variables advanced by per-lane increments (0 for a lane without rays);
//       - where the walk waits for memory rather than for the VALU (the global-field form, a spread cloud) a lane walks TWO rays
//         at once (MCL_SW2_*);
//       - ranges beyond the window: the HYBRID form -- a lane's walk has a budget of samples inside the window, and a ray that is
//         unstopped when it is spent goes on in the mirrored wedge fields in global memory (MCL_SW_ESCAPE).
//   * A work item is (run of units, group of G wedges; G = 1 by default), planned on the device per update (k_sweep_plan).  The sum of
//     a particle's rays in a wedge goes to its slot's accumulator with one fp64 atomic -- sixty-four lanes on 512 contiguous
//     bytes of the sorted order.
#pragma once

namespace mcl {

constexpr int kSwUnder = 127;            // table rows below "no hit": samples left in [-127, -1] after an over-long jump
constexpr int kSwUnit = 1024;            // particles per scheduling unit: one pass of the 16 waves of a workgroup
constexpr int kSwFx = 32;                // fractional bits of a position: the low dword of a 64-bit value, the cell index is the high dword
constexpr int kSwSide = 256;             // window side in cells
// LDS row pitch of the window in bytes: (cell y, cell x) -> y * kSwPitch + x.  With 256 every row starts in the same LDS bank, and
// the lanes of a wave sample along a ray -- a line that crosses rows: SQ_LDS_BANK_CONFLICT was a quarter of the LDS unit's busy
// cycles.  Rounds 2-4 had no one-instruction address with another pitch (the row was a bit field of a fixed-point word); with the
// cell in the high dword of a 64-bit position it is v_mad_u32_u24 row, pitch, column -- any pitch.
// Ray kernel ms at 4M x 1081 with pitch 256 / 260 / 264 / 272: Spielberg 4.31 / 4.27 / 4.29 / 4.28, the levine stand-in (corridors: rays
// along the axes) 5.72 / 5.02 / 5.03 / - (profiles/r05_experiments/window_pitch.txt); 288 no longer fits two workgroups per CU.
#ifndef MCL_SW_PITCH
#define MCL_SW_PITCH 260
#endif
constexpr int kSwPitch = MCL_SW_PITCH;
static_assert(kSwPitch >= kSwSide && (kSwPitch % 4) == 0, "window rows are written as dwords");
constexpr int kSwWinBytes = kSwSide * kSwPitch;  // the window in LDS
// The HYBRID form (ranges beyond what the window holds): the window is laid out as for a range of kSwHybReach px; a lane's walk
// gets a BUDGET of samples -- as many as no ray of the wedge can leave the window with, from the lane's origin and the wedge's
// largest direction components -- and a ray that is still unstopped when the budget is spent goes on in the global wedge fields.
#ifndef MCL_SW_HYB_PLAY
#define MCL_SW_HYB_PLAY 40
#endif
constexpr int kSwHybPlay = MCL_SW_HYB_PLAY;      // cells of play its windows leave the particles of a work item
constexpr int kSwHybReach = kSwSide - (kSwHybPlay + 3) - 2;      // 211: S - (reach + 2) - 3 = play
// 1 / (largest x component) and 1 / (largest y component) of a unit step in the wedges of the mirrored frame (wedge w of the quadrant:
// directions w .. w + 1 times 90 / (kWedges / 4) degrees): 1 / cos(lower bound), 1 / sin(upper bound)
#if MCL_KWEDGES == 8
__device__ constexpr double kSwHybInvDx[2] = {1.0, 1.414213562373095};
__device__ constexpr double kSwHybInvDy[2] = {1.4142135623730951, 1.0};
#elif MCL_KWEDGES == 16
__device__ constexpr double kSwHybInvDx[4] = {1.0, 1.082392200292394, 1.414213562373095, 2.6131259297527527};
__device__ constexpr double kSwHybInvDy[4] = {2.613125929752753, 1.4142135623730951, 1.082392200292394, 1.0};
#elif MCL_KWEDGES == 32
__device__ constexpr double kSwHybInvDx[8] = {1.0, 1.0195911582083184, 1.082392200292394, 1.2026897738700906, 1.414213562373095, 1.7999524462728311, 2.6131259297527527, 5.125830895483011};
__device__ constexpr double kSwHybInvDy[8] = {5.125830895483013, 2.613125929752753, 1.7999524462728316, 1.4142135623730951, 1.2026897738700906, 1.082392200292394, 1.0195911582083184, 1.0};
#elif MCL_KWEDGES == 64
__device__ constexpr double kSwHybInvDx[16] = {1.0, 1.0048385723763114, 1.0195911582083184, 1.0449972298793777, 1.082392200292394, 1.1338880696327154, 1.2026897738700906, 1.2936435667199802, 1.414213562373095, 1.5763092469025004, 1.7999524462728311, 2.121355371980695, 2.6131259297527527, 3.4448941964766684, 5.125830895483011, 10.202297237378334};
__device__ constexpr double kSwHybInvDy[16] = {10.202297237378328, 5.125830895483013, 3.4448941964766684, 2.613125929752753, 2.121355371980695, 1.7999524462728316, 1.5763092469025004, 1.4142135623730951, 1.2936435667199802, 1.2026897738700906, 1.1338880696327154, 1.082392200292394, 1.0449972298793777, 1.0195911582083184, 1.0048385723763114, 1.0};
#endif
constexpr int kSwMinExtent = 8;          // cells of play a window must leave for the particles of a work item (P <= 243)
// fixed-point scale of a direction component: 2^32 - 3, so that |component| = 1 stays below 2^32 (the operand of v_mad_u64_u32
// has 32 bits) with the + 1 of MCL_SW_ROTATE and its two roundings on top; the guard pays for it with 3 units (2^-32 px) per sample
constexpr double kSwDirScale = 4294967296.0 - 3.0;
// level-1 error bound per axis in units of 2^-32 px: a direction component is off by at most 3 units (scale: -3 .. 0, the + 1 of
// the inner magic, two roundings of half a unit: -3 .. +2) on each of at most P samples, the origin by half a unit, the
// reference's own accumulated rounding (cpp:622-625: P sequential fp64 adds) by < 0.2 unit (3e-11 px, mcl_kernels.h); + slack
__host__ __device__ inline uint32_t sweep_guard_units(int P) { return 3u * (uint32_t)(P + 1) + ((uint32_t)(P + 1) >> 6) + 8u; }

__host__ __device__ inline bool sweep_window_fits(int P) { return kSwSide - (P + 2) - 3 >= kSwMinExtent; }

// rows of the per-update fp64 table of k_rays_sweep: row r = samples left + kSwUnder, plus one all-zero row
__host__ __device__ inline int sweep_table_rows(int P) { return P + kSwUnder + 2; }

// Ltd[r][j]: r = left + kSwUnder for left in [-kSwUnder, P] -> (double)L[obs_idx[j]][P - max(left, 0)];
// r = P + kSwUnder + 1 -> 0 (undecided rays: k_rays_fix adds their entry).  Beam j lives in column j + margin; the columns
// before beam 0 and from beam B on are 0 (lanes without rays, and the virtual beams that pad a scan-edge wedge, see k_rays_sweep).
__global__ void k_build_ltd(const float *__restrict__ L, const int32_t *__restrict__ obs_idx, int B, int cols, int P, int margin,
                            double *__restrict__ Ltd)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;        // column c holds beam c - margin
    const int r = blockIdx.y;
    if (c >= cols) return;
    const int j = c - margin;
    double v = 0.0;
    if (j >= 0 && j < B && r <= P + kSwUnder) {
        const int left = r - kSwUnder;
        const int d = P - (left > 0 ? left : 0);
        v = (double)L[(size_t)obs_idx[j] * (P + 1) + d];
    }
    Ltd[(size_t)r * cols + c] = v;
}

// Units of the sorted order: at most kSwUnit consecutive slots each, ub[u] = first slot of unit u, ub[M] = n, *m_out = M.
// A SPARSE set (SortLayout: ordered by whole buckets of cells -- 32 x 32 tiles, or smaller ones where the play of a window is
// less than a tile -- with the heading below: the spread cloud of a global re-localisation) is cut at the bucket borders as
// well: the particles of a unit then lie within ONE bucket, which the play of a window covers, where a unit that straddles
// two (64 x 32 cells or worse) loses its minority side to the far pass.  Any other set gets the plain grid of kSwUnit slots.
// Where a bucket starts: counting sort -- hist, the bucket offsets k_hist_final left (copy of XCD 0 = the first slot of a key),
// and part, the histogram tiles' totals (their exclusive prefix, formed here, is the offset of every key of a tile nobody fell into); radix
// sort -- cut_start / cut_end, written by k_sort_gather where the sorted keys change bucket (zeroed here for the next update).
// One workgroup.
__global__ __launch_bounds__(1024) void k_unit_table(const int *__restrict__ bbox, int64_t n, const uint32_t *__restrict__ hist,
                                                    const uint32_t *__restrict__ part, const uint32_t *__restrict__ tile_used,
                                                    uint32_t *__restrict__ cut_start, uint32_t *__restrict__ cut_end, uint32_t *__restrict__ ub,
                                                    int *__restrict__ m_out, int max_units)
{
    __shared__ uint32_t ws[16];
    __shared__ uint32_t carry_sh, nbig_sh;
    __shared__ uint4 big_sh[1024];
    constexpr int kHistParts = (int)(kSortKeySpace / kHistTile);
    static_assert(kHistParts % 1024 == 0, "the scan of the tiles' totals below takes 1024 at a time");
    __shared__ uint32_t part_ex[kHistParts];
    const SortLayout L = sort_layout(bbox, n);
    const int shift = L.cut_shift();
    const bool cut = L.cuts();
    const int64_t Mgrid = (n + kSwUnit - 1) / kSwUnit;
    if (!cut) {
        for (int64_t u = threadIdx.x; u <= Mgrid; u += 1024) ub[u] = (uint32_t)(u * kSwUnit < n ? u * kSwUnit : n);
        if (threadIdx.x == 0) m_out[0] = (int)Mgrid;
        return;
    }
    static_assert(kHistTile == 4096, "a bucket's keys (at most 2^8 headings) lie inside one histogram tile");
    const int ntl = (int)L.cut_groups();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (threadIdx.x == 0) { carry_sh = 0u; nbig_sh = 0u; }
    if (!cut_start) {
        // counting sort: the histogram tiles' totals (k_hist_partials) -> the slots in front of every tile
        for (int t0 = 0; t0 < kHistParts; t0 += 1024) {
            const uint32_t v = part[t0 + threadIdx.x];
            uint32_t inc = v;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
            if (lane == 63) ws[wv] = inc;
            __syncthreads();
            uint32_t off = carry_sh + inc - v;
            for (int k = 0; k < wv; ++k) off += ws[k];
            part_ex[t0 + threadIdx.x] = off;
            __syncthreads();
            if (threadIdx.x == 1023) carry_sh = off + v;
            __syncthreads();
        }
        if (threadIdx.x == 0) carry_sh = 0u;
    }
    __syncthreads();
    auto start_of = [&](int t) -> uint32_t {               // counting sort: first slot of bucket t's first key
        if (t >= ntl) return (uint32_t)n;
        const uint32_t key0 = (uint32_t)t << shift;
        return tile_used[key0 >> 12] ? hist[key0] : part_ex[key0 >> 12];
    };
    for (int t0 = 0; t0 < ntl; t0 += 1024) {
        const int t = t0 + (int)threadIdx.x;
        uint32_t s0 = 0, cnt = 0;
        if (t < ntl) {
            if (cut_start) {
                const uint32_t a1 = cut_start[t];
                if (a1 != 0u) { s0 = a1 - 1u; const uint32_t e = cut_end[t]; cnt = e > s0 ? e - s0 : 0u; cut_start[t] = 0u; cut_end[t] = 0u; }
            } else {
                s0 = start_of(t);
                const uint32_t s1 = start_of(t + 1);
                cnt = s1 > s0 ? s1 - s0 : 0u;
            }
        }
        const uint32_t nu = (cnt + kSwUnit - 1) / kSwUnit;
        uint32_t inc = nu;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(inc, o, 64); if (lane >= o) inc += v; }
        if (lane == 63) ws[wv] = inc;
        __syncthreads();
        uint32_t at = carry_sh + inc - nu;
        for (int k = 0; k < wv; ++k) at += ws[k];
        // a group of a few units is written by its thread; a long one (a dense set has thousands of particles per tile) by all
        if (nu <= 4u) {
            for (uint32_t j = 0; j < nu; ++j)
                if (at + j < (uint32_t)max_units) ub[at + j] = s0 + j * kSwUnit;
        } else {
            const uint32_t e = atomicAdd(&nbig_sh, 1u);
            big_sh[e] = make_uint4(at, s0, nu, 0u);
        }
        __syncthreads();
        for (uint32_t e = 0; e < nbig_sh; ++e) {
            const uint4 b = big_sh[e];
            for (uint32_t j = threadIdx.x; j < b.z; j += 1024u)
                if (b.x + j < (uint32_t)max_units) ub[b.x + j] = b.y + j * kSwUnit;
        }
        if (threadIdx.x == 1023) carry_sh = at + nu;
        __syncthreads();
        if (threadIdx.x == 0) nbig_sh = 0u;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        int M = (int)carry_sh;
        if (M > max_units) M = max_units;                  // cannot happen: max_units = ceil(n / kSwUnit) + buckets (never write out of bounds)
        ub[M] = (uint32_t)n;
        m_out[0] = M;
    }
}

// Per unit of the sorted order, over its finite positions: out[2u] = (sum of pixel x, sum of pixel y, count, -) and
// out[2u + 1] = (min x, max x, min y, max y).  A work item centres its windows on the bounding box of its units; k_sweep_plan
// joins neighbouring units into one item while their particles still fit one window.
// (hist / tile_used / nparts, when given: k_hist_clear's work -- the used tiles of the counting sort's histogram back to zero for the
//  next update -- done here, after k_unit_table has read the bucket offsets: one launch less in front of the ray kernel)
__global__ __launch_bounds__(256) void k_unit_sums(const double4 *__restrict__ pcs, const uint32_t *__restrict__ ub, const int *__restrict__ m_ptr,
                                                  double4 *__restrict__ out, uint32_t *__restrict__ hist, uint32_t *__restrict__ tile_used, int nparts)
{
    __shared__ double sm[4][7];
    if (hist) {
        for (int t = (int)blockIdx.x; t < nparts; t += (int)gridDim.x) {
            if (!tile_used[t]) continue;                       // (block-uniform)
            const uint4 z = make_uint4(0u, 0u, 0u, 0u);
            for (int x = 0; x < kSortXcds; ++x) {
                uint4 *p = reinterpret_cast<uint4 *>(hist + (size_t)x * kSortKeySpace + (size_t)t * kHistTile) + threadIdx.x * 4;
#pragma unroll
                for (int k = 0; k < 4; ++k) p[k] = z;
            }
            __syncthreads();
            if (threadIdx.x == 0) tile_used[t] = 0u;
        }
    }
    // (the number of units is only known on the device: a fixed grid strides over them -- one block per POSSIBLE unit was 65 000
    //  empty blocks, 10 us of a 262 144-particle update)
    for (int u = (int)blockIdx.x; u < m_ptr[0]; u += (int)gridDim.x) {
    if (u != (int)blockIdx.x) __syncthreads();             // sm is reused
    const int64_t p_begin = (int64_t)ub[u];
    const int64_t p_end = (int64_t)ub[u + 1];
    double sx = 0.0, sy = 0.0, cnt = 0.0, x0 = INFINITY, x1 = -INFINITY, y0 = INFINITY, y1 = -INFINITY;
    for (int64_t s = p_begin + threadIdx.x; s < p_end; s += blockDim.x) {
        const double4 c = pcs[s];
        if (c.z == c.z && c.w == c.w && fabs(c.z) < 1e9 && fabs(c.w) < 1e9) {
            sx += c.z; sy += c.w; cnt += 1.0;
            x0 = fmin(x0, c.z); x1 = fmax(x1, c.z); y0 = fmin(y0, c.w); y1 = fmax(y1, c.w);
        }
    }
    sx = wave_sum(sx); sy = wave_sum(sy); cnt = wave_sum(cnt);
    x0 = -wave_max(-x0); x1 = wave_max(x1); y0 = -wave_max(-y0); y1 = wave_max(y1);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sm[w][0] = sx; sm[w][1] = sy; sm[w][2] = cnt; sm[w][3] = x0; sm[w][4] = x1; sm[w][5] = y0; sm[w][6] = y1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        out[2 * (size_t)u] = make_double4(sm[0][0] + sm[1][0] + sm[2][0] + sm[3][0], sm[0][1] + sm[1][1] + sm[2][1] + sm[3][1],
                                                  sm[0][2] + sm[1][2] + sm[2][2] + sm[3][2], 0.0);
        out[2 * (size_t)u + 1] = make_double4(fmin(fmin(sm[0][3], sm[1][3]), fmin(sm[2][3], sm[3][3])), fmax(fmax(sm[0][4], sm[1][4]), fmax(sm[2][4], sm[3][4])),
                                                      fmin(fmin(sm[0][5], sm[1][5]), fmin(sm[2][5], sm[3][5])), fmax(fmax(sm[0][6], sm[1][6]), fmax(sm[2][6], sm[3][6])));
    }
    }
}

// Work items of k_rays_sweep: (first unit, units, wedge group).  One thread looks at an aligned block of kSwRunMax units:
// the run length the guided schedule wants there (long runs while plenty of work is left -- fewer window loads and fewer
// workgroup barriers per particle --, single units towards the end, so that the persistent workgroups finish within
// one small item of each other) is halved until the particles of every run fit one window (`half_play` cells either side of
// the centre of the run's bounding box, both axes; a sparse cloud or a long range thus gets shorter runs instead of off-window
// particles).  Every run is listed once per wedge group.  One workgroup; items come out in unit order.
// (the guide: a run is as long as remaining items / (guide x workgroups) allows; `guide` comes from the host, MCL_SW_GUIDE=<n> overrides
//  it.  Ray kernel ms at 4M / 1M / 262 144 x 1081 with 1: 4.40 / 1.26 / 0.47, 2: 4.24 / 1.22 / 0.41, 3 -- rounds 3-4, tuned on a
//  kernel a third slower --: 4.34 / 1.26 / 0.42, 4: 4.38 / 1.29 / 0.44; on the 479-px map the hybrid form 2: 5.98, 3: 5.72, 4: 5.77 and the global-field form 2: 12.21, 3: 12.02 (their
//  chunks vary more: a probe served from L2 costs several times one from LDS); profiles/r05_experiments/guide.txt)
#ifndef MCL_SW_GUIDE
#define MCL_SW_GUIDE 2
#endif
#ifndef MCL_SW_RUNMAX
#define MCL_SW_RUNMAX 16
#endif
constexpr int kSwRunMax = MCL_SW_RUNMAX;
static_assert(kSwRunMax == 16, "k_sweep_plan's bounding-box pyramid has four levels above the units");
constexpr int kPlanBlocks = 256;                 // blocks of kSwRunMax units planned per pass of the workgroup
constexpr int kPlanLds = (kPlanBlocks * kSwRunMax + kPlanBlocks * kSwRunMax * 15 / 16) * (int)sizeof(float4);     // the unit boxes and levels 1..4 of one pass: 126 976 bytes
__global__ __launch_bounds__(1024) void k_sweep_plan(const double4 *__restrict__ unit_stats, const int *__restrict__ m_ptr, int ngroups, int nwg, double half_play,
                                                    int guide, int4 *__restrict__ items, int4 *runs, int *__restrict__ nitems_out)
{
    // Bounding boxes of the aligned runs of 2, 4, 8 and 16 units as a pyramid in LDS (floats: the plan is a heuristic, the ray
    // kernel tests every particle against its window exactly), built by all threads; then one thread per block of 16 units
    // halves its candidate runs until they fit, one pyramid node per test.
    extern __shared__ __attribute__((aligned(16))) unsigned char plan_lds[];
    // (the unit boxes themselves too: the planning threads read them one after the other -- from global memory that was a chain of
    //  sixteen dependent loads per thread and most of this kernel's 18 .. 29 us)
    constexpr int U = kPlanBlocks * kSwRunMax;                   // units per pass
    float4 *box0 = reinterpret_cast<float4 *>(plan_lds);         // boxes of the units of this pass
    float4 *pyr = box0 + U;                                      // level l (1..4) of this pass at pyr + lvl_off[l]
    __shared__ int wave_tot[4];
    __shared__ int carry_sh;
    const int M = m_ptr[0];                   // units of this update's sorted order (k_unit_table)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    constexpr int off1 = 0, off2 = U / 2, off3 = off2 + U / 4, off4 = off3 + U / 8;
    const float hp = (float)half_play;
    if (threadIdx.x == 0) carry_sh = 0;
    __syncthreads();
    const int nblocks = (M + kSwRunMax - 1) / kSwRunMax;
    auto unit_box = [&](int u) -> float4 {                       // (x0, x1, y0, y1); empty: x1 < x0
        if (u >= M) return make_float4(INFINITY, -INFINITY, INFINITY, -INFINITY);
        const double4 bb = unit_stats[2 * (size_t)u + 1];
        return make_float4((float)bb.x, (float)bb.y, (float)bb.z, (float)bb.w);
    };
    auto join = [](float4 p, float4 q) { return make_float4(fminf(p.x, q.x), fmaxf(p.y, q.y), fminf(p.z, q.z), fmaxf(p.w, q.w)); };
    for (int b0 = 0; b0 < nblocks; b0 += kPlanBlocks) {
        const int ubase = b0 * kSwRunMax;
        // (only the blocks of this pass that hold units: a 262 144-particle set is 256 units of the 4096 a pass has room for)
        const int ucnt = min(U, ((M - ubase + kSwRunMax - 1) / kSwRunMax) * kSwRunMax);
        for (int k = threadIdx.x; k < ucnt; k += 1024) box0[k] = unit_box(ubase + k);
        __syncthreads();
        for (int k = threadIdx.x; k < ucnt / 2; k += 1024) pyr[off1 + k] = join(box0[2 * k], box0[2 * k + 1]);
        __syncthreads();
        for (int k = threadIdx.x; k < ucnt / 4; k += 1024) pyr[off2 + k] = join(pyr[off1 + 2 * k], pyr[off1 + 2 * k + 1]);
        __syncthreads();
        for (int k = threadIdx.x; k < ucnt / 8; k += 1024) pyr[off3 + k] = join(pyr[off2 + 2 * k], pyr[off2 + 2 * k + 1]);
        __syncthreads();
        for (int k = threadIdx.x; k < ucnt / 16; k += 1024) pyr[off4 + k] = join(pyr[off3 + 2 * k], pyr[off3 + 2 * k + 1]);
        __syncthreads();
        // box of the aligned run of cc units (a power of two) that starts at unit index r of this pass
        auto run_box = [&](int r, int cc) -> float4 {
            return cc == 1 ? box0[r] : cc == 2 ? pyr[off1 + (r >> 1)] : cc == 4 ? pyr[off2 + (r >> 2)] : cc == 8 ? pyr[off3 + (r >> 3)] : pyr[off4 + (r >> 4)];
        };
        const int b = b0 + (int)threadIdx.x;
        const int u0 = b * kSwRunMax, r0 = (int)threadIdx.x * kSwRunMax;
        const bool mine = (int)threadIdx.x < kPlanBlocks && b < nblocks;
        unsigned int starts = 0u;                 // bit k: a run starts at unit k of the block (it ends where the next one starts)
        int nruns = 0;
        if (mine) {
            long long want = ((long long)(M - u0) * ngroups) / ((long long)(guide > 0 ? guide : MCL_SW_GUIDE) * (nwg > 0 ? nwg : 1));
            int c = 1;                            // (a floor of 2 or 4 units per run costs 0.5 / 2.5 %: the tail of the launch)
            while (c * 2 <= kSwRunMax && c * 2 <= want) c *= 2;
            // runs of c units, each halved until it fits
            for (int k0 = 0; k0 < kSwRunMax && u0 + k0 < M; k0 += c) {
                int cc = c;
                int k = k0;
                while (k < k0 + c && u0 + k < M) {
                    bool fits = true;
                    if (cc > 1) {
                        // the windows of a run are centred on the bounding box of its particles: they fit when the box is
                        // narrower than the play on both axes
                        const float4 bb = run_box(r0 + k, cc);
                        if (bb.y >= bb.x) fits = 0.5f * (bb.y - bb.x) < hp && 0.5f * (bb.w - bb.z) < hp;
                    }
                    if (fits || cc == 1) { starts |= 1u << k; ++nruns; k += cc; }
                    else cc >>= 1;                      // try the first half; the second half is tried at the same (halved) length
                }
            }
        }
        // position of this block's first run: exclusive scan of nruns over the planning threads, carried across passes
        int inc = nruns;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
        if (lane == 63 && wv < 4) wave_tot[wv] = inc;
        __syncthreads();
        const int at_base = carry_sh;
        int at = at_base + inc - nruns;
        for (int k = 0; k < wv && k < 4; ++k) at += wave_tot[k];
        if (mine) {
            const int uend = (u0 + kSwRunMax < M) ? kSwRunMax : M - u0;          // units of this block
            unsigned int rest = starts;
            while (rest) {
                const int k = __ffs((int)rest) - 1;
                rest &= rest - 1u;
                const int knext = rest ? __ffs((int)rest) - 1 : uend;
                const int len = knext - k;
                // power-of-two cover of the run for its box: a run is an aligned power of two, cut short only by the end of the set
                int cc = 1;
                while (cc < len) cc <<= 1;
                const float4 bb = run_box(r0 + k, cc);
                const float mx = bb.y >= bb.x ? 0.5f * (bb.x + bb.y) : 0.0f, my = bb.y >= bb.x ? 0.5f * (bb.z + bb.w) : 0.0f;
                runs[at++] = make_int4((int)floorf(mx) + 1, (int)floorf(my) + 1, u0 + k, len);      // padded cell of the centre, first unit, units
            }
        }
        __syncthreads();                          // the run table of this pass is complete (and visible: same workgroup)
        if ((int)threadIdx.x == kPlanBlocks - 1) carry_sh = at;      // the last planning thread's end position = the total so far
        __syncthreads();
        // every run once per wedge group, written by all threads
        const int nr = carry_sh - at_base;
        for (int idx = threadIdx.x; idx < nr * ngroups; idx += 1024) {
            const int r = at_base + idx / ngroups, g = idx - (idx / ngroups) * ngroups;
            const int4 rn = runs[r];
            items[(size_t)r * ngroups + g] = make_int4(rn.z, rn.w, (g + rn.z) % ngroups, r);
        }
        __syncthreads();                          // the pyramid is rebuilt by the next pass
    }
    if (threadIdx.x == 0) nitems_out[0] = carry_sh * ngroups;
}

// One probe trip.  A position is a 64-bit fixed-point number per axis: the cell index in the high dword, a 32-bit fraction
// in the low one; it advances by T += skip * X with ONE v_mad_u64_u32 per axis (4.5 cycles, profiles/r02_op_rates.txt; the
// 24-bit form of rounds 2-4 needed a v_mad_u32_u24 and a v_and each).  The window is stored MIRRORED per quadrant so that every
// ray of the wedge runs towards +x and +y: both direction components are unsigned 32-bit operands X.  The high dwords are the
// cell coordinates as they stand (LDS address = row << 8 | column: one v_lshl_or_b32), the low dwords are the fractions as they
// stand: positions carry the guard bias G, so
//   frac(T) < 2G  <=>  the unbiased sample lies within G units (2^-32 px) of a cell boundary (either side)
// is one v_min3_u32 over the two low dwords, no masks; the cell is read at the biased position, which differs from the true cell
// only for such a sample.  Five VALU per trip (rounds 2-4: seven).
// BYIN = the skip that leads to this sample (the own cell's on the first trip, the byte just read afterwards).
#define MCL_SW_TRIP(REM, GIN, BYIN, TXIN, TYIN, TX, TY, TXLO, TXHI, TYLO, TYHI, AD, BY, GOUT, REMOUT, XX, XY, LB) \
    "v_mad_u64_u32 " TX ", vcc, " BYIN ", " XX ", " TXIN "\n\t"                                           \
    "v_mad_u64_u32 " TY ", vcc, " BYIN ", " XY ", " TYIN "\n\t"                                           \
    "v_mad_u32_u24 " AD ", " TYHI ", %[wp], " TXHI "\n\t"                                                 \
    "ds_read_i8 " BY ", " AD " offset:" LB "\n\t"                                                         \
    "v_min3_u32 " GOUT ", " GIN ", " TXLO ", " TYLO "\n\t"                                                \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                            \
    "v_sub_co_u32 " REMOUT ", vcc, " REM ", " BY "\n\t"                                                   \
    "s_andn2_b64 exec, exec, vcc\n\t"

// ---- the direction of a beam, two ways ----------------------------------------------------------------------------------------
// TAB (any monotone scan): (cos a_j, sin a_j) of every beam from a table in global memory (RayArgs::beam_csx), one 16-byte
// gather per ray, rotated by the particle's heading:
//   Xx = aq cb - bq sb,  Xy = +-(bq cb + aq sb)   (aq, bq = the particle's cos / sin times +-kSwDirScale; v[40:43] = cb, sb)
// with the 1.5 * 2^52 magic folded into the inner FMA: two v_fma_f64 per component (rounds 2-4: mul, fma, add), each rounding to
// a whole unit -- |error| <= 1 unit, paid for by the guard.  The inner magic carries +1: the sum is then >= 0 whatever the two
// roundings do to a component that is 0 in exact arithmetic (a ray along an axis), see sweep_guard_units.
// NEGA / NEGB = "-" in the quadrants where the sign of the y component differs from x's.
#define MCL_SW_DIR_TAB(NEGA, NEGB)                                                                                              \
        "s_waitcnt vmcnt(1)\n\t" /* direction landed (the table entry may be in flight) */                                     \
        "v_fma_f64 v[48:49], -%[bq], v[42:43], %[magic1]\n\t"                                                                  \
        "v_fma_f64 v[44:45], %[aq], v[40:41], v[48:49]\n\t"                                                                    \
        "v_fma_f64 v[48:49], " NEGA "%[aq], v[42:43], %[magic1]\n\t"                                                           \
        "v_fma_f64 v[46:47], " NEGB "%[bq], v[40:41], v[48:49]\n\t"                                                            \
        "v_add_u32 %[j16], %[j16], %[inc16]\n\t"                                                                               \
        "global_load_dwordx4 v[40:43], %[j16], %[csb]\n\t" /* next beam's direction */
// REC (an evenly spaced scan, mcl_set_beam_angles: every angle within 2e-6 rad of the grid a0 + j inc -- any real lidar's): no
// gather.  The 16-byte gather of TAB costs the ray stage a tenth of its time (the texture path takes 64 addresses and 1 KB per
// ray; timing-only variants in profiles/r05_experiments).  Here the lane carries the scaled, mirrored directions of the GRID
// angles of TWO consecutive beams as fp64 pairs (xa, ya), (xb, yb) and steps them with the three-term recurrence of a rotation,
//   d(j + 2) = K d(j + 1) - d(j),   K = 2 cos(inc)
// -- one v_fma_f64 per component and beam, the same in every mirrored frame (it is linear in each component); 90 steps leave an
// error of 2^-9 unit -- and the beam's own offset from the grid e_j = a_j - (a0 + j inc) -- the float rounding of its angle, a
// few 1e-8 rad, 8 bytes per beam in LDS behind the window -- enters to first order where the integers are made:
//   Xx = x - nu e y + (magic + 1),  Xy = y + nu e x + (magic + 1)     (nu = -1 in the quadrants mirrored in one axis)
// (the second-order term e^2 / 2 is below 1e-2 unit).  Same two roundings, same bound as TAB.  Six VALU per beam (TAB: four and
// the gather).  NEGX = "-" / NEGY = "" normally, "" / "-" where nu = -1.  E = the register pair holding e_j.
#define MCL_SW_INTS_REC(NEGX, NEGY, E, XS, YS)                                                                                  \
        "v_add_f64 v[48:49], " XS ", %[magic1]\n\t"                                                                            \
        "v_fma_f64 v[44:45], " NEGX E ", " YS ", v[48:49]\n\t"                                                                  \
        "v_add_f64 v[48:49], " YS ", %[magic1]\n\t"                                                                            \
        "v_fma_f64 v[46:47], " NEGY E ", " XS ", v[48:49]\n\t"
#define MCL_SW_STEP_REC(XS, YS, XO, YO) /* (XS, YS) two beams on: K * other - self */                                           \
        "v_fma_f64 " XS ", %[rk], " XO ", -" XS "\n\t"                                                                          \
        "v_fma_f64 " YS ", %[rk], " YO ", -" YS "\n\t"

// ---- the trips of one ray: the first, then six per turn of a loop -- a ray makes 3.4, a wave 4.5 -- so that the walk sees not-taken
// exit branches only (the countdown and its taken branch cost a wave ~20 cycles per trip in a chain that is latency-bound,
// tools/ubench/trip_rates.hip)
#define MCL_SW_TRIP_FIRST MCL_SW_TRIP("%[rem0]", "%[g]", "%[s0]", "%[p0x]", "%[p0y]", "v[52:53]", "v[54:55]", "v52", "v53", "v54", "v55", "v48", "v49", "%[g]", "v57", "v44", "v46", "%[lb]")
#define MCL_SW_TRIP_NEXT MCL_SW_TRIP("v57", "%[g]", "v49", "v[52:53]", "v[54:55]", "v[52:53]", "v[54:55]", "v52", "v53", "v54", "v55", "v48", "v49", "%[g]", "v57", "v44", "v46", "%[lb]")
#define MCL_SW_TRIPS_LDS                                                                                                       \
        "s_movk_i32 %[cd], 50\n\t"                                                                                             \
        MCL_SW_TRIP_FIRST                                                                                                      \
        "s_cbranch_execz 2f\n"                                                                                                 \
        "1:\n\t"                                                                                                               \
        MCL_SW_TRIP_NEXT "s_cbranch_execz 2f\n\t"                                                                               \
        MCL_SW_TRIP_NEXT "s_cbranch_execz 2f\n\t"                                                                               \
        MCL_SW_TRIP_NEXT "s_cbranch_execz 2f\n\t"                                                                               \
        MCL_SW_TRIP_NEXT "s_cbranch_execz 2f\n\t"                                                                               \
        MCL_SW_TRIP_NEXT "s_cbranch_execz 2f\n\t"                                                                               \
        MCL_SW_TRIP_NEXT "s_cbranch_execz 2f\n\t"                                                                               \
        "s_sub_u32 %[cd], %[cd], 1\n\t"                                                                                        \
        "s_cbranch_scc0 1b\n\t"                                                                                                \
        "s_mov_b32 %[expired], 1\n" /* malformed window (impossible): the pass is redone by the fix-up path */                 \
        "2:\n\t"                                                                                                               \
        "s_mov_b64 exec, -1\n\t"

// The beam walk over the slots every live lane of the wave has, as one asm block: all 64 lanes active, no per-slot validity
// tests, no per-ray test for undecided rays either: the guard minimum %[g] runs over ALL the rays of the walk and is looked at
// once, after it (with 32 fractional bits a sample falls inside the guard once in a few million; the lane that has one hands
// the walk's rays to the fix-up list, see below).
//   v[40:43] direction of the beam (TAB) / v[40:41] its offset from the grid (REC); v[44:45] / v[46:47] rotated direction +
//   magic: Xx = v44, Xy = v46; v[48:49] scratch, then LDS address / cell byte / table offset; v[50:51] pending table entry;
//   v[52:53] Tx (fraction, cell); v[54:55] Ty; v57 samples left
// The table entry of a ray is requested when the ray ends and added when the NEXT ray has ended: TAB keeps two reads in flight
// (entry, next direction: they return in order), REC one.
#define MCL_SW_TAIL(WAITCNT)                                                                                                   \
        WAITCNT /* previous beam's table entry landed */                                                                       \
        "v_add_f64 %[acc], %[acc], v[50:51]\n\t"                                                                               \
        "v_mad_i32_i24 v48, v57, %[st8], %[j8b]\n\t"                                                                           \
        "v_add_u32 %[j8b], %[j8b], %[inc8]\n\t"                                                                                \
        "global_load_dwordx2 v[50:51], v48, %[ltb]\n\t"                                                                        \
        "s_sub_u32 %[tc], %[tc], 1\n\t"                                                                                        \
        "s_cbranch_scc0 3b\n\t"                                                                                                \
        "s_waitcnt vmcnt(0)\n\t"                                                                                               \
        "v_add_f64 %[acc], %[acc], v[50:51]"
// (REC: four beams per turn of the loop -- the table columns of the four are 8 bytes apart, immediate offsets: the column offset
//  and the pointer to the beams' offsets move once per turn, a quarter of a VALU instruction per ray each --; a beam's tail leaves
//  the loop when it was the walk's last)
#define MCL_SW_TAIL_R(OFF)                                                                                                     \
        "s_waitcnt vmcnt(0)\n\t"                                                                                               \
        "v_add_f64 %[acc], %[acc], v[50:51]\n\t"                                                                               \
        "v_mad_i32_i24 v48, v57, %[st8], %[j8b]\n\t"                                                                           \
        "global_load_dwordx2 v[50:51], v48, %[ltb] offset:" OFF "\n\t"                                                         \
        "s_sub_u32 %[tc], %[tc], 1\n\t"                                                                                        \
        "s_cbranch_scc1 5f\n\t"
#define MCL_SW_TAIL_R_LAST(OFF)                                                                                                \
        "s_waitcnt vmcnt(0)\n\t"                                                                                               \
        "v_add_f64 %[acc], %[acc], v[50:51]\n\t"                                                                               \
        "v_mad_i32_i24 v48, v57, %[st8], %[j8b]\n\t"                                                                           \
        "v_add_u32 %[j8b], %[j8b], %[ince]\n\t"                                                                                \
        "global_load_dwordx2 v[50:51], v48, %[ltb] offset:" OFF "\n\t"                                                         \
        "s_sub_u32 %[tc], %[tc], 1\n\t"                                                                                        \
        "s_cbranch_scc0 3b\n"                                                                                                  \
        "5:\n\t"                                                                                                               \
        "s_waitcnt vmcnt(0)\n\t"                                                                                               \
        "v_add_f64 %[acc], %[acc], v[50:51]"
#define MCL_SW_CLOBBERS "memory", "vcc", "scc", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v57"

#define MCL_SW_WALK_TAB(NEGA, NEGB)                                                                                            \
    asm volatile(                                                                                                              \
        "global_load_dwordx4 v[40:43], %[j16], %[csb]\n\t"                                                                     \
        "v_mov_b32 v48, %[zoff]\n\t"                                                                                           \
        "global_load_dwordx2 v[50:51], v48, %[ltb]\n\t" /* 0.0: keeps the vmcnt pattern of the steady state */                 \
        "3:\n\t"                                                                                                               \
        MCL_SW_DIR_TAB(NEGA, NEGB)                                                                                             \
        MCL_SW_TRIPS_LDS                                                                                                       \
        MCL_SW_TAIL("s_waitcnt vmcnt(1)\n\t")                                                                                  \
        : [acc] "+v"(acc_fast), [j16] "+v"(j16), [j8b] "+v"(j8b), [g] "+v"(gwalk),                                              \
          [tc] "+s"(tc), [expired] "+s"(expired), [cd] "=&s"(cd)                                                               \
        : [aq] "v"(aq), [bq] "v"(bq), [p0x] "v"(P0x), [p0y] "v"(P0y), [rem0] "v"(rem_start), [s0] "v"(s0e),                     \
          [inc16] "v"(inc16), [inc8] "v"(inc8), [csb] "s"(a.beam_csx), [ltb] "s"(a.Ltd), [st8] "s"(st8),                        \
          [magic1] "s"(6755399441055745.0), [zoff] "s"(zoff), [lb] "n"(kQLdsBase), [wp] "s"(wpitch)                                              \
        : MCL_SW_CLOBBERS)

#define MCL_SW_WALK_REC_T(NEGX, NEGY, TRIPS, XOUT, ...)                                                                                           \
    asm volatile(                                                                                                              \
        "ds_read2_b64 v[40:43], %[je] offset1:1\n\t" /* offsets of the first two beams; the next two are read two beams ahead */ \
        "v_mov_b32 v48, %[zoff]\n\t"                                                                                           \
        "global_load_dwordx2 v[50:51], v48, %[ltb]\n\t" /* 0.0 */                                                              \
        "3:\n\t"                                                                                                               \
        "s_waitcnt lgkmcnt(0)\n\t"                                                                                             \
        MCL_SW_INTS_REC(NEGX, NEGY, "v[40:41]", "%[xa]", "%[ya]")                                                              \
        TRIPS                                                                                                                  \
        MCL_SW_TAIL_R("0")                                                                                                     \
        MCL_SW_STEP_REC("%[xa]", "%[ya]", "%[xb]", "%[yb]")                                                                    \
        MCL_SW_INTS_REC(NEGX, NEGY, "v[42:43]", "%[xb]", "%[yb]")                                                              \
        "ds_read2_b64 v[40:43], %[je] offset0:2 offset1:3\n\t" /* (landed by the first trip's wait) */                         \
        TRIPS                                                                                                                  \
        MCL_SW_STEP_REC("%[xb]", "%[yb]", "%[xa]", "%[ya]")                                                                    \
        MCL_SW_TAIL_R("8")                                                                                                     \
        MCL_SW_INTS_REC(NEGX, NEGY, "v[40:41]", "%[xa]", "%[ya]")                                                              \
        TRIPS                                                                                                                  \
        MCL_SW_TAIL_R("16")                                                                                                    \
        MCL_SW_STEP_REC("%[xa]", "%[ya]", "%[xb]", "%[yb]")                                                                    \
        MCL_SW_INTS_REC(NEGX, NEGY, "v[42:43]", "%[xb]", "%[yb]")                                                              \
        "v_add_u32 %[je], %[je], %[ince]\n\t"                                                                                  \
        "ds_read2_b64 v[40:43], %[je] offset1:1\n\t"                                                                           \
        TRIPS                                                                                                                  \
        MCL_SW_STEP_REC("%[xb]", "%[yb]", "%[xa]", "%[ya]")                                                                    \
        MCL_SW_TAIL_R_LAST("24")                                                                                               \
        : [acc] "+v"(acc_fast), [je] "+v"(je), [j8b] "+v"(j8b), [g] "+v"(gwalk), [xa] "+v"(xa), [ya] "+v"(ya), [xb] "+v"(xb),   \
          [yb] "+v"(yb), [tc] "+s"(tc), [expired] "+s"(expired), [cd] "=&s"(cd) XOUT                                           \
        : [p0x] "v"(P0x), [p0y] "v"(P0y), [rem0] "v"(rem_walk), [s0] "v"(s0e),                                                  \
          [ince] "v"(inc32), [rk] "s"(a.rec_k), [ltb] "s"(a.Ltd), [st8] "s"(st8),                                               \
          [magic1] "s"(6755399441055745.0), [zoff] "s"(zoff), [lb] "n"(kQLdsBase), [wp] "s"(wpitch) __VA_ARGS__                                              \
        : MCL_SW_CLOBBERS)

#define MCL_SW_NO_XOUT
#define MCL_SW_HYB_XOUT , [em] "=&s"(hyb_em)
#define MCL_SW_WALK_REC(NEGX, NEGY) MCL_SW_WALK_REC_T(NEGX, NEGY, MCL_SW_TRIPS_LDS, MCL_SW_NO_XOUT)
#define MCL_SW_WALK_HYB(NEGX, NEGY) MCL_SW_WALK_REC_T(NEGX, NEGY, MCL_SW_TRIPS_LDS MCL_SW_ESCAPE, MCL_SW_HYB_XOUT, , [gx] "s"(hyb_gx), [gy] "s"(hyb_gy), [pitch] "s"(gpitch), [gbase] "s"(a.distg), [dd] "v"(hyb_dd), [dm] "s"(hyb_dm))

// ---- the same walk on the wedge fields in GLOBAL memory (k_rays_sweep<.., GLOBAL>): ranges beyond what a 256-cell LDS window
// holds (cpp:195 puts no bound on MAX_RANGE_PX).  The fields are read in place from copies that are MIRRORED per quadrant like
// the LDS windows (RayArgs::distg: every ray runs towards +x, +y; direction components are unsigned), each with a two-cell ring of
// stop bytes and a tail of stop rows behind it.  There is no window: the cell dwords of a position are the lane's ABSOLUTE
// row and byte offset in the allocation (the field's offset rides in the column dword), the address of a cell is
// row * pitch + column from the start of the allocation: one v_mad_u32_u24 -- five VALU per trip here as well (round 4: ten),
// and no bound on the range from the position format.  Memory safety does not rest on the rays being valid: a walk advances by at most P
// samples in total (a skip larger than the samples left ends it before the jump is made), so from a cell of the ringed grid it can only form
// addresses inside its field's tail, whatever bytes it reads (mcl_set_map sizes the tail; tests/test_sweep_addressing.py).
// A trip's latency (L1 / L2 hits: the fields around the cloud stay resident) is covered by the other seven waves of the SIMD.
#define MCL_SWG_TRIP(REM, GIN, BYIN, TXIN, TYIN, TX, TY, TXLO, TXHI, TYLO, TYHI, AD, BY, GOUT, REMOUT, XX, XY, PITCH, BASE) \
    "v_mad_u64_u32 " TX ", vcc, " BYIN ", " XX ", " TXIN "\n\t"                                           \
    "v_mad_u64_u32 " TY ", vcc, " BYIN ", " XY ", " TYIN "\n\t"                                           \
    "v_mad_u32_u24 " AD ", " TYHI ", " PITCH ", " TXHI "\n\t"                                             \
    "global_load_sbyte " BY ", " AD ", " BASE "\n\t"                                                      \
    "v_min3_u32 " GOUT ", " GIN ", " TXLO ", " TYLO "\n\t"                                                \
    "s_waitcnt vmcnt(0)\n\t"                                                                              \
    "v_sub_co_u32 " REMOUT ", vcc, " REM ", " BY "\n\t"                                                   \
    "s_andn2_b64 exec, exec, vcc\n\t"

// (memory reads return in order: the trip's vmcnt(0) also lands the next beam's direction and the previous beam's table entry,
//  which were requested before it -- the vmcnt waits of the LDS walk are satisfied trivially here)
#define MCL_SWG_TRIP_FIRST MCL_SWG_TRIP("%[rem0]", "%[g]", "%[s0]", "%[p0x]", "%[p0y]", "v[52:53]", "v[54:55]", "v52", "v53", "v54", "v55", "v48", "v49", "%[g]", "v57", "v44", "v46", "%[pitch]", "%[gbase]")
#define MCL_SWG_TRIP_NEXT MCL_SWG_TRIP("v57", "%[g]", "v49", "v[52:53]", "v[54:55]", "v[52:53]", "v[54:55]", "v52", "v53", "v54", "v55", "v48", "v49", "%[g]", "v57", "v44", "v46", "%[pitch]", "%[gbase]")
#define MCL_SW_TRIPS_GLB                                                                                                       \
        "s_mov_b32 %[cd], %[cdinit]\n\t"                                                                                       \
        MCL_SWG_TRIP_FIRST                                                                                                     \
        "s_cbranch_execz 2f\n"                                                                                                 \
        "1:\n\t"                                                                                                               \
        MCL_SWG_TRIP_NEXT "s_cbranch_execz 2f\n\t"                                                                              \
        MCL_SWG_TRIP_NEXT "s_cbranch_execz 2f\n\t"                                                                              \
        MCL_SWG_TRIP_NEXT "s_cbranch_execz 2f\n\t"                                                                              \
        MCL_SWG_TRIP_NEXT "s_cbranch_execz 2f\n\t"                                                                              \
        "s_sub_u32 %[cd], %[cd], 1\n\t"                                                                                        \
        "s_cbranch_scc0 1b\n\t"                                                                                                \
        "s_mov_b32 %[expired], 1\n"                                                                                            \
        "2:\n\t"                                                                                                               \
        "s_mov_b64 exec, -1\n\t"

// HYBRID: inside the window samples left count against the lane's BUDGET (the table offset of the lane carries the deficit
// %[dd] = range - budget, so a stop inside the window finds its row without an instruction).  The lanes whose ray spent the budget
// without a stop (samples left negative -- a skip beyond what was left -- in a lane whose budget was cut short: the mask %[dm]; the
// others ended at a stop or ran out of RANGE) go on in the mirrored wedge fields in global memory: samples left of the true range
// restored (the borrow undone, the deficit back), the cell dwords moved from the window's frame to the ringed field's (%[gx] carries
// the field's byte offset as well), and a first trip that advances by zero -- it reads, from global memory, the cell the ray stands
// in.  Back in the window's terms afterwards: a stop keeps its samples left minus the deficit, "no stop in range" becomes the row
// just below the deficit.  The usual case -- nobody left the window -- costs one compare and a branch per ray.
#define MCL_SW_ESCAPE                                                                                                          \
        "v_cmp_gt_i32 vcc, 0, v57\n\t"                                                                                         \
        "s_and_b64 vcc, vcc, %[dm]\n\t"                                                                                        \
        "s_cbranch_scc0 7f\n\t"                                                                                                \
        "s_mov_b64 %[em], vcc\n\t"                                                                                             \
        "s_mov_b64 exec, vcc\n\t"                                                                                              \
        "v_add3_u32 v57, v57, v49, %[dd]\n\t"                                                                                  \
        "v_add_u32 v53, %[gx], v53\n\t"                                                                                        \
        "v_add_u32 v55, %[gy], v55\n\t"                                                                                        \
        "v_mov_b32 v49, 0\n\t"                                                                                                 \
        "s_movk_i32 %[cd], 0x7fff\n" /* (a bound for a malformed field only: the samples left end every walk) */                 \
        "6:\n\t"                                                                                                               \
        MCL_SWG_TRIP_NEXT "s_cbranch_execz 8f\n\t"                                                                              \
        "s_sub_u32 %[cd], %[cd], 1\n\t"                                                                                        \
        "s_cbranch_scc0 6b\n\t"                                                                                                \
        "s_mov_b32 %[expired], 1\n"                                                                                            \
        "8:\n\t"                                                                                                               \
        "s_mov_b64 exec, %[em]\n\t"                                                                                            \
        "v_cmp_lt_i32 vcc, 0, v57\n\t"                                                                                         \
        "v_sub_u32 v57, v57, %[dd]\n\t"                                                                                        \
        "v_not_b32 v48, %[dd]\n\t"                                                                                             \
        "v_cndmask_b32 v57, v48, v57, vcc\n\t"                                                                                 \
        "s_mov_b64 exec, -1\n"                                                                                                 \
        "7:\n\t"

#define MCL_SWG_WALK_TAB(NEGA, NEGB)                                                                                           \
    asm volatile(                                                                                                              \
        "global_load_dwordx4 v[40:43], %[j16], %[csb]\n\t"                                                                     \
        "v_mov_b32 v48, %[zoff]\n\t"                                                                                           \
        "global_load_dwordx2 v[50:51], v48, %[ltb]\n\t"                                                                        \
        "3:\n\t"                                                                                                               \
        MCL_SW_DIR_TAB(NEGA, NEGB)                                                                                             \
        MCL_SW_TRIPS_GLB                                                                                                       \
        MCL_SW_TAIL("s_waitcnt vmcnt(1)\n\t")                                                                                  \
        : [acc] "+v"(acc_fast), [j16] "+v"(j16), [j8b] "+v"(j8b), [g] "+v"(gwalk),                                              \
          [tc] "+s"(tc), [expired] "+s"(expired), [cd] "=&s"(cd)                                                               \
        : [aq] "v"(aq), [bq] "v"(bq), [p0x] "v"(P0x), [p0y] "v"(P0y), [rem0] "v"(rem_start), [s0] "v"(s0e),                     \
          [inc16] "v"(inc16), [inc8] "v"(inc8), [csb] "s"(a.beam_csx), [ltb] "s"(a.Ltd), [st8] "s"(st8),                        \
          [pitch] "s"(gpitch), [gbase] "s"(a.distg), [cdinit] "s"(cdinit4),                                                     \
          [magic1] "s"(6755399441055745.0), [zoff] "s"(zoff)                                                                   \
        : MCL_SW_CLOBBERS)

// ---- TWO rays per lane (REC walks with an even number of slots): beams j and j + 1 of the lane's particle march in the same
// trip loop.  The walk is bound by the latency of its dependent chain (mad -> address -> LDS read -> wait -> borrow -> exec ->
// branch: ~150 cycles per trip and wave, eight waves per SIMD cover 8 x 20 issue cycles of it; SQ_WAIT_ANY = 70 % of the wave
// cycles, profiles/r05_*), so a second, independent chain in the same wave runs in the shadow of the first
// (tools/ubench/trip_rates.hip: 19.4 -> 15.7 SIMD cycles per ray and trip).  Ray A lives under exec, ray B under the lane mask
// %[mb]; a ray that has ended is masked off (its samples-left register keeps its final value) while its sibling goes on; the
// loop ends when no lane has a ray left.  Registers: v[42:43] / v[44:45] pending table entries; v46 / v48 direction of ray A
// (v47 its LDS address, then cell byte), v50 / v52 of ray B (v49 address / byte); v51 / v53 samples left of A / B;
// v[54:57] position of A, v[58:61] of B (before the trips: the two beams' offsets from the grid, FMA scratch).
#define MCL_SW2_PROLOGUE(NEGX, NEGY)                                                                                            \
        "ds_read2_b64 v[54:57], %[je] offset1:1\n\t"                                                                           \
        "s_waitcnt lgkmcnt(0)\n\t"                                                                                             \
        "v_add_f64 v[58:59], %[xa], %[magic1]\n\t"                                                                             \
        "v_fma_f64 v[46:47], " NEGX "v[54:55], %[ya], v[58:59]\n\t"                                                            \
        "v_add_f64 v[58:59], %[ya], %[magic1]\n\t"                                                                             \
        "v_fma_f64 v[48:49], " NEGY "v[54:55], %[xa], v[58:59]\n\t"                                                            \
        "v_add_f64 v[58:59], %[xb], %[magic1]\n\t"                                                                             \
        "v_fma_f64 v[50:51], " NEGX "v[56:57], %[yb], v[58:59]\n\t"                                                            \
        "v_add_f64 v[58:59], %[yb], %[magic1]\n\t"                                                                             \
        "v_fma_f64 v[52:53], " NEGY "v[56:57], %[xb], v[58:59]\n\t"                                                            \
        MCL_SW_STEP_REC("%[xa]", "%[ya]", "%[xb]", "%[yb]")                                                                    \
        MCL_SW_STEP_REC("%[xb]", "%[yb]", "%[xa]", "%[ya]")                                                                    \
        "v_add_u32 %[je], %[je], %[ince]\n\t"

// one trip of both rays.  READ_A / READ_B: address + read of the cell byte (LDS window or global field); WAITALL: both bytes here
#define MCL_SW2_TRIP(REMA, REMB, BYA, BYB, TAX, TAY, TBX, TBY, READ_A, READ_B, WAITALL, SWITCH_B, SWITCH_A)                     \
        "v_mad_u64_u32 v[54:55], vcc, " BYA ", v46, " TAX "\n\t"                                                               \
        "v_mad_u64_u32 v[56:57], vcc, " BYA ", v48, " TAY "\n\t"                                                               \
        READ_A                                                                                                                 \
        "v_min3_u32 %[g], %[g], v54, v56\n\t"                                                                                  \
        SWITCH_B                                                                                                               \
        "v_mad_u64_u32 v[58:59], vcc, " BYB ", v50, " TBX "\n\t"                                                               \
        "v_mad_u64_u32 v[60:61], vcc, " BYB ", v52, " TBY "\n\t"                                                               \
        READ_B                                                                                                                 \
        "v_min3_u32 %[g], %[g], v58, v60\n\t"                                                                                  \
        WAITALL                                                                                                                \
        "v_sub_co_u32 v53, vcc, " REMB ", v49\n\t"                                                                             \
        "s_andn2_b64 %[mb], exec, vcc\n\t"                                                                                     \
        SWITCH_A                                                                                                               \
        "v_sub_co_u32 v51, vcc, " REMA ", v47\n\t"                                                                             \
        "s_andn2_b64 exec, exec, vcc\n\t"                                                                                      \
        "s_or_b64 %[st], exec, %[mb]\n\t"                                                                                      \
        "s_cbranch_scc0 2f\n\t"
#define MCL_SW2_READ_A_LDS "v_mad_u32_u24 v47, v57, %[wp], v55\n\t" "ds_read_i8 v47, v47 offset:%[lb]\n\t"
#define MCL_SW2_READ_B_LDS "v_mad_u32_u24 v49, v61, %[wp], v59\n\t" "ds_read_i8 v49, v49 offset:%[lb]\n\t"
#define MCL_SW2_READ_A_GLB "v_mad_u32_u24 v47, v57, %[pitch], v55\n\t" "global_load_sbyte v47, v47, %[gbase]\n\t"
#define MCL_SW2_READ_B_GLB "v_mad_u32_u24 v49, v61, %[pitch], v59\n\t" "global_load_sbyte v49, v49, %[gbase]\n\t"
// (an odd number of slots: the walk's last pair has no ray B -- its mask starts empty and its table row is the all-zero one)
#define MCL_SW2_TRIP_FIRST(RA, RB, W)                                                                                          \
        "s_or_b32 %[cd], %[tc], %[even]\n\t"                                                                                   \
        "s_cmp_eq_u32 %[cd], 0\n\t"                                                                                            \
        "s_cselect_b64 %[mb], 0, -1\n\t"                                                                                       \
        MCL_SW2_TRIP("%[rem0]", "%[rem0]", "%[s0]", "%[s0]", "%[p0x]", "%[p0y]", "%[p0x]", "%[p0y]", RA, RB, W,                 \
                     "s_mov_b64 %[sa], exec\n\t" "s_mov_b64 exec, %[mb]\n\t", "s_mov_b64 exec, %[sa]\n\t")
#define MCL_SW2_TRIP_NEXT(RA, RB, W)                                                                                           \
        MCL_SW2_TRIP("v51", "v53", "v47", "v49", "v[54:55]", "v[56:57]", "v[58:59]", "v[60:61]", RA, RB, W,                      \
                     "s_mov_b64 %[sa], exec\n\t" "s_mov_b64 exec, %[mb]\n\t", "s_mov_b64 exec, %[sa]\n\t")
#define MCL_SW2_TRIPS(CDSET, RA, RB, W)                                                                                        \
        MCL_SW2_TRIP_FIRST(RA, RB, W)                                                                                          \
        CDSET                                                                                                                  \
        "1:\n\t"                                                                                                               \
        MCL_SW2_TRIP_NEXT(RA, RB, W)                                                                                           \
        MCL_SW2_TRIP_NEXT(RA, RB, W)                                                                                           \
        MCL_SW2_TRIP_NEXT(RA, RB, W)                                                                                           \
        MCL_SW2_TRIP_NEXT(RA, RB, W)                                                                                           \
        "s_sub_u32 %[cd], %[cd], 1\n\t"                                                                                        \
        "s_cbranch_scc0 1b\n\t"                                                                                                \
        "s_mov_b32 %[expired], 1\n"                                                                                            \
        "2:\n\t"                                                                                                               \
        "s_mov_b64 exec, -1\n\t"                                                                                               \
        "s_or_b32 %[cd], %[tc], %[even]\n\t"                                                                                   \
        "s_cmp_eq_u32 %[cd], 0\n\t"                                                                                            \
        "s_cbranch_scc0 4f\n\t"                                                                                                \
        "v_mov_b32 v53, %[zrow]\n"                                                                                             \
        "4:\n\t"
#define MCL_SW2_EPILOGUE                                                                                                       \
        "s_waitcnt vmcnt(0)\n\t" /* the previous pair's table entries landed */                                                \
        "v_add_f64 %[acc], %[acc], v[42:43]\n\t"                                                                               \
        "v_add_f64 %[acc], %[acc], v[44:45]\n\t"                                                                               \
        "v_mad_i32_i24 v47, v51, %[st8], %[j8b]\n\t"                                                                           \
        "v_mad_i32_i24 v49, v53, %[st8], %[j8b]\n\t"                                                                           \
        "v_add_u32 %[j8b], %[j8b], %[ince]\n\t"                                                                                \
        "global_load_dwordx2 v[42:43], v47, %[ltb]\n\t"                                                                        \
        "global_load_dwordx2 v[44:45], v49, %[ltb] offset:8\n\t"                                                               \
        "s_sub_u32 %[tc], %[tc], 1\n\t"                                                                                        \
        "s_cbranch_scc0 3b\n\t"                                                                                                \
        "s_waitcnt vmcnt(0)\n\t"                                                                                               \
        "v_add_f64 %[acc], %[acc], v[42:43]\n\t"                                                                               \
        "v_add_f64 %[acc], %[acc], v[44:45]"
#define MCL_SW2_CLOBBERS "memory", "vcc", "scc", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61"
#define MCL_SW2_OUTPUTS [acc] "+v"(acc_fast), [je] "+v"(je), [j8b] "+v"(j8b), [g] "+v"(gwalk), [xa] "+v"(xa), [ya] "+v"(ya),      \
          [xb] "+v"(xb), [yb] "+v"(yb),                                                                                        \
          [tc] "+s"(tc), [expired] "+s"(expired), [cd] "=&s"(cd), [mb] "=&s"(x2_mb), [sa] "=&s"(x2_sa), [st] "=&s"(x2_st)
#define MCL_SW2_INPUTS [p0x] "v"(P0x), [p0y] "v"(P0y), [rem0] "v"(rem_start), [s0] "v"(s0e), [ince] "v"(inc16), [rk] "s"(a.rec_k),   \
          [ltb] "s"(a.Ltd), [st8] "s"(st8), [magic1] "s"(6755399441055745.0), [zoff] "s"(zoff), [zrow] "s"(zrow), [even] "s"(x2_even)

#define MCL_SW2_WALK(NEGX, NEGY)                                                                                               \
    asm volatile(                                                                                                              \
        "v_mov_b32 v47, %[zoff]\n\t"                                                                                           \
        "global_load_dwordx2 v[42:43], v47, %[ltb]\n\t" /* 0.0 twice: the pattern of the steady state */                       \
        "global_load_dwordx2 v[44:45], v47, %[ltb]\n\t"                                                                        \
        "3:\n\t"                                                                                                               \
        MCL_SW2_PROLOGUE(NEGX, NEGY)                                                                                           \
        MCL_SW2_TRIPS("s_movk_i32 %[cd], 75\n\t", MCL_SW2_READ_A_LDS, MCL_SW2_READ_B_LDS, "s_waitcnt lgkmcnt(0)\n\t")           \
        MCL_SW2_EPILOGUE                                                                                                       \
        : MCL_SW2_OUTPUTS                                                                                                      \
        : MCL_SW2_INPUTS, [lb] "n"(kQLdsBase), [wp] "s"(wpitch)                                                                                  \
        : MCL_SW2_CLOBBERS)

#define MCL_SWG2_WALK(NEGX, NEGY)                                                                                              \
    asm volatile(                                                                                                              \
        "v_mov_b32 v47, %[zoff]\n\t"                                                                                           \
        "global_load_dwordx2 v[42:43], v47, %[ltb]\n\t"                                                                        \
        "global_load_dwordx2 v[44:45], v47, %[ltb]\n\t"                                                                        \
        "3:\n\t"                                                                                                               \
        MCL_SW2_PROLOGUE(NEGX, NEGY)                                                                                           \
        MCL_SW2_TRIPS("s_mov_b32 %[cd], %[cdinit]\n\t", MCL_SW2_READ_A_GLB, MCL_SW2_READ_B_GLB, "s_waitcnt vmcnt(0)\n\t")       \
        MCL_SW2_EPILOGUE                                                                                                       \
        : MCL_SW2_OUTPUTS                                                                                                      \
        : MCL_SW2_INPUTS, [pitch] "s"(gpitch), [gbase] "s"(a.distg), [cdinit] "s"(cdinit4)                                      \
        : MCL_SW2_CLOBBERS)

// REC: the walk turns the beam direction instead of fetching it (MCL_SW_INTS_REC, MCL_SW_STEP_REC); PAIRS (with REC): two rays per lane (MCL_SW2_*).
// Measured at 4M x 1081 (profiles/r05_experiments): REC -11 % everywhere; PAIRS on top of it -6 % in the global-field form and
// -11 % on the uniform cloud of a first update (both wait for memory), +-0 on the tracking cloud in LDS windows and +2 % on the
// levine stand-in (both bound by VALU issue by then): the host asks for PAIRS in the global-field form and for a freshly
// initialised set.
// HYB (with REC, without GLOBAL / PAIRS): ranges beyond the window -- the walk runs in the LDS window as far as that reaches and
// goes on in the global wedge fields where a ray leaves it (MCL_SW_ESCAPE).  At 479 px (a 0.025 m map, 12 m) most rays end inside.
template <bool COUNT, bool GLOBAL = false, bool REC = false, bool PAIRS = false, bool HYB = false>
__global__ __launch_bounds__(kRayThreads, 8) void k_rays_sweep(RayArgs a)
{
    static_assert(!HYB || (REC && !GLOBAL && !PAIRS), "the hybrid form is the turned-direction walk in LDS windows");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    __shared__ int item_sh;
    __shared__ unsigned int fixn_sh;                           // entries this workgroup appended to ITS segment of the fix-up list
    __shared__ unsigned int chunk_sh;                          // next 64-slot chunk of the current (run, wedge) pass
    const int lane = threadIdx.x & 63;
    unsigned long long cnt_probe = 0;
    const int G = a.sweep_g;                                   // wedges per work item (divides kWedges)
    const int nitems = a.nitems_ptr ? a.nitems_ptr[0] : a.nitems;       // the plan is made on the device (k_sweep_plan)
    // The probe trip addresses the window from the raw LDS offset kQLdsBase.  mcl_create checks the layout on the host and keeps
    // AUTO / MCL_RAYS_SWEEP off this kernel when it differs (choose_ray_mode), so this branch is not reachable through the ABI;
    // should a toolchain ever lay the static words out differently anyway, the launch reports a full fix-up list -- which the
    // host answers by re-running the stage with k_rays_skip -- instead of aborting the device.
    if ((!GLOBAL || REC) && (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)lds_raw != (uint32_t)kQLdsBase) {
        if (threadIdx.x == 0) a.fix_count[(size_t)blockIdx.x * 8] = a.fix_cap + 1ull;
        return;
    }
    // extent of the mirrored frame in cells: the 256-cell LDS window, or (GLOBAL) the ringed field (its two axes differ)
    const int Sx = GLOBAL ? a.Wp + 4 : kSwSide, Sy = GLOBAL ? a.Hp + 4 : kSwSide;
    const uint32_t guard_units = sweep_guard_units(a.P);       // level-1 error bound per axis, units of 2^-32 px
    const uint32_t gthresh = a.force_exact ? 0xFFFFFFFFu : 2u * guard_units;
    const uint32_t gpitch = (uint32_t)a.distg_pitch;           // GLOBAL: row pitch of the ringed wedge fields
    const uint32_t cdinit = (uint32_t)a.P + 64u;               // GLOBAL: trip countdown of the per-slot loop (a walk makes at most P trips)
    const uint32_t cdinit4 = cdinit / 4u + 1u;                 // ... and of the asm walk's loop, four trips per turn
    const uint32_t st8 = (uint32_t)__builtin_amdgcn_readfirstlane(a.ltd_cols * 8);
    [[maybe_unused]] const uint32_t wpitch = (uint32_t)kSwPitch;      // the window's row pitch as a scalar operand of the probe trip's address
    const uint32_t zrow = (uint32_t)(a.P + 1);                 // "samples left" that selects the all-zero row
    const unsigned char *ldsb = lds_raw;
    // REC: every beam's offset from the scan's angular grid (RayArgs::beam_err, 8 bytes per table column), behind the window; the
    // walk reads it from the raw LDS offset `ebase` (published by the first barrier of the item loop, like fixn_sh)
    constexpr uint32_t ebase = (uint32_t)kQLdsBase + (GLOBAL ? 0u : (uint32_t)kSwWinBytes);
    if (REC) {
        double *etab = reinterpret_cast<double *>(lds_raw + (GLOBAL ? 0 : kSwWinBytes));
        for (int c = threadIdx.x; c < a.ltd_cols; c += kRayThreads) etab[c] = a.beam_err[c];
    }
    if (threadIdx.x == 0) fixn_sh = 0u;                        // published by the first barrier of the item loop
    // the segment belongs to this workgroup alone: the append counter lives in LDS and entries are plain stores (the list
    // is read by k_rays_fix, a later kernel); the count goes to memory once, at the end
    unsigned long long *const fix_seg = a.fix_list + (size_t)blockIdx.x * a.fix_cap;
    unsigned long long dbg_t0 = 0, dbg_wait = 0, dbg_bar = 0;               // MCL_DEBUG_WG: start / end stamps (100 MHz) and items of this workgroup
    unsigned int dbg_items = 0;
    if (a.dbg) dbg_t0 = __builtin_amdgcn_s_memrealtime();
    for (;;) {
    __syncthreads();
    if (threadIdx.x == 0) item_sh = (int)atomicAdd(a.work_counter, 1ull);
    __syncthreads();
    const int item = item_sh;
    if (item >= nitems) break;
    ++dbg_items;
    // (first unit, units, wedge group): the host lists big items first and single units last (guided schedule), so the
    // persistent workgroups finish within one small item of each other
    const int4 it = a.items[item];                                   // wave-uniform: scalar loads
    const int grp = it.z;
    // units are runs of the sorted order (k_unit_table); slots stay below 2^27: 32-bit slot arithmetic (registers are scarce here)
    const uint32_t p_begin = a.unit_begin[it.x];
    const uint32_t p_end = a.unit_begin[it.x + it.y];
    if (p_begin >= p_end) continue;
    const int4 ctr = a.centres[it.w];                                  // window centre of the run (k_sweep_plan): padded cell
    for (int gw = 0; gw < G; ++gw) {
    const int kbin = grp * G + gw;
    const int q = kbin >> kWedgeShift;
    const int sxp = (q == 0 || q == 3), syp = (q == 0 || q == 1);      // the wedge's rays run up (true) or down the axis
    const bool negy = sxp != syp;
    const int mlo = 3;
    int wx0, wy0;
    // GLOBAL: no window.  Field kbin of RayArgs::distg is mirrored like a window would be, and a position's cell dwords are the
    // absolute row and (field offset + column) in the allocation: nothing a particle could miss but the ringed grid itself
    const uint32_t foff = GLOBAL ? (uint32_t)((size_t)kbin * a.distg_stride) : 0u;     // < 2^32: checked by mcl_set_map
    [[maybe_unused]] uint32_t hyb_gx = 0u, hyb_gy = 0u;                  // HYB: window frame -> ringed field (set with the window's origin)
    if (GLOBAL) {
        wx0 = -2; wy0 = -2;                                                // ringed frame: column c = padded cell c - 2
        if (gw > 0) __syncthreads();                                       // every wave is done with the previous pass's chunk counter
        if (threadIdx.x == 0) chunk_sh = 0u;
        __syncthreads();
    } else {
        constexpr int S = kSwSide;
        const int Pw = HYB ? kSwHybReach : a.P;                            // the range the window is laid out for
        const int E = S - (Pw + 2) - mlo;
        const int back = E / 2 + mlo;
        const int cxm = ctr.x, cym = ctr.y;
        wx0 = sxp ? cxm - back : cxm + back - S;
        wy0 = syp ? cym - back : cym + back - S;
        if (HYB) {
            // column c of the mirrored window is column c + (wx0 + 2) of the mirrored ringed field where the rays run up the axis,
            // c + (Wp - (S - 2) - wx0) where they run down (the two frames differ by whole cells: positions keep their
            // fractions); the field's byte offset in the allocation rides in the column like in the global form
            hyb_gx = (uint32_t)(sxp ? wx0 + 2 : a.Wp - (S - 2) - wx0) + (uint32_t)((size_t)kbin * a.distg_stride);
            hyb_gy = (uint32_t)(syp ? wy0 + 2 : a.Hp - (S - 2) - wy0);
        }
        uint32_t *win = reinterpret_cast<uint32_t *>(lds_raw);             // rows kSwPitch bytes apart, written as dword pairs
        const uint8_t *fieldq = a.distw + (size_t)kbin * a.distw_stride;   // only stops a wedge-kbin ray can reach bound its jumps
        constexpr int wpr = S >> 3, nwords = wpr * S;
        unsigned long long dbg_w0 = 0;
        if (a.dbg) dbg_w0 = __builtin_amdgcn_s_memrealtime();
        if (gw > 0) __syncthreads();                                       // every wave is done with the previous window
        if (a.dbg) { const unsigned long long tq = __builtin_amdgcn_s_memrealtime(); dbg_bar += tq - dbg_w0; }
        for (int wi = threadIdx.x; wi < nwords; wi += kRayThreads) {
            const int row = wi / wpr, cw = wi - row * wpr;
            // LDS cell (row, col) = grid cell (wy0 + row, wx0 + col), mirrored in the axes along which the wedge's rays
            // run backwards: word cw of a mirrored row is the source word at the far end with its bytes reversed
            const int gy = syp ? wy0 + row : wy0 + (S - 1) - row;
            const int gx = sxp ? wx0 + cw * 8 : wx0 + (S - 8) - cw * 8;
            // the wedge fields are stored in the LDS encoding (stop = 0xFF, skips 1..127); outside the grid is stop
            uint64_t b8 = ~0ull;
            if (gy >= 0 && gy < a.Hp) {
                const uint8_t *rowp = fieldq + (size_t)gy * a.Wps;
                if (gx >= 0 && gx + 8 <= a.Wp) {
                    b8 = *reinterpret_cast<const uint64_t *>(rowp + gx);
                } else {
                    for (int k = 0; k < 8; ++k)
                        if (gx + k >= 0 && gx + k < a.Wp) b8 = (b8 & ~(0xFFull << (8 * k))) | ((uint64_t)rowp[gx + k] << (8 * k));
                }
            }
            const uint64_t w8 = sxp ? b8 : __builtin_bswap64(b8);
            uint32_t *wp = win + row * (kSwPitch / 4) + cw * 2;
            wp[0] = (uint32_t)w8; wp[1] = (uint32_t)(w8 >> 32);
        }
        if (threadIdx.x == 0) chunk_sh = 0u;
        __syncthreads();
        if (a.dbg) dbg_wait += __builtin_amdgcn_s_memrealtime() - dbg_w0;
    }

    // The waves take 64-slot chunks of the run from a counter instead of a fixed stride: a wave whose particles had short rays
    // takes another chunk while the slow ones finish, so the barrier before the next window waits for the slowest CHUNK, not
    // for the slowest of sixteen fixed shares (the window phase, barrier waits included, was 9 % of a workgroup's time)
    // A pass of at most eight chunks (the runs of a sparse cloud cut at tile or bucket borders) would leave half of the sixteen
    // waves or more without work: its chunks are handed out in pieces -- the same 64 slots, 1/parts of their beams -- so that
    // chunks x parts ~ 16.  A piece repeats the per-particle setup and its load latencies (about a tenth of a chunk), which is
    // why longer passes are left alone (quartering the last 16 chunks of every pass: 3 % slower on the tracking cloud).
    const uint32_t nchunks = (p_end - p_begin + 63u) >> 6;
    const bool want_steps = a.steps != nullptr || a.steps16 != nullptr;
    // (split16: passes of nine to sixteen chunks -- one per wave, nothing for a quick wave to take over -- are halved as well;
    //  set by the host for small launches, where a workgroup sees a handful of items and waits at the end of every one)
    const uint32_t parts_all = (COUNT || want_steps || nchunks > (a.split16 ? 16u : 8u)) ? 1u : nchunks > 4u ? 2u : nchunks > 2u ? 4u : nchunks > 1u ? 8u : 16u;
    const uint32_t npieces = nchunks * parts_all;
    for (;;) {
        uint32_t piece = 0;
        if (lane == 0) piece = atomicAdd(&chunk_sh, 1u);
        piece = (uint32_t)__builtin_amdgcn_readfirstlane((int)piece);
        if (piece >= npieces) break;
        // piece p = part p / nchunks of chunk p % nchunks: the first round gives every chunk's first part to a different wave
        const uint32_t chunk = parts_all > 1u ? piece % nchunks : piece, part = parts_all > 1u ? piece / nchunks : 0u, parts = parts_all;
        const uint32_t s0g = p_begin + chunk * 64u;
        const uint32_t slot = s0g + (uint32_t)lane;
        const bool have = slot < p_end;
        const uint32_t sl = have ? slot : p_end - 1u;
        const double4 pci = a.pcs[sl];
        // beams of this particle in wedge kbin: [ja, jb) and, for scans wider than a turn minus one wedge, [ja2, B)
        int ja = 0, jb = 0, ja2 = a.B;
        {
            const double th = a.ths[sl];
            if (th == th && fabs(th) < 1e6) {
                const int w0 = beam_wedge_d(th, a.beam_a0), wl = beam_wedge_d(th, a.beam_alast);      // (the scan's first and last angle as doubles: no loads)
                const int m = w0 + ((kbin - w0) & (kWedges - 1));
                if (m <= wl) {
                    ja = m == w0 ? 0 : first_beam_in_wedge<REC>(th, a.beam_angle, a.B, m, a.beam_a0, a.beam_inv_inc);
                    jb = m == wl ? a.B : first_beam_in_wedge<REC>(th, a.beam_angle, a.B, m + 1, a.beam_a0, a.beam_inv_inc);
                    if (m + kWedges <= wl) ja2 = first_beam_in_wedge<REC>(th, a.beam_angle, a.B, m + kWedges, a.beam_a0, a.beam_inv_inc);
                }
            } else if (kbin == 0) {
                jb = a.B;            // garbage heading: one range in quadrant 0 like k_particle_prep (position is NaN -> far path)
            }
        }
        int n1 = jb > ja ? jb - ja : 0, n2 = a.B > ja2 ? a.B - ja2 : 0;
        if (!have) { n1 = 0; n2 = 0; }
        // position in the (unmirrored) frame: column c of the frame = padded cell wx0 + c
        const double wpx = pci.z - (double)(wx0 - 1), wpy = pci.w - (double)(wy0 - 1);
        bool inwin;
        if (GLOBAL) {
            // the particle's own padded cell must exist in the field (NaN fails the comparisons)
            inwin = pci.z >= -1.0 && pci.z < (double)(a.Wp - 1) && pci.w >= -1.0 && pci.w < (double)(a.Hp - 1);
        } else {
            const double fwd = (double)((HYB ? kSwHybReach : a.P) + 2), bwd = 2.0;
            const bool inx = sxp ? (wpx - bwd >= 0.0 && wpx + fwd < (double)Sx) : (wpx - fwd >= 0.0 && wpx + bwd < (double)Sx);
            const bool iny = syp ? (wpy - bwd >= 0.0 && wpy + fwd < (double)Sy) : (wpy - fwd >= 0.0 && wpy + bwd < (double)Sy);
            inwin = inx && iny;
            // (HYB: a ray may go on in the global fields, whose walk starts from a cell of the ringed grid: the particle's own padded
            //  cell must exist there as well, like in the global-field form)
            if (HYB) inwin = inwin && pci.z >= -1.0 && pci.z < (double)(a.Wp - 1) && pci.w >= -1.0 && pci.w < (double)(a.Hp - 1);
        }
        uint32_t i = 0xFFFFFFFFu;                                  // particle index, loaded by the rare paths that need it
        if (!inwin && n1 + n2 > 0) {                           // not in this window (or NaN): k_rays_far does this pair
            // flags, lists and sums are slot-indexed; the first flag of a slot also lists it for k_rays_far
            if (atomicOr(reinterpret_cast<unsigned int *>(a.far_flags) + sl, 1u << (8 * q)) == 0u) {
                const unsigned long long k = atomicAdd(a.far_count, 1ull);
                if (k < (unsigned long long)a.n) a.far_list[k] = (uint32_t)sl;
            }
            n1 = 0; n2 = 0;
        }
        const int total = n1 + n2;
        // a lane without rays gets a zero direction and zero samples from the window's cell (2, 2): every probe it
        // makes reads that cell, whose byte is never 0, and leaves the loop at once; its table column is the zero column B
        const bool live = total > 0;
        // position in the mirrored frame (the rays of the wedge run towards +x, +y there)
        const double dpos = 2.5;
        // (x or S - x by one FMA with wave-uniform factors: the same single rounding as the subtraction)
        const double lpx = live ? __builtin_fma(wpx, sxp ? 1.0 : -1.0, sxp ? 0.0 : (double)Sx) : dpos;
        const double lpy = live ? __builtin_fma(wpy, syp ? 1.0 : -1.0, syp ? 0.0 : (double)Sy) : dpos;
        // origin in the mirrored frame in 2^-32 px, biased by the guard (see MCL_SW_TRIP), rounded to nearest by the 2^52 magic add:
        // fraction in the low dword, cell in the high one (GLOBAL: plus the field's byte offset in the column dword).  The cell
        // and the guard test of the origin come from the same number: an origin within the guard of a cell boundary (where the
        // biased cell may be the neighbour's) has a low dword below twice the guard like any such sample -- its walk goes to the
        // fix-up list whatever it read.
        const double m0x = __builtin_fma(lpx, 4294967296.0, (double)guard_units) + 4503599627370496.0;
        const double m0y = __builtin_fma(lpy, 4294967296.0, (double)guard_units) + 4503599627370496.0;
        const uint32_t lox = (uint32_t)__double2loint(m0x), loy = (uint32_t)__double2loint(m0y);
        const uint32_t cx0 = (uint32_t)__double2hiint(m0x) & 0xFFFFFu, cy0 = (uint32_t)__double2hiint(m0y) & 0xFFFFFu;
        int d0;
        if (GLOBAL) d0 = (int)a.distg[(size_t)foff + (size_t)cy0 * gpitch + cx0];      // the own cell (a dead lane: cell (2, 2) of the ringed field)
        else d0 = ldsb[(cy0 & (kSwSide - 1)) * (uint32_t)kSwPitch + (cx0 & (kSwSide - 1))];
        const int s0 = (d0 > 127 || d0 < 1) ? 1 : d0;               // own cell is a stop: first sample one step away
        // no stop within range: look at sample P only (it is free: the skip says so), which ends the walk with "no hit"
        const uint32_t s0e = (uint32_t)(s0 <= a.P ? s0 : a.P);
        const uint32_t g0 = lox < loy ? lox : loy;
        const unsigned long long P0x = ((unsigned long long)(cx0 + foff) << 32) | lox;
        const unsigned long long P0y = ((unsigned long long)cy0 << 32) | loy;
        const int rem_start = live ? a.P - (int)s0e : 0;
        // HYB: the budget of this lane's walk in the window -- after fewer samples than this no ray of the wedge has reached the
        // window's last row or column: a sample moves a ray by at most dx_max / dy_max cells (the wedge's bounds in the mirrored
        // frame; the beams' offsets from the grid and the roundings are far inside the cell of margin) -- and what it leaves of
        // the range
        [[maybe_unused]] int rem_walk = rem_start, hyb_dd = 0;
        [[maybe_unused]] unsigned long long hyb_dm = 0ull;
        if (HYB) {
            const int wq = kbin & (kWedges / 4 - 1), wm = (q & 1) ? (kWedges / 4 - 1) - wq : wq;      // the wedge's place in the mirrored quadrant
            const double bx = ((double)(kSwSide - 1) - lpx) * kSwHybInvDx[wm], by = ((double)(kSwSide - 1) - lpy) * kSwHybInvDy[wm];
            const double bmin = bx < by ? bx : by;
            const int budget = bmin >= (double)a.P ? a.P : (bmin > (double)s0e ? (int)bmin : (int)s0e);      // (inside the window a lane has kSwHybReach + 2 cells ahead: far more than a skip)
            hyb_dd = live ? a.P - budget : 0;
            rem_walk = live ? budget - (int)s0e : 0;
            hyb_dm = __ballot(hyb_dd > 0);
        }
        // direction components in the mirrored frame: Xx = aq cb - bq sb, Xy = +-(bq cb + aq sb), both >= 0
        const double dsc = sxp ? kSwDirScale : -kSwDirScale;
        const double aq = live ? pci.x * dsc : 0.0, bq = live ? pci.y * dsc : 0.0;
        // A lane whose scan BEGINS or ENDS inside this wedge has fewer beams here than the lanes whose scan covers the wedge
        // (with sixty-four headings 20 degrees apart -- the chunks of a uniform cloud -- anything from 1 to a full wedge's 90),
        // and the lock-step walk below only runs as far as the shortest lane goes: the rest went through the slow per-slot loop,
        // which made such a chunk several times slower than its fifteen neighbours and the workgroup wait for it.  Such a lane now
        // continues on VIRTUAL beams -- the scan's angular grid continued beyond its ends (host: beam_csx), table columns all
        // zero -- up to the common length: its rays there are walked and add 0.0.  beam_pad = beams of a full wedge (0: scan not
        // evenly spaced or too wide, no padding).
        const bool edge = a.beam_pad > 0 && (n1 + n2 > 0) && n2 == 0 && ((ja == 0) != (jb == a.B));
        // (wave-wide reductions without an LDS round trip: the longest lane and the shortest full one together, the longest
        //  scan-edge lane only in the chunks that have one)
        int tmax = total, tmin = (total > 0 && !edge) ? -total : -0x7fffffff;
        wave_max2_i32(tmax, tmin);
        tmin = -tmin;
        if (__ballot(edge) != 0ull) {         // the common length: the shortest full lane, or the longest edge lane if there is no full one
            const int temax = wave_max_i32(edge ? total : 0);
            const int tstar = tmin != 0x7fffffff ? tmin : temax;
            tmin = tstar < a.beam_pad ? tstar : a.beam_pad;
        }
        const bool padded = edge && total < tmin;
        // a second range only exists for scans wider than three quadrants; the common case steps j by one
        const bool wraps = __ballot(n2 > 0 && n1 > 0) != 0ull;
        const int jfirst = n1 > 0 ? ja : (n2 > 0 ? ja2 : 0);
        const int jwalk = padded && ja == 0 ? jb - tmin : jfirst;       // first (possibly virtual) beam of the lock-step walk
        double acc_fast = 0.0, acc = 0.0;
        uint32_t gwalk = g0;                                       // guard minimum over every sample of the asm walk
        int t_done = 0;
        bool expired_fast = false;
        const bool fast = !COUNT && !want_steps && !wraps && tmax > 0 && tmin > 0;
        if (!fast && part > 0u) continue;                         // the slow forms of a pass are not split: piece 0 does all of it
        int walk_t0 = 0, walk_n = 0;                               // slots [walk_t0, walk_t0 + walk_n) of every lane: this piece's share
        if (fast) {
            walk_t0 = (int)(((uint32_t)tmin * part) / parts);
            walk_n = (int)(((uint32_t)tmin * (part + 1u)) / parts) - walk_t0;
        }
        if (fast && walk_n > 0) {
            // beam_csx and the table's columns are biased by beam_margin entries (virtual beams before beam 0)
            const int jw = jwalk + walk_t0;
            uint32_t j16 = live ? (uint32_t)(jw + a.beam_margin) << 4 : 0u;
            // table column offset, biased by kSwUnder rows so that (samples left) * row bytes + j8b is never negative
            uint32_t j8b = (uint32_t)kSwUnder * st8 + ((uint32_t)((live ? jw : a.B) + a.beam_margin) << 3);
            if (HYB) j8b += (uint32_t)hyb_dd * st8;                   // samples left inside the window count against the lane's budget: rows further down by the deficit
            const uint32_t inc16 = live ? 16u : 0u, inc8 = live ? 8u : 0u;
            const uint32_t inc32 = live ? 32u : 0u;                   // (REC: bytes of table columns / beam offsets per turn of the walk's loop)
            uint32_t tc = (uint32_t)walk_n - 1u, expired = 0u, cd;
            const uint32_t zoff = (zrow + (uint32_t)kSwUnder) * st8;      // any column of the zero row
            if constexpr (REC) {
                // the directions of the GRID angles of the walk's first two beams, scaled and mirrored like aq / bq (MCL_SW_INTS_REC)
                const double2 ci = a.beam_csi[(live ? jw : a.B) + a.beam_margin], cj = a.beam_csi[(live ? jw : a.B) + a.beam_margin + 1];
                double xa = __builtin_fma(aq, ci.x, -(bq * ci.y)), ya = __builtin_fma(bq, ci.x, aq * ci.y);
                double xb = __builtin_fma(aq, cj.x, -(bq * cj.y)), yb = __builtin_fma(bq, cj.x, aq * cj.y);
                if (negy) { ya = -ya; yb = -yb; }
                uint32_t je = ebase + ((uint32_t)((live ? jw : a.B) + a.beam_margin) << 3);
                (void)j16;
                if constexpr (!PAIRS) {
                    static_assert(PAIRS || !GLOBAL || !REC, "the global-field form walks pairs when it turns the directions");
                    if constexpr (HYB) { unsigned long long hyb_em; if (negy) MCL_SW_WALK_HYB("", "-"); else MCL_SW_WALK_HYB("-", ""); }
                    else if constexpr (!GLOBAL) { if (negy) MCL_SW_WALK_REC("", "-"); else MCL_SW_WALK_REC("-", ""); }
                } else {
                    // two rays per lane (MCL_SW2_*): ceil(walk_n / 2) pairs of slots
                    const uint32_t x2_even = (walk_n & 1) ? 0u : 1u;
                    tc = (uint32_t)((walk_n + 1) >> 1) - 1u;
                    unsigned long long x2_mb, x2_sa, x2_st;
                    if constexpr (GLOBAL) { if (negy) MCL_SWG2_WALK("", "-"); else MCL_SWG2_WALK("-", ""); }
                    else { if (negy) MCL_SW2_WALK("", "-"); else MCL_SW2_WALK("-", ""); }
                }
            } else {
                if constexpr (GLOBAL) { if (negy) MCL_SWG_WALK_TAB("-", "-"); else MCL_SWG_WALK_TAB("", ""); }
                else { if (negy) MCL_SW_WALK_TAB("-", "-"); else MCL_SW_WALK_TAB("", ""); }
            }
            expired_fast = expired != 0u;
        }
        if (fast) t_done = part + 1u == parts ? tmin : tmax;       // the ragged rest belongs to the last piece
        // ---- remaining slots (lanes differ in how many rays they have), scans that wrap, probe counting, step output ----
        if (t_done < tmax) {
            // slot t of this lane is beam ja + t for t < n1, then ja2 + (t - n1); past its last slot a lane repeats
            // its last beam (result discarded) so that it stays on a valid in-window ray
            const int jlast = total > 0 ? (n2 > 0 ? a.B - 1 : jb - 1) : 0;
            int j = jfirst + t_done;                                  // t_done > 0 only without a second range
            if (j > jlast) j = jlast;
            if (i == 0xFFFFFFFFu && (want_steps || COUNT)) i = a.perm[sl];
            double2 cs = *reinterpret_cast<const double2 *>(reinterpret_cast<const char *>(a.beam_cs) + ((uint32_t)j << 4));
            double lt_pending = 0.0;
            for (int t = t_done; t < tmax; ++t) {
                acc += lt_pending;
                asm volatile("" : "+v"(acc));
                const bool valid = t < total;
                const int jcur = j;
                {
                    int jn = j + 1;
                    if (wraps && jn == jb && n1 > 0 && t < n1) jn = ja2;    // end of the first range: continue with the second
                    j = min(jn, jlast);
                }
                // (the same two nested FMAs as MCL_SW_ROTATE: >= 0 by construction)
                const uint32_t Xx = (uint32_t)__double2loint(__builtin_fma(aq, cs.x, __builtin_fma(-bq, cs.y, 6755399441055745.0)));
                const uint32_t Xy = negy ? (uint32_t)__double2loint(__builtin_fma(-bq, cs.x, __builtin_fma(-aq, cs.y, 6755399441055745.0)))
                                         : (uint32_t)__double2loint(__builtin_fma(bq, cs.x, __builtin_fma(aq, cs.y, 6755399441055745.0)));
                cs = *reinterpret_cast<const double2 *>(reinterpret_cast<const char *>(a.beam_cs) + ((uint32_t)j << 4));
                int rem;
                uint32_t g;
                bool expired = false;
                // (HYB: the per-slot forms walk in the global fields from the start -- the origin moved into their frame --; the own
                //  cell's skip came from the window, where it is at most what the field says)
                // (a lane without rays stands in cell (2, 2) of the window -- of the ringed FIELD here: the window may lie off the grid)
                [[maybe_unused]] const unsigned long long P0xs = !HYB ? P0x : live ? P0x + ((unsigned long long)hyb_gx << 32)
                                                               : ((unsigned long long)(2u + (uint32_t)((size_t)kbin * a.distg_stride)) << 32) | lox;
                [[maybe_unused]] const unsigned long long P0ys = !HYB ? P0y : live ? P0y + ((unsigned long long)hyb_gy << 32) : (2ull << 32) | loy;
                if (!COUNT && (GLOBAL || HYB)) {
                    unsigned long long saved_exec;
                    uint32_t countdown;
                    asm volatile(
                        "s_mov_b64 %[sv], exec\n\t"
                        "s_mov_b32 %[cd], %[cdinit]\n\t"
                        MCL_SWG_TRIP("%[rem0]", "%[g0]", "%[s0]", "%[p0x]", "%[p0y]", "v[52:53]", "v[54:55]", "v52", "v53", "v54", "v55", "v48", "v49", "%[g]", "%[rem]", "%[xx]", "%[xy]", "%[pitch]", "%[gbase]")
                        "s_cbranch_execz 2f\n"
                        "1:\n\t"
                        MCL_SWG_TRIP("%[rem]", "%[g]", "v49", "v[52:53]", "v[54:55]", "v[52:53]", "v[54:55]", "v52", "v53", "v54", "v55", "v48", "v49", "%[g]", "%[rem]", "%[xx]", "%[xy]", "%[pitch]", "%[gbase]")
                        "s_cbranch_execz 2f\n\t"
                        "s_sub_u32 %[cd], %[cd], 1\n\t"
                        "s_cbranch_scc0 1b\n"
                        "2:\n\t"
                        "s_mov_b64 exec, %[sv]"
                        : [g] "=&v"(g), [rem] "=&v"(rem), [sv] "=&s"(saved_exec), [cd] "=&s"(countdown)
                        : [rem0] "v"(rem_start), [g0] "v"(g0), [s0] "v"(s0e), [xx] "v"(Xx), [xy] "v"(Xy), [p0x] "v"(P0xs), [p0y] "v"(P0ys),
                          [pitch] "s"(gpitch), [gbase] "s"(a.distg), [cdinit] "s"(cdinit)
                        : "memory", "vcc", "scc", "v48", "v49", "v52", "v53", "v54", "v55");
                    expired = __builtin_amdgcn_readfirstlane((int)countdown) < 0;
                } else if (!COUNT) {
                    unsigned long long saved_exec;
                    uint32_t countdown;
                    asm volatile(
                        "s_mov_b64 %[sv], exec\n\t"
                        "s_movk_i32 %[cd], 300\n\t"
                        MCL_SW_TRIP("%[rem0]", "%[g0]", "%[s0]", "%[p0x]", "%[p0y]", "v[52:53]", "v[54:55]", "v52", "v53", "v54", "v55", "v48", "v49", "%[g]", "%[rem]", "%[xx]", "%[xy]", "%[lb]")
                        "s_cbranch_execz 2f\n"
                        "1:\n\t"
                        MCL_SW_TRIP("%[rem]", "%[g]", "v49", "v[52:53]", "v[54:55]", "v[52:53]", "v[54:55]", "v52", "v53", "v54", "v55", "v48", "v49", "%[g]", "%[rem]", "%[xx]", "%[xy]", "%[lb]")
                        "s_cbranch_execz 2f\n\t"
                        "s_sub_u32 %[cd], %[cd], 1\n\t"
                        "s_cbranch_scc0 1b\n"
                        "2:\n\t"
                        "s_mov_b64 exec, %[sv]"
                        : [g] "=&v"(g), [rem] "=&v"(rem), [sv] "=&s"(saved_exec), [cd] "=&s"(countdown)
                        : [rem0] "v"(rem_start), [g0] "v"(g0), [s0] "v"(s0e), [xx] "v"(Xx), [xy] "v"(Xy), [p0x] "v"(P0x), [p0y] "v"(P0y),
                          [lb] "n"(kQLdsBase), [wp] "s"(wpitch)
                        : "memory", "vcc", "scc", "v48", "v49", "v52", "v53", "v54", "v55");
                    // the countdown only expires if the window is malformed (impossible): every ray of the pass to the fix-up list
                    expired = __builtin_amdgcn_readfirstlane((int)countdown) < 0;
                } else {
                    bool go;
                    int trips = 0;
                    rem = rem_start;
                    g = g0;
                    unsigned long long Tx = P0xs, Ty = P0ys;
                    uint32_t by = s0e;
                    do {
                        Tx += (unsigned long long)by * Xx;
                        Ty += (unsigned long long)by * Xy;
                        const uint32_t gm = (uint32_t)Tx < (uint32_t)Ty ? (uint32_t)Tx : (uint32_t)Ty;
                        g = g < gm ? g : gm;
                        if (GLOBAL || HYB) by = (uint32_t)(int)(int8_t)a.distg[(size_t)(uint32_t)((uint32_t)(Ty >> 32) * gpitch + (uint32_t)(Tx >> 32))];
                        else by = (uint32_t)(int)(int8_t)ldsb[((uint32_t)(Ty >> 32) & (kSwSide - 1)) * (uint32_t)kSwPitch + ((uint32_t)(Tx >> 32) & (kSwSide - 1))];
                        uint32_t nr;
                        const bool over = __builtin_usub_overflow((uint32_t)rem, by, &nr);
                        go = !over;
                        rem = (int)nr;
                        cnt_probe += (go && valid) ? 1 : 0;
                    } while (go && ++trips <= ((GLOBAL || HYB) ? (int)cdinit : 300));
                    if (go) g = 0u;
                }
                if (COUNT && valid) ++cnt_probe;
                const uint32_t thr = expired ? 0xFFFFFFFFu : gthresh;     // wave-uniform: a scalar select
                const bool amb = valid && g < thr;
                lt_pending = 0.0;
                if (valid && !amb) {
                    const int left = rem > 0 ? rem : 0;                 // samples left at the hit; 0 = no hit (step index P)
                    lt_pending = *reinterpret_cast<const double *>(reinterpret_cast<const char *>(a.Ltd) +
                                                                   mad_u24_s((uint32_t)(left + kSwUnder), st8, (uint32_t)(jcur + a.beam_margin) << 3));
                    if (want_steps) store_step(a, (int64_t)i, jcur, a.P - left);
                }
                if (amb) {
                    const unsigned int fslot = atomicAdd(&fixn_sh, 1u);
                    if (fslot < a.fix_cap) fix_seg[fslot] = ((unsigned long long)sl << 16) | (unsigned long long)jcur;
                }
            }
            acc += lt_pending;
        }
        // A lane of the asm walk with a sample inside the guard (one walk in several thousand: the guard is 2^-22 px wide), or
        // an expired countdown: the real beams of its walk go to the fix-up list and what it summed there is dropped
        if (live && walk_n > 0 && (expired_fast || gwalk < gthresh)) {
            const int jr0 = n1 > 0 ? ja : ja2, jr1 = n1 > 0 ? jb : a.B;      // the lane's real beams of this walk (virtual ones add 0 whatever they hit)
            for (int k = 0; k < walk_n; ++k) {
                const int jamb = jwalk + walk_t0 + k;
                if (jamb < jr0 || jamb >= jr1) continue;
                const unsigned int fslot = atomicAdd(&fixn_sh, 1u);
                if (fslot < a.fix_cap) fix_seg[fslot] = ((unsigned long long)sl << 16) | (unsigned long long)(jamb & 0xFFFF);
            }
            acc_fast = 0.0;
        }
        // the sum of this particle's rays in this wedge joins the slot's accumulator: one fp64 atomic per (slot, wedge), a wave's
        // sixty-four on 512 contiguous bytes; exact and order-independent (E4).  Round 2 kept one partial-sum array per wedge
        // group instead, read-modify-written by the owning lane and summed by k_combine_logw: 0.09 ms more per update.
        if (have) atomicAdd(&a.logw[slot], acc_fast + acc);
    }
    }   // wedges of the group
    }   // work items
    __syncthreads();
    if (a.dbg && threadIdx.x == 0) {
        unsigned long long *d = a.dbg + (size_t)blockIdx.x * 4;
        d[0] = dbg_t0; d[1] = __builtin_amdgcn_s_memrealtime(); d[2] = dbg_items | (dbg_bar << 32); d[3] = dbg_wait;
    }
    if (threadIdx.x == 0) a.fix_count[(size_t)blockIdx.x * 8] = fixn_sh;      // may exceed fix_cap: k_fix_overflow reports it
    if (COUNT && a.counters) {
        cnt_probe = wave_sum_u64(cnt_probe);
        if (lane == 0 && cnt_probe) atomicAdd(&a.counters[2], cnt_probe);
    }
}

// log-weight of every particle = its slot's accumulator (k_rays_sweep's per-wedge sums and what k_rays_far / k_rays_skip<FAR> /
// k_rays_fix / k_rays_exact added: fp64 atomics of earlier kernels, which plain loads see, profiles/r02_coherence.txt), read
// coalesced in sorted-slot order; one scattered 8-byte store per particle puts it where the rest of the update expects it.
// Also the per-workgroup maximum for the normalisation (k_final_max reduces `maxpart`).
// The first workgroup also raises the overflow flag of the fix-up lists (what k_fix_overflow does for the other ray kernels: *over =
// number of segments whose append count exceeded the capacity; the counts are final, every kernel that appends ran before this one).
__global__ __launch_bounds__(256) void k_combine_logw(int64_t n, const uint32_t *__restrict__ perm, const double *__restrict__ acc,
                                                     double *__restrict__ logw, double *__restrict__ maxpart,
                                                     const unsigned long long *__restrict__ fix_counts, int nseg, unsigned long long fix_cap,
                                                     unsigned long long *__restrict__ over)
{
    __shared__ double sm[4];
    __shared__ unsigned int over_sh;
    if (blockIdx.x == 0 && over) {
        if (threadIdx.x == 0) over_sh = 0u;
        __syncthreads();
        unsigned int c = 0;
        for (int k = threadIdx.x; k < nseg; k += blockDim.x) c += fix_counts[(size_t)k * 8] > fix_cap ? 1u : 0u;
        if (c) atomicAdd(&over_sh, c);
        __syncthreads();
        if (threadIdx.x == 0) *over = over_sh;
    }
    double m = -INFINITY;
    // four slots per trip: their (coalesced) loads are all requested before the first scattered store goes out
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; s + 3 * stride < n; s += 4 * stride) {
        const double v0 = acc[s], v1 = acc[s + stride], v2 = acc[s + 2 * stride], v3 = acc[s + 3 * stride];
        const uint32_t i0 = perm[s], i1 = perm[s + stride], i2 = perm[s + 2 * stride], i3 = perm[s + 3 * stride];
        logw[i0] = v0; logw[i1] = v1; logw[i2] = v2; logw[i3] = v3;
        m = fmax(fmax(m, v0), fmax(fmax(v1, v2), v3));
    }
    for (; s < n; s += stride) {
        const double v = acc[s];
        logw[perm[s]] = v;
        m = fmax(m, v);
    }
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) maxpart[blockIdx.x] = fmax(fmax(sm[0], sm[1]), fmax(sm[2], sm[3]));
}

}  // namespace mcl
