// mcl_engine.hip — C-ABI shim (include/mcl_hip_engine.h) over the HIP kernels in mcl_kernels.h.
//
// Host-side work done here, all of it init-time or O(beams) per update:
//   * sensor table (cpp:233-292) in double, its fp32 log form, the padded distance field;
//   * motion scalars (cpp:452-471), obs_idx (cpp:549-554,570,573);
//   * kernel launches on the engine's own stream + HIP-event stage timings (utils.hpp:51-57).
// There is NO CPU fallback: without a gfx950 device mcl_create fails with MCL_ERR_NO_DEVICE.
#include "../../include/mcl_hip_engine.h"

#include <cstring>
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <rocprim/device/device_radix_sort.hpp>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "mcl_kernels.h"

namespace {

thread_local std::string g_create_error;

enum { EV_START = 0, EV_RESAMPLE, EV_QUERY, EV_RAYS, EV_SENSOR, EV_K0, EV_K1, EV_COUNT };   // K0..K1 bracket the dominant kernel

}  // namespace

static_assert(MCL_WEDGES == mcl::kWedges || MCL_KWEDGES != 16, "include/mcl_hip_engine.h and csrc/mcl_wedge.h disagree");

constexpr unsigned long long kExactCap = 1ull << 16;   // level-3 rays per launch handled by k_rays_exact (more: inline)
// d_result / h_result: [0..7] scalars, [8..11] counters, [12] work-list overflow flag, [13] work counter, [14] level-3 list
// length, [15] far-list length (and the staging word of a global maximum), [16] length of the compact parent list
constexpr int kResultWords = 17;
constexpr int kResultStage = 40;             // h_result word that stages a host value on its way to the device
constexpr int kResultStamp = 32;             // h_result word a small update's last kernel stamps (the host polls it)

struct mcl_comm;
static void comm_free(struct mcl_comm *c);
static void comm_forget(struct mcl_comm *c);
struct mcl_engine {
    mcl_config_t cfg{};
    int num_cu = 256;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;      // the per-update observation tables are built here, beside the resampling and ordering kernels
    hipEvent_t ev_obs = nullptr;        // ... and the ray stage waits for this
    hipEvent_t ev[EV_COUNT]{};
    std::string err;

    // map
    bool have_map = false;
    int W = 0, H = 0, P = 0, Wp = 0, Hp = 0, Wps = 0, tw_cells = 0;
    double res = 0, ox = 0, oy = 0;
    std::vector<double> table;          // (P+1)^2 column-major (d*(P+1)+r)
    int8_t *d_grid = nullptr;
    uint8_t *d_dist = nullptr;
    uint8_t *d_dist4 = nullptr;         // nibble-packed copy of d_dist (k_rays_skip's LDS window is a straight copy of it)
    uint8_t *d_distq[4]{};              // directional skip fields, one per quadrant (k_rays_quad / k_rays_far)
    float *d_L = nullptr;               // [r_obs][d]
    double *d_table = nullptr;          // double table (product mode)

    // beams
    int B = 0, bpad = 0;
    std::vector<float> angles;
    float *d_angle = nullptr;
    double2 *d_beam_cs = nullptr;
    double2 *d_beam_csx = nullptr;      // k_rays_sweep's copy with virtual beams either side (set_beam_angles)
    double2 *d_beam_csxg = nullptr;     // the same for the global-field form: a virtual beam repeats the first / last REAL beam
    // k_rays_sweep<.., REC> (an evenly spaced scan): directions of the grid angles a0 + j inc, every beam's offset from its grid angle
    // (one entry per table column), cos / sin of the increment; rec_ok: the scan qualifies
    double2 *d_beam_csi = nullptr;
    double *d_beam_err = nullptr;
    double rec_c = 1.0, rec_s = 0.0;
    bool rec_ok = false;
    int beam_pad = 0, beam_margin = 0;
    int32_t *d_obs_idx = nullptr;
    float *d_obs = nullptr;
    float *h_obs = nullptr;             // pinned staging for the per-update scan
    uint32_t *d_free = nullptr;         // linear indices of free cells (data == 0), row-major, cpp:199-213
    uint64_t n_free = 0;
    uint32_t init_idx = 0;
    float *d_Lt = nullptr;
    size_t lt_capacity = 0;
    double *d_Ltd = nullptr;            // k_rays_sweep's fp64 table (mcl_rays_sweep.h), built per update when that kernel runs
    size_t ltd_capacity = 0;
    int ltd_cols = 0;
    bool ltd_ready = false;             // d_Ltd holds the table of the observation in d_obs_idx (cleared when a new scan is staged)
    bool sweep_layout_ok = false;       // k_rays_sweep's static LDS ends where its raw window offset (kQLdsBase) assumes
    int last_sweep_global = 0, last_sweep_rec = 0, last_sweep_pairs = 0;    // the form of k_rays_sweep the last ray stage ran
    int env_sweep_pairs = -1;           // MCL_SWEEP_PAIRS: -1 the engine decides, 0 / 1 forced (A/B measurements)
    bool sweep_rec_layout_ok = false;   // the same for the <.., REC> instantiations (LDS form: window + offset table; global form: the table)
    bool quad_layout_ok = false, cell_layout_ok = false;   // the same for k_rays_quad / k_rays_cell
    bool skip_layout_ok = false;        // k_rays_skip has no static LDS (its window is addressed from LDS offset 0)
    // k_rays_sweep on the wedge fields in GLOBAL memory (ranges a 256-cell LDS window cannot hold; MCL_SWEEP_GLOBAL=1 forces it)
    uint8_t *d_distg = nullptr;         // kWedges mirrored fields with a two-cell stop ring and a tail of stop rows each (k_ring_field)
    int distg_pitch = 0;                // row pitch
    size_t distg_stride = 0;            // bytes per field, tail included
    const char *distg_why_not = nullptr;     // why the global-field form of k_rays_sweep is not available for this map, if it is not
    bool sweep_global = false;          // this map takes the global-field variant (decided at mcl_set_map)
    bool env_sweep_global = false;
    int env_sw_split16 = -1;            // MCL_SW_SPLIT16=0/1 forces RayArgs::split16 (default: by size)
    bool env_no_obs_overlap = false;    // MCL_NO_OBS_OVERLAP: the observation tables are built on the main stream, after the resampling kernel
    bool env_no_prep_fold = false;      // MCL_NO_PREP_FOLD: k_prep_small stays a launch of its own
    // the few words the ray stage wants cleared (k_prep_small's job): what the last windowed launch passed, so that the NEXT
    // update's resampling kernel can do it (prep_folded: it did, with exactly prep_cache)
    mcl::PrepClear prep_cache{};
    bool prep_cache_valid = false, prep_folded = false;
    int64_t prep_cache_n = 0;
    bool max_partials_ready = false;    // k_combine_logw left the per-workgroup maxima of d_logw in d_part
    int4 *d_items = nullptr;            // k_rays_sweep's work items (guided schedule), planned on the device every update
    int4 *d_centres = nullptr;          // per run of units: window centre, first unit, units (k_sweep_plan)
    size_t items_capacity = 0;
    int *d_nitems = nullptr;            // number of work items (written by k_sweep_plan)
    int64_t plan_n = 0;
    double4 *d_unit_sums = nullptr;     // per unit of the sorted order: (sum px, sum py, count, -), bounding box
    uint32_t *d_unit_begin = nullptr;   // first slot of every unit + one (k_unit_table)
    int *d_nunits = nullptr;            // number of units of this update's sorted order
    size_t unit_sums_capacity = 0;
    // environment knobs, read once at mcl_create (0 / negative = default)
    int64_t env_cell_min = 0, env_cell_slice = 0;
    int env_qslices_per_cu = 0, env_qside = 0, env_sweep_g = 0;
    std::string env_debug_wg;

    // particles
    int64_t cap = 0, N = 0;
    bool have_particles = false;
    double *d_x[2]{}, *d_y[2]{}, *d_th[2]{};
    int cur = 0;
    double *d_w = nullptr, *d_logw = nullptr, *d_tmp = nullptr;   // tmp: cap*3 doubles
    double *d_carry[2]{};               // logw - max of the last update (adaptive resampling: what a kept particle carries)
    int carry_idx = 0;                  // d_carry[carry_idx] is current; k_weights writes the other one
    bool carry_valid = false, carry_pending = false;
    bool resampled_last = true;
    // hipGraph of the update's tail (observation upload ... result read-back) for the k_rays_skip path, one per particle buffer
    hipGraphExec_t graph_exec[2] = {nullptr, nullptr};
    bool graph_warm = false;            // a regular update has run since the sizes / map / beams last changed
    bool capturing = false;
    double *d_logw_acc = nullptr;       // k_rays_quad/far/fix accumulate here with atomics; k_gather_logw copies to d_logw
    uint64_t *d_q = nullptr, *d_cdf = nullptr, *d_blocktot = nullptr;
    uint32_t *d_bm = nullptr;           // mcl_stage_distinct_parents: bitmap over the global particle indices, its popcounts and their prefix
    uint64_t *d_bm_pop = nullptr, *d_bm_pref = nullptr;
    size_t bm_capacity = 0;
    // compact list of the particles with a non-zero fixed-point weight, written by the scan of d_q (mcl::CompactOut)
    uint32_t *d_blockcnt = nullptr;     // per scan tile (blocktot_capacity entries)
    uint64_t *d_ccdf = nullptr, *d_ctop = nullptr;
    uint32_t *d_cidx = nullptr;
    double4 *d_crec = nullptr;
    int64_t compact_cap = 0;            // room in the list (cap / 4, at least 4096)
    int64_t compact_n = -1;             // entries of the list that describes d_cdf / the current particles; -1: none
    bool compact_pending = false;       // the last scan wrote a list; its length arrives with the next result read-back
    bool compact_used = false;          // the last resampling drew from a compact list
    uint64_t *d_gcdf = nullptr, *d_gtop = nullptr;   // merged CDF of the shards' gathered lists (mcl_stage_resample_compact)
    size_t gcdf_capacity = 0;
    int env_no_compact = 0;
    uint64_t *d_leaders = nullptr;      // last CDF entry of every 16-entry group of the array d_blocktot describes
    size_t leaders_capacity = 0;
    double4 *d_pack[2]{};               // (x, y, theta, -) records of buffer 0/1, written by k_resample_motion
    bool pack_valid[2] = {false, false};
    int32_t *d_idx = nullptr;
    uint8_t *d_steps = nullptr;
    size_t steps_capacity = 0, blocktot_capacity = 0;
    const uint64_t *blocktot_for = nullptr;   // which CDF array d_blocktot currently describes
    int64_t blocktot_n = 0;
    double *d_part = nullptr;           // kRedBlocks * 8: per-workgroup partial sums (k_weights)
    double *d_maxpart = nullptr;        // kRedBlocks: per-workgroup maxima of d_logw (k_combine_logw / k_reduce_max)
    bool sums_pending = false;          // k_weights left partial sums that the next scan's spine turns into scalars[1..7]
    double *d_scalars = nullptr;        // 8
    unsigned long long *d_counters = nullptr;  // 4
    double *d_inject = nullptr;         // cap*4 (normals + uniforms)
    double4 *d_pc = nullptr;            // cap: per-particle constants for k_rays_skip / k_rays_quad
    short4 *d_qr = nullptr;             // cap: per-particle quadrant ranges (k_rays_quad)
    bool quad_ok = false;               // beam angles monotone over less than a full turn
    int qside = 0;                      // k_rays_quad window side (0: not usable for this map)
    unsigned long long *d_fix_list = nullptr, *d_fix_count = nullptr, *d_fix_over = nullptr;
    unsigned long long *d_exact_list = nullptr;   // level-3 rays for k_rays_exact; its counter is word 14 of d_result
    unsigned long long fix_cap = 0, fix_alloc = 0;
    size_t fix_count_alloc = 0;
    int fix_segments = 0;
    uint8_t *d_far = nullptr;           // cap * 4 flags
    uint32_t *d_far_list = nullptr;     // k_rays_sweep: slots with a flagged quadrant (cap entries, allocated on first use)
    uint32_t *d_far_sorted = nullptr, *d_far_cnt = nullptr;   // the same in ascending order (k_far_*), per-2048-slot counts
    // cell sort for k_rays_cell
    double4 *d_pcs = nullptr;           // cap: pc in sorted order
    double *d_ths = nullptr;            // cap: heading in sorted order
    uint8_t *d_distw = nullptr;         // kWedges wedge fields for k_rays_cell, each Hp x Wps bytes
    uint32_t *d_perm = nullptr, *d_skey = nullptr, *d_srank = nullptr;   // cap each
    uint32_t *d_skey2 = nullptr, *d_sval2 = nullptr;   // MCL_SORT=radix: sorted keys / indices
    void *d_sort_tmp = nullptr;
    size_t sort_tmp_bytes = 0;
    int env_sort_radix = -1;           // MCL_SORT=radix / hist forces one ordering path; default: by size
    uint32_t *d_tile_used = nullptr;    // one mark per kHistTile buckets of the sort histogram: touched by this update's sort
    uint32_t *d_hist = nullptr, *d_histpart = nullptr;                   // kSortBuckets, kSortBuckets / kHistTile
    int *d_bbox = nullptr;              // 7: bounding box, occupied tiles, numbering in use, window play
    uint32_t *d_cut_start = nullptr, *d_cut_end = nullptr;   // kSwMaxCuts each: where the buckets of a sparse set start / end in the radix-sorted order (zero between sorts)
    bool env_no_bucket_cuts = false;    // MCL_NO_BUCKET_CUTS: units on the plain grid of 1024 slots, sparse sets ordered by whole tiles (rounds 2-3 before the cuts)
    int *d_tilemap = nullptr, *d_tilemark = nullptr;   // kSortMaxTiles each: tile of the map -> compact id; marks of the occupied tiles (zero between sorts)
    // The ordering layout (bounding box, occupied tiles) of an update's children is made on the second stream right after the
    // resampling kernel and used by the NEXT update, whose resampling kernel then writes the sort keys itself: d_bbox / d_tilemap are
    // the layout in use, *_nx the one being made; swapped at the end of an update.  layout_valid: d_bbox describes the previous
    // update's children of this configuration (cleared by graph_reset: map, beams, particles set from outside).
    int *d_bbox_nx = nullptr, *d_tilemap_nx = nullptr, *d_tilemark_nx = nullptr;
    hipEvent_t ev_children = nullptr, ev_layout = nullptr;
    hipEvent_t ev_ext_in = nullptr, ev_ext_out = nullptr;   // ordering against a caller's stream (mcl_stream_wait_external / mcl_external_wait_stream)
    bool stage_async_rays = false, stage_async_weights = false;
    bool stage_kept = false;            // the staged flow's last children are the previous particles themselves (mcl_stage_keep)
    struct mcl_comm *comm = nullptr;    // RCCL communicator of a sharded set (mcl_comm_create), or null
    unsigned long long list_epoch = 0;  // counts the rewrites of the compact list (a gathered copy of an older one is stale)
    bool layout_valid = false, layout_pending = false;
    int64_t layout_n = 0;
    // Stage events BOUND TO DISPATCHES: the stop event of hipExtLaunchKernelGGL costs nothing, a hipEventRecord between two kernels
    // of a stream ~3 us of pipeline (tools/ubench/event_cost.hip; elapsed times across different launches are valid).  Set by the
    // launch that bound the event, cleared by the code that would otherwise record it.
    bool ev_resample_bound = false, ev_rays_bound = false, ev_sensor_bound = false, ev_query_skipped = false;
    bool bind_sensor_event = false;     // the next scan of the engine's own weights binds EV_SENSOR to its last kernel
    bool layout_wanted = false;         // the resampling kernel left children to make the next layout of (layout_mark -> next_layout_launch)
    bool layout_stale_used = false;     // this update orders by the previous update's layout (do_update -> launch_rays)
    bool keys_done = false;             // ... and its resampling kernel wrote the (key, index) pairs
    bool env_no_stale_layout = false;   // MCL_NO_STALE_LAYOUT: every update makes its own layout first (rounds 1-3)
    bool env_comm_no_lists = false;     // MCL_COMM_NO_LISTS: mcl_comm_update takes the dense exchange on every update
    bool env_comm_no_pregather = false; // MCL_COMM_NO_PREGATHER: mcl_comm_update gathers the lists when it starts, not when the previous one ends
    mcl::PrepClear prep_passed{};       // what the resampling kernel was given to clear (prep_folded)
    double2 *d_slice_mean = nullptr;    // one per slice of the sorted order
    size_t slice_mean_capacity = 0;
    bool last_quad = false;             // the last ray stage ran k_rays_quad (overflow check pending)
    int last_mode = 0;                  // 1 march, 2 skip, 3 quad, 4 cell, 5 sweep
    unsigned long long result_seq = 0;  // stamps the result block a small update writes to pinned memory (h_result[kResultStamp])
    int env_tiny_poll = 1;
    bool far_fresh = true;              // no ray stage has seen the current particle set yet (set / initialised since the last one)
    bool pc_ready = false;              // d_pc already holds the constants of the current particles (written by k_resample_motion)
    int reserved_cus = 0;               // CUs k_rays_quad's persistent grid leaves free (for RCCL kernels running beside it)
    unsigned long long *d_result = nullptr;   // [0..7] scalars, [8..11] counters, [12..13] overflow flag + work counter: one D2H copy
    unsigned long long *h_result = nullptr;   // pinned mirror of d_result
    double h_scalars[8]{};
    uint64_t q_total = 0;
    double global_sums[5]{};            // sum w, wx, wy, wsin, wcos actually used for outputs
    bool have_idx = false, have_steps = false, have_logw = false;
    uint32_t update_idx = 0;
    double timings[6]{};
    double ray_ms = 0;
    bool ray_ms_is_graph_tail = false;  // ray_ms is the whole captured tail of a small update, not one kernel
    unsigned long long h_counters[4]{};
    unsigned long long h_fix_count = 0;
};

namespace {

#define HIPCHK(h, call)                                                                          \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess) {                                                                  \
            (h)->err = std::string(#call) + ": " + hipGetErrorString(e_);                        \
            return MCL_ERR_HIP;                                                                  \
        }                                                                                        \
    } while (0)

int fail(mcl_engine *h, int code, const char *msg)
{
    if (h) h->err = msg;
    return code;
}
int fail(mcl_engine *h, int code, const std::string &msg) { return fail(h, code, msg.c_str()); }

template <class T>
void dfree(T *&p)
{
    if (p) { (void)hipFree(p); p = nullptr; }
}

// cpp:233-292, restated; column-major (d*(tw)+r).
void build_sensor_table(const mcl_config_t &c, int P, std::vector<double> &t)
{
    const int tw = P + 1;
    t.assign((size_t)tw * tw, 0.0);
    for (int d = 0; d < tw; ++d) {
        double norm = 0.0;
        for (int r = 0; r < tw; ++r) {
            double prob = 0.0;
            double z = (double)(r - d);
            prob += c.z_hit * std::exp(-(z * z) / (2.0 * c.sigma_hit * c.sigma_hit)) / (c.sigma_hit * std::sqrt(2.0 * M_PI));
            if (r < d) prob += 2.0 * c.z_short * (d - r) / (double)d;
            if (r == P) prob += c.z_max;
            if (r < P) prob += c.z_rand * 1.0 / (double)P;
            norm += prob;
            t[(size_t)d * tw + r] = prob;
        }
        if (norm > 0)
            for (int r = 0; r < tw; ++r) t[(size_t)d * tw + r] /= norm;
    }
}

// Padded stop grid + skip-distance field.
// Padded cell (xp,yp), xp in [0,W], yp in [0,H], stands for reference cell (max(xp-1,0), max(yp-1,0)):
// the reference truncates toward zero (cpp:628-629), so pixel coordinates in (-1,0) read cell 0.
// Everything outside the padded grid is "stop" (map boundary, cpp:632-636).
//
// skip(c) = how far the fixed-step march may jump from a sample inside cell c without being able to
// land in a stop cell earlier.  Samples are exactly one pixel apart along the ray, so sample k+j lies at
// Euclidean distance j from sample k; it can be inside stop cell t only if j >= dist(p_k, t) >= gap(c, t),
// the distance between the two (closed) cell squares, with equality only for p_k on the boundary of c
// (such samples are caught by the kernel's boundary guard).  Hence skip(c) = floor(min_t gap(c,t)) + 1.
// gap^2(c,t) = max(|dx|-1,0)^2 + max(|dy|-1,0)^2 is the squared centre distance from c to the 3x3
// dilation of t, so one exact integer squared-EDT (Felzenszwalb & Huttenlocher lower envelopes) of the
// dilated stop set gives it.  Stop cells get 0; values are capped at 255.
void edt_1d(const int64_t *f, int n, int64_t *d, int *v, double *z)
{
    const int64_t INF = (int64_t)1 << 40;
    int k = 0;
    v[0] = 0; z[0] = -1e30; z[1] = 1e30;
    for (int q = 1; q < n; ++q) {
        if (f[q] >= INF) continue;
        while (true) {
            if (f[v[k]] >= INF) { v[k] = q; z[k] = -1e30; z[k + 1] = 1e30; break; }
            double s = ((double)(f[q] + (int64_t)q * q) - (double)(f[v[k]] + (int64_t)v[k] * v[k])) / (2.0 * q - 2.0 * v[k]);
            if (s <= z[k]) { --k; if (k < 0) { k = 0; v[0] = q; z[0] = -1e30; z[1] = 1e30; break; } continue; }
            ++k; v[k] = q; z[k] = s; z[k + 1] = 1e30;
            break;
        }
    }
    k = 0;
    for (int q = 0; q < n; ++q) {
        while (z[k + 1] < q) ++k;
        int64_t dq = (int64_t)(q - v[k]);
        d[q] = (f[v[k]] >= INF) ? INF : dq * dq + f[v[k]];
    }
}

void build_distance_field(const int8_t *grid, int W, int H, int Wp, int Hp, int Wps, std::vector<uint8_t> &dist)
{
    // work grid = padded grid plus a one-cell stop border on every side
    const int Ww = Wp + 2, Hw = Hp + 2;
    std::vector<uint8_t> stop((size_t)Hw * Ww, 1), dil((size_t)Hw * Ww, 0);
    for (int yp = 0; yp < Hp; ++yp)
        for (int xp = 0; xp < Wp; ++xp) {
            int gx = std::max(xp - 1, 0), gy = std::max(yp - 1, 0);
            stop[(size_t)(yp + 1) * Ww + xp + 1] = grid[(size_t)gy * W + gx] > 50;
        }
    for (int y = 0; y < Hw; ++y)
        for (int x = 0; x < Ww; ++x) {
            if (!stop[(size_t)y * Ww + x]) continue;
            for (int dy = -1; dy <= 1; ++dy)
                for (int dx = -1; dx <= 1; ++dx) {
                    int yy = y + dy, xx = x + dx;
                    if (yy >= 0 && yy < Hw && xx >= 0 && xx < Ww) dil[(size_t)yy * Ww + xx] = 1;
                }
        }
    const int64_t INF = (int64_t)1 << 40;
    std::vector<int64_t> g((size_t)Hw * Ww);
    const int nmax = std::max(Ww, Hw);
    std::vector<int64_t> f(nmax), d(nmax);
    std::vector<int> v(nmax + 1);
    std::vector<double> z(nmax + 2);
    for (int x = 0; x < Ww; ++x) {            // columns
        for (int y = 0; y < Hw; ++y) f[y] = dil[(size_t)y * Ww + x] ? 0 : INF;
        edt_1d(f.data(), Hw, d.data(), v.data(), z.data());
        for (int y = 0; y < Hw; ++y) g[(size_t)y * Ww + x] = d[y];
    }
    dist.assign((size_t)Hp * Wps, 0);
    for (int y = 1; y <= Hp; ++y) {           // rows
        for (int x = 0; x < Ww; ++x) f[x] = g[(size_t)y * Ww + x];
        edt_1d(f.data(), Ww, d.data(), v.data(), z.data());
        for (int x = 1; x <= Wp; ++x) {
            int val = 0;
            if (!stop[(size_t)y * Ww + x]) {
                int64_t g2 = d[x];
                int64_t r = (int64_t)std::sqrt((double)g2);
                while (r * r > g2) --r;
                while ((r + 1) * (r + 1) <= g2) ++r;
                val = (int)std::min<int64_t>(r + 1, 255);
            }
            dist[(size_t)(y - 1) * Wps + (x - 1)] = (uint8_t)val;
        }
    }
}

// Directional skip field for k_rays_quad, quadrant q = (sx, sy): a ray whose direction has sign sx in x and sy
// in y can only ever enter cells t with sx*(t_x - c_x) >= 0 and sy*(t_y - c_y) >= 0, so only those stop cells
// bound the jump: skip_q(c) = floor(min over forward stop cells t of gap(c, t)) + 1, gap as in
// build_distance_field.  Walls beside or behind a ray no longer shorten its jumps (-30 % probes on the
// benchmark input).  Exact integer arithmetic: per row the forward x-gap h to the next stop, then per column
// a one-sided squared distance transform (lower envelope of parabolas, sources only ahead of the query).
void build_directional_field(const int8_t *grid, int W, int H, int Wp, int Hp, int Wps, int sx, int sy, std::vector<uint8_t> &dist)
{
    const int64_t INF = (int64_t)1 << 40;
    // stop(xf, yf) in "forward" coordinates: xf = sx > 0 ? xp : Wp-1-xp, same for y
    auto stop_at = [&](int xf, int yf) -> bool {
        int xp = sx > 0 ? xf : Wp - 1 - xf, yp = sy > 0 ? yf : Hp - 1 - yf;
        int gx = std::max(xp - 1, 0), gy = std::max(yp - 1, 0);
        return grid[(size_t)gy * W + gx] > 50;
    };
    // h[yf][xf]: gap in x to the nearest stop at x' >= xf in the same row (the cell just outside the grid is a stop)
    std::vector<int32_t> h((size_t)(Hp + 1) * Wp);
    for (int yf = 0; yf < Hp; ++yf) {
        int nxt = Wp;
        for (int xf = Wp - 1; xf >= 0; --xf) {
            if (stop_at(xf, yf)) nxt = xf;
            h[(size_t)yf * Wp + xf] = std::max(nxt - xf - 1, 0);
        }
    }
    for (int xf = 0; xf < Wp; ++xf) h[(size_t)Hp * Wp + xf] = 0;      // the row beyond the grid is all stop
    dist.assign((size_t)Hp * Wps, 0);
    std::vector<int> vp(Hp + 2);          // envelope: source positions (in r = decreasing-y order)
    std::vector<double> z(Hp + 3);
    std::vector<int64_t> hg(Hp + 2);      // heights of the sources
    for (int xf = 0; xf < Wp; ++xf) {
        // g2(yf) = min( h(yf)^2 , min over p >= yf of (p - yf)^2 + h(p+1)^2 ): sources p = Hp-1 .. 0 arrive in
        // decreasing p, i.e. increasing r = Hp-1-p; the query sits at the newest source's position.
        int k = -1;
        for (int yf = Hp - 1; yf >= 0; --yf) {
            const int r = Hp - 1 - yf;
            const int64_t hv = h[(size_t)(yf + 1) * Wp + xf];
            const int64_t fh = hv * hv;
            // insert parabola (r, fh)
            while (true) {
                if (k < 0) { k = 0; vp[0] = r; hg[0] = fh; z[0] = -1e30; z[1] = 1e30; break; }
                double sI = ((double)(fh + (int64_t)r * r) - (double)(hg[k] + (int64_t)vp[k] * vp[k])) / (2.0 * r - 2.0 * vp[k]);
                if (sI <= z[k]) { --k; continue; }
                ++k; vp[k] = r; hg[k] = fh; z[k] = sI; z[k + 1] = 1e30;
                break;
            }
            // query at r: the parabola whose interval contains r
            int kk = k;
            while (z[kk] > (double)r) --kk;
            int64_t dq = (int64_t)(r - vp[kk]);
            int64_t g2 = dq * dq + hg[kk];
            // exactness of the envelope near interval ends: also try the neighbours
            if (kk > 0) { int64_t d2 = (int64_t)(r - vp[kk - 1]); g2 = std::min(g2, d2 * d2 + hg[kk - 1]); }
            if (kk < k) { int64_t d2 = (int64_t)(r - vp[kk + 1]); g2 = std::min(g2, d2 * d2 + hg[kk + 1]); }
            const int64_t hs = h[(size_t)yf * Wp + xf];
            g2 = std::min(g2, hs * hs);
            int val = 0;
            if (!stop_at(xf, yf)) {
                int64_t rt = (int64_t)std::sqrt((double)g2);
                while (rt * rt > g2) --rt;
                while ((rt + 1) * (rt + 1) <= g2) ++rt;
                val = (int)std::min<int64_t>(rt + 1, 255);
            }
            int xp = sx > 0 ? xf : Wp - 1 - xf, yp = sy > 0 ? yf : Hp - 1 - yf;
            dist[(size_t)yp * Wps + xp] = (uint8_t)val;
        }
    }
    (void)INF; (void)H;
}

// cpp:452-471
void motion_scalars(const double action[3], double &dt, double &v, double &w)
{
    dt = 0.01; v = 0.0; w = 0.0;
    double fd = action[0], ad = action[2];
    if (std::abs(fd) > 0.001) {
        if (std::abs(fd) < 0.1) dt = std::abs(fd) / 1.0;
        else dt = std::abs(fd) / 5.0;
        dt = std::max(0.001, std::min(dt, 0.1));
        v = fd / dt;
    }
    if (std::abs(ad) > 0.001) w = ad / dt;
}

int ensure_lt(mcl_engine *h)
{
    size_t need = (size_t)(h->P + 1) * h->bpad;
    if (need > h->lt_capacity) {
        dfree(h->d_Lt);
        HIPCHK(h, hipMalloc(&h->d_Lt, 2 * need * sizeof(float)));      // [Lt | Lt with the rows reversed (k_rays_cell)]
        h->lt_capacity = need;
    }
    h->ltd_cols = (h->B + 2 * h->beam_margin + 64) & ~63;              // beam j in column j + beam_margin; at least one all-zero column after the last beam
    const size_t need_d = (size_t)mcl::sweep_table_rows(h->P) * h->ltd_cols;
    if (need_d > h->ltd_capacity) {
        dfree(h->d_Ltd);
        HIPCHK(h, hipMalloc(&h->d_Ltd, need_d * sizeof(double)));
        h->ltd_capacity = need_d;
        h->ltd_ready = false;
    }
    return MCL_OK;
}

void graph_reset(mcl_engine *h);
void build_ltd(mcl_engine *h);

int scan_weights(mcl_engine *h, const uint64_t *d_q, uint64_t *d_cdf, int64_t n, uint64_t offset, uint64_t *d_total)
{
    int nb = (int)((n + mcl::kScanTile - 1) / mcl::kScanTile);
    // the scan of the engine's own weights also leaves the compact list of the particles that carry weight
    mcl::CompactOut co{};
    const bool own = d_q == h->d_q && d_cdf == h->d_cdf && offset == 0 && !h->env_no_compact && h->d_ccdf;
    if (own) {
        co.block_cnt = h->d_blockcnt; co.ccdf = h->d_ccdf; co.cidx = h->d_cidx; co.crec = h->d_crec; co.ctop = h->d_ctop;
        co.x = h->d_x[h->cur]; co.y = h->d_y[h->cur]; co.th = h->d_th[h->cur];
        co.cap = (uint32_t)h->compact_cap; co.total = h->d_result + 16;
    }
    hipLaunchKernelGGL(mcl::k_scan_partials, dim3(nb), dim3(mcl::kScanThreads), 0, h->stream, d_q, n, h->d_blocktot, co.block_cnt);
    // (the spine is one workgroup that runs right after k_weights: it also finishes that kernel's partial sums, when asked)
    const bool fold = h->sums_pending && d_q == h->d_q;
    hipLaunchKernelGGL(mcl::k_scan_spine, dim3(1), dim3(1024), 0, h->stream, h->d_blocktot, nb, offset, d_total, co.block_cnt, co.total,
                       fold ? h->d_part : (const double *)nullptr, mcl::kRedBlocks, h->d_scalars);
    if (fold) h->sums_pending = false;
    const size_t nlead = (size_t)((n + 15) >> mcl::kLeaderShift) + 1;
    if (nlead > h->leaders_capacity) {
        graph_reset(h);                    // a captured update graph holds the old pointer
        dfree(h->d_leaders);
        HIPCHK(h, hipMalloc(&h->d_leaders, nlead * 8));
        h->leaders_capacity = nlead;
    }
    if (h->bind_sensor_event && own && !h->capturing) {          // the update's last kernel: EV_SENSOR is its stop event
        hipExtLaunchKernelGGL(mcl::k_scan_final, dim3(nb), dim3(mcl::kScanThreads), 0, h->stream, nullptr, h->ev[EV_SENSOR], 0, d_q, n, h->d_blocktot, d_cdf,
                              h->d_leaders, co);
        h->ev_sensor_bound = true; h->bind_sensor_event = false;
    } else {
        hipLaunchKernelGGL(mcl::k_scan_final, dim3(nb), dim3(mcl::kScanThreads), 0, h->stream, d_q, n, h->d_blocktot, d_cdf, h->d_leaders, co);
    }
    HIPCHK(h, hipGetLastError());
    h->blocktot_for = d_cdf; h->blocktot_n = n;
    if (d_cdf == h->d_cdf) { h->compact_n = -1; h->compact_pending = own; h->list_epoch++; }
    return MCL_OK;
}

// weights/q/sums from either log-weights (from_log) or raw weights already in d_w.  defer_sums: the caller scans d_q next
// (scan_weights), whose one-workgroup spine then also reduces the partial sums -- one launch less.
int weight_stats(mcl_engine *h, bool from_log, const double *d_max_override, bool defer_sums = false)
{
    const int64_t n = h->N;
    const double *src = from_log ? h->d_logw : h->d_w;
    const double *max_parts = nullptr;
    if (!d_max_override) {
        if (!(from_log && h->max_partials_ready))        // k_combine_logw already left the per-workgroup maxima in d_maxpart
            hipLaunchKernelGGL(mcl::k_reduce_max, dim3(mcl::kRedBlocks), dim3(mcl::kRedThreads), 0, h->stream, src, n, h->d_maxpart);
        max_parts = h->d_maxpart;                        // every workgroup of k_weights reduces them itself (no k_final_max in between)
    }
    h->max_partials_ready = false;
    hipLaunchKernelGGL(mcl::k_weights, dim3(mcl::kRedBlocks), dim3(mcl::kRedThreads), 0, h->stream, src, from_log ? 1 : 0,
                       d_max_override ? d_max_override : h->d_scalars, h->d_x[h->cur], h->d_y[h->cur], h->d_th[h->cur], n, h->d_w, h->d_q, h->d_part,
                       (from_log && h->cfg.resample_neff_permille > 0) ? h->d_carry[h->carry_idx ^ 1] : (double *)nullptr,
                       max_parts, mcl::kRedBlocks);
    h->carry_pending = from_log && h->cfg.resample_neff_permille > 0;     // the caller commits it (commit_carry)
    if (!from_log) h->carry_valid = false;
    if (defer_sums) h->sums_pending = true;
    else hipLaunchKernelGGL(mcl::k_final_sums, dim3(1), dim3(mcl::kRedThreads), 0, h->stream, h->d_part, mcl::kRedBlocks, h->d_scalars);
    HIPCHK(h, hipGetLastError());
    return MCL_OK;
}

void unpack_result(mcl_engine *h);

int fetch_scalars(mcl_engine *h)
{
    // scalars, counters and the work-list overflow flag in one copy into pinned memory
    HIPCHK(h, hipMemcpyAsync(h->h_result, h->d_result, kResultWords * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    unpack_result(h);
    return MCL_OK;
}

void unpack_result(mcl_engine *h)
{
    std::memcpy(h->h_scalars, h->h_result, 8 * sizeof(double));
    std::memcpy(h->h_counters, h->h_result + 8, 4 * sizeof(unsigned long long));
    h->h_fix_count = h->h_result[12];
    uint64_t qt;
    std::memcpy(&qt, &h->h_scalars[2], 8);
    h->q_total = qt;
    h->global_sums[0] = h->h_scalars[1];
    h->global_sums[1] = h->h_scalars[3];
    h->global_sums[2] = h->h_scalars[4];
    h->global_sums[3] = h->h_scalars[5];
    h->global_sums[4] = h->h_scalars[6];
    // length of the compact list the scan of this update's weights wrote (word 16); unusable when it outgrew its arrays
    if (h->compact_pending) {
        const unsigned long long na = h->h_result[16];
        h->compact_n = na <= (unsigned long long)h->compact_cap ? (int64_t)na : -1;       // 0: a valid, empty list (no weight here)
        h->compact_pending = false;
    }
}

// Work items of k_rays_sweep: made on the device from this update's unit statistics (k_sweep_plan, mcl_rays_sweep.h);
// the host only sizes the list (every unit on its own, once per wedge group, is the longest it can get).
// upper bound on the units of n sorted particles: the plain grid plus one cut per bucket of a sparse set (mcl::k_unit_table)
// cells the particles of one k_rays_sweep work item may spread over (per axis): what the 256-cell LDS window leaves beside a
// ray's reach, or -- on the global wedge fields -- the span of the cell field minus the reach on both sides
int sweep_play(const mcl_engine *h)
{
    if (h->sweep_global) return 1 << 20;             // every lane has its own origin there: no window a run could outgrow
    return mcl::kSwSide - (h->P + 2) - 3;
}

int64_t max_sweep_units(int64_t n) { return (n + mcl::kSwUnit - 1) / mcl::kSwUnit + mcl::kSwMaxCuts + 2; }

int launch_sweep_plan(mcl_engine *h, int64_t n, int nwg, int g)
{
    const int ngroups = mcl::kWedges / g;
    const size_t need = (size_t)max_sweep_units(n) * ngroups;
    if (need > h->items_capacity) {
        dfree(h->d_items); dfree(h->d_centres);
        h->items_capacity = 0;
        HIPCHK(h, hipMalloc(&h->d_items, need * sizeof(int4)));
        HIPCHK(h, hipMalloc(&h->d_centres, (size_t)max_sweep_units(n) * sizeof(int4)));
        h->items_capacity = need;
    }
    if (!h->d_nitems) HIPCHK(h, hipMalloc(&h->d_nitems, sizeof(int)));
    const int play = sweep_play(h);                                     // cells a window leaves for the particles of an item
    // (EV_K0 = the stop event of the kernel before the ray kernel, EV_K1 = the ray kernel's own: its duration, dispatch included, at no cost)
    hipExtLaunchKernelGGL(mcl::k_sweep_plan, dim3(1), dim3(1024), mcl::kPlanLds, h->stream, nullptr, h->ev[EV_K0], 0, h->d_unit_sums, h->d_nunits, ngroups, nwg,
                          (double)(play / 2 - 1), h->d_items, h->d_centres, h->d_nitems);
    HIPCHK(h, hipGetLastError());
    return MCL_OK;
}

// Which ray kernel a launch over n particles takes: 1 march, 2 skip, 3 quad, 4 cell, 5 sweep; 0 = the configured kernel
// cannot run with this map / beam set.  A pure function of the configuration, the map, the beam set and n, so that
// callers (graph eligibility, table building, mcl_get_planned_ray_kernel) can ask before anything is launched.  *why, when
// given, receives a static sentence saying what decided (the kernels that address their LDS window from a raw offset are
// only chosen when the host-side layout check of mcl_create passed: no configuration can make the device abort).
int choose_ray_mode(const mcl_engine *h, int64_t n, bool force_skip, const char **why = nullptr)
{
    const char *dummy;
    const char *&w = why ? *why : dummy;
    const int rk = h->cfg.ray_kernel;
    if (rk == MCL_RAYS_MARCH) { w = "configured: MCL_RAYS_MARCH"; return 1; }
    // k_rays_skip without its LDS layout (never seen): the literal march, same results
    if (rk == MCL_RAYS_SKIP || force_skip) {
        if (!h->skip_layout_ok) { w = "k_rays_skip's LDS layout check failed at mcl_create: literal march"; return 1; }
        w = force_skip ? "fix-up list overflow: the stage is re-run with k_rays_skip" : "configured: MCL_RAYS_SKIP";
        return 2;
    }
    const char *no_windows = nullptr;                        // why the windowed kernels (quad / cell / sweep) are out, if they are
    if (!h->quad_ok) no_windows = "beam angles are not monotone over less than a turn (or more than 16383 beams): the windowed kernels need contiguous beam ranges per direction wedge";
    else if (h->qside <= 0) no_windows = "MAX_RANGE_PX leaves less than 32 cells of play in a 280-cell byte window: the windowed kernels cannot hold a ray";
    const bool windows_ok = no_windows == nullptr;           // monotone beams over less than a turn, room for a byte window
    if (rk == MCL_RAYS_QUAD) { w = windows_ok && h->quad_layout_ok ? "configured: MCL_RAYS_QUAD" : (no_windows ? no_windows : "k_rays_quad's LDS layout check failed"); return windows_ok && h->quad_layout_ok ? 3 : 0; }
    if (rk == MCL_RAYS_CELL) { w = windows_ok && h->cell_layout_ok ? "configured: MCL_RAYS_CELL" : (no_windows ? no_windows : "k_rays_cell's LDS layout check failed"); return windows_ok && h->cell_layout_ok ? 4 : 0; }
    // its windows are 256 cells wide (mcl_rays_sweep.h) and addressed from a raw LDS offset checked at mcl_create
    // ... or, for ranges beyond that, probes the same wedge fields in global memory
    const char *no_sweep = h->quad_ok ? nullptr : no_windows;
    if (!no_sweep) {
        if (h->sweep_global) { if (!h->d_distg) no_sweep = h->distg_why_not ? h->distg_why_not : "k_rays_sweep's global-field form: fields not built"; }
        else if (!mcl::sweep_window_fits(h->P)) no_sweep = "MAX_RANGE_PX > 243: a 256-cell window of k_rays_sweep cannot hold a ray plus 8 cells of play";
        else if (!h->sweep_layout_ok) no_sweep = "k_rays_sweep's LDS layout check failed at mcl_create";
    }
    const bool sweep_ok = no_sweep == nullptr;
    if (rk == MCL_RAYS_SWEEP) { w = sweep_ok ? "configured: MCL_RAYS_SWEEP" : no_sweep; return sweep_ok ? 5 : 0; }
    // AUTO: one particle per lane on cell-sorted particles pays once there are enough particles to fill the machine
    // with 64-particle groups and enough rays to amortise the sort (measured, wall ms skip / quad / cell:
    // 4096 x 1081 0.16/0.28/0.40, 65536 x 1081 0.55/0.56/0.44, 65536 x 61 0.21/0.33/0.25, 262144 x 61 0.47/0.76/0.40);
    // below that the self-contained k_rays_skip (one launch, no work lists) is the quickest
    const int64_t cell_min = h->env_cell_min > 0 ? h->env_cell_min : 65536;
    const bool big = n >= cell_min && n * (int64_t)h->B >= (8 << 20);
    if (big && sweep_ok) {
        w = h->sweep_global ? "AUTO: at least 65536 particles and 2^23 rays, monotone beams; MAX_RANGE_PX > 243 (or MCL_SWEEP_GLOBAL): k_rays_sweep probes the wedge fields in global memory"
                            : "AUTO: at least 65536 particles and 2^23 rays, monotone beams, MAX_RANGE_PX <= 243";
        return 5;
    }
    if (big && windows_ok && h->cell_layout_ok) { w = no_sweep; return 4; }
    if (!h->skip_layout_ok) { w = "k_rays_skip's LDS layout check failed at mcl_create: literal march"; return 1; }
    w = !big ? "AUTO: fewer than 65536 particles or 2^23 rays: the self-contained k_rays_skip is the quickest" : no_sweep;
    return 2;
}

// The windowed pass over what k_rays_sweep's windows did not fit (k_rays_skip<.., FAR>): ordered list of the flagged slots,
// then persistent workgroups with the 568-cell nibble window.  Every kernel stands down on the device when the launch flagged
// fewer than kFarWindowedMin slots (the tracking regime: none), and k_rays_far stands down when this pass runs.
int launch_far_windowed(mcl_engine *h, const mcl::RayArgs &a, int64_t n, bool count)
{
    const unsigned nb = (unsigned)((n + mcl::kFarTile - 1) / mcl::kFarTile);
    const uint32_t *flags32 = reinterpret_cast<const uint32_t *>(h->d_far);
    hipLaunchKernelGGL(mcl::k_far_count, dim3(nb), dim3(256), 0, h->stream, flags32, n, a.far_count, h->d_far_cnt);
    hipLaunchKernelGGL(mcl::k_far_spine, dim3(1), dim3(1024), 0, h->stream, h->d_far_cnt, (int)nb, a.far_count);
    hipLaunchKernelGGL(mcl::k_far_scatter, dim3(nb), dim3(256), 0, h->stream, flags32, n, a.far_count, h->d_far_cnt, h->d_far_sorted);
    const size_t lds = (size_t)h->tw_cells * h->tw_cells / 2;
    if (count) hipLaunchKernelGGL((mcl::k_rays_skip<1, true, true>), dim3(h->num_cu), dim3(mcl::kRayThreads), lds, h->stream, a);
    // (one ray per lane: two / four in flight measured 52 / 67 ms against 46 on the uniform levine cloud)
    else hipLaunchKernelGGL((mcl::k_rays_skip<1, false, true>), dim3(h->num_cu), dim3(mcl::kRayThreads), lds, h->stream, a);
    HIPCHK(h, hipGetLastError());
    return MCL_OK;
}

int launch_rays(mcl_engine *h, const double *x, const double *y, const double *th, int64_t n, bool force_skip = false, bool direct_table = false)
{
    mcl::RayArgs a{};
    a.x = x; a.y = y; a.th = th; a.n = n;
    a.pc = h->d_pc;
    a.B = h->B; a.bpad = h->bpad; a.P = h->P;
    a.beam_cs = h->d_beam_cs; a.beam_angle = h->d_angle; a.Lt = h->d_Lt;
    a.Ltr = h->d_Lt + (size_t)(h->P + 1) * h->bpad;
    a.beam_a0 = h->B > 0 ? (double)h->angles[0] : 0.0;
    {
        const double span = h->B > 1 ? (double)h->angles[h->B - 1] - (double)h->angles[0] : 0.0;
        a.beam_inv_inc = span > 0.0 ? (double)(h->B - 1) / span : 0.0;
    }
    a.logw = h->d_logw;
    if (h->cfg.keep_ray_steps) {
        // one byte per step index up to 255 px of range, two beyond (allocated here: both the map and the beam set size it)
        const size_t need = (size_t)h->cap * h->B * (h->P > 255 ? 2 : 1);
        if (need > h->steps_capacity) {
            if (h->capturing) return fail(h, MCL_ERR_HIP, "step buffer missing during capture (internal)");
            graph_reset(h);
            dfree(h->d_steps);
            h->steps_capacity = 0;
            HIPCHK(h, hipMalloc(&h->d_steps, need));
            h->steps_capacity = need;
        }
        if (h->P > 255) a.steps16 = reinterpret_cast<uint16_t *>(h->d_steps);
        else a.steps = h->d_steps;
    }
    a.grid = h->d_grid; a.W = h->W; a.H = h->H;
    a.res = h->res; a.ox = h->ox; a.oy = h->oy;
    if (direct_table) { a.Ldirect = h->d_L; a.obs_idx = h->d_obs_idx; }       // small updates: no per-update table (do_update)
    a.dist = h->d_dist; a.dist4 = h->d_dist4; a.Wp = h->Wp; a.Hp = h->Hp; a.Wps = h->Wps;
    for (int q = 0; q < 4; ++q) a.distq[q] = h->d_distq[q];
    a.tw_cells = h->tw_cells;
    a.counters = h->d_counters;
    a.force_exact = h->cfg.debug_force_exact;
    h->last_quad = false;
    h->max_partials_ready = false;
    const int mode = choose_ray_mode(h, n, force_skip);
    if (mode == 0) return fail(h, MCL_ERR_UNSUPPORTED, "MCL_RAYS_QUAD / MCL_RAYS_CELL / MCL_RAYS_SWEEP not usable with this map / beam set");
    const bool windows = mode >= 3;                 // quadrant / wedge windows + work lists
    const bool cell = mode >= 4;                    // cell-sorted particles, one particle per lane
    const bool sweep = mode == 5;
    int64_t want = (n + 15) / 16;
    int grid = (int)std::max<int64_t>(1, std::min<int64_t>(h->num_cu, want));
    if (mode == 2 && !h->pc_ready)             // (an update's resampling kernel has already left them in d_pc otherwise)
        hipLaunchKernelGGL(mcl::k_particle_prep, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, x, y, th, n, h->ox, h->oy,
                           h->res, h->d_pc, h->d_angle, h->B, (short4 *)nullptr, mcl::PrepClear{});
    const bool count = h->cfg.debug_count_probes != 0;
    size_t lds = (size_t)h->tw_cells * h->tw_cells / 2;
    dim3 g(grid), b(mcl::kRayThreads);
    int R = h->cfg.rays_per_lane;
    if (R <= 0) R = 1;   // measured on MI355X: the kernel is VALU-issue-bound, extra chains per lane only add idle slots
    if (!windows && !h->capturing && !direct_table) HIPCHK(h, hipEventRecord(h->ev[EV_K0], h->stream));
    if (mode == 1) {
        if (count) hipLaunchKernelGGL((mcl::k_rays_march<true>), g, b, 0, h->stream, a);
        else hipLaunchKernelGGL((mcl::k_rays_march<false>), g, b, 0, h->stream, a);
    } else if (windows) {
        // work list for undecided rays (~0.06 % of the rays in practice): one segment per persistent workgroup of
        // k_rays_quad with room for 1/256 of that workgroup's share of the rays (at least 2048 entries)
        // item granularity: 32 slices per CU (measured best at 4M: 4/8/16/32/64 -> 22.1/20.2/19.7/19.6/20.2 ms), but at
        // least 256 particles per slice so that the 78 KB window load stays amortised
        const int spc = h->env_qslices_per_cu > 0 ? h->env_qslices_per_cu : 32;
        int nsl = (int)std::max<int64_t>(1, std::min<int64_t>((int64_t)spc * h->num_cu, (n + 255) / 256));
        int sweep_g = 1;
        if (cell) {
            // k_rays_cell: a slice is a run of the sorted order; 2048 particles = two 64-particle groups per wave.  Longer
            // slices amortise the window load better, shorter ones balance the persistent workgroups better
            // (measured at 4M: 1024/2048/4096/8192/16384 -> 8.59/8.04/7.86/7.96/8.64 ms; at 256k: 2048/4096 -> 0.68/0.79 ms)
            int64_t slice_len = h->env_cell_slice > 0 ? std::max<int64_t>(64, h->env_cell_slice) : (n >= (1 << 21) ? 4096 : 2048);
            nsl = (int)std::max<int64_t>(1, (n + slice_len - 1) / slice_len);
        }
        const int max_wg = 2 * (h->num_cu - h->reserved_cus);             // persistent: 2 workgroups per CU
        if (sweep) {
            // k_rays_sweep: a work item is (run of 1024-particle units, G wedges); the G wedges of a group share one
            // partial-sum array.  G = 2 (measured at 4M x 1081, ray kernel / update ms, same device: G = 1 6.50 / 7.87,
            // 2 6.54 / 7.86, 4 6.69 / 7.97, 8 6.99 / 8.27, 16 7.62 / 8.90: a longer item is a longer tail); G = 1 up to 2M
            // particles, where the finer items balance better than the sixteen partial-sum arrays cost (262 144 x 1081:
            // 0.58 / 0.80 -> 0.53 / 0.77, 1M: 1.72 / 2.13 -> 1.63 / 2.07; at 4M G = 1 and 2 are level).
            const int64_t M = (n + mcl::kSwUnit - 1) / mcl::kSwUnit;
            // Round 3: with the per-wedge sums added atomically (no partial-sum array per group) G = 1 wins at every size
            // (4M x 1081, ms per update G = 1 / 2 / 4: 6.80 / 6.91 / 7.08; levine stand-in 7.89 / 8.00 / 8.24)
            sweep_g = h->env_sweep_g > 0 ? h->env_sweep_g : 1;
            if (sweep_g > mcl::kWedges || (mcl::kWedges % sweep_g) != 0) sweep_g = 1;
            nsl = (int)M;
        }
        const int items_per_slice = sweep ? mcl::kWedges / sweep_g : (cell ? mcl::kWedges : 4);
        const int nseg = (int)std::min<int64_t>(max_wg, items_per_slice * (int64_t)nsl);   // one segment per persistent workgroup
        unsigned long long rays_per_seg = (unsigned long long)n * h->B / nseg + 64;
        // (both forms of k_rays_sweep work with 32 fractional bits: a walk of ~70 rays is handed over once in several thousand)
        const unsigned long long seg_div = 256;
        unsigned long long segcap = std::max<unsigned long long>(2048, (rays_per_seg / seg_div + 7) & ~7ull);
        if ((unsigned long long)n * h->B <= (4ull << 20)) segcap = (2 * rays_per_seg + 7) & ~7ull;   // small launch: room for every ray
        if ((unsigned long long)nseg * segcap > h->fix_alloc) {
            dfree(h->d_fix_list);
            HIPCHK(h, hipMalloc(&h->d_fix_list, (size_t)nseg * segcap * 8));
            h->fix_alloc = (unsigned long long)nseg * segcap;
        }
        if ((size_t)nseg > h->fix_count_alloc) {
            dfree(h->d_fix_count);
            HIPCHK(h, hipMalloc(&h->d_fix_count, (size_t)nseg * 64));
            h->fix_count_alloc = nseg;
        }
        h->fix_cap = segcap;
        h->fix_segments = nseg;
        a.qr = cell ? nullptr : h->d_qr;
        a.qside = sweep ? mcl::kSwSide : h->qside;
        a.nslices = nsl;
        a.fix_list = h->d_fix_list; a.fix_count = h->d_fix_count; a.fix_cap = h->fix_cap; a.fix_segments = nseg;
        a.exact_list = h->d_exact_list; a.exact_count = h->d_result + 14; a.exact_cap = kExactCap;
        a.far_flags = h->d_far;
        a.work_counter = h->d_fix_over + 1;                // second word of the 16-byte scratch block
        a.logw = h->d_logw_acc;
        {   // per-particle constants; the same pass zeroes the stage's scratch (partial sums are added atomically)
            mcl::PrepClear clr{};
            clr.logw_acc = h->d_logw_acc; clr.far_flags = reinterpret_cast<uint32_t *>(h->d_far);
            clr.fix_count = h->d_fix_count; clr.fix_words = nseg * 8; clr.fix_over = h->d_fix_over; clr.exact_count = h->d_result + 14;
            if (sweep) clr.far_count = h->d_result + 15;
            const bool stale_layout = cell && h->layout_stale_used;      // d_bbox / d_tilemap hold the layout to order by: not remade
            if (cell && !stale_layout) clr.bbox = h->d_bbox;          // (the histogram is left all-zero by k_hist_clear of the previous sort)
            // the window play k_sweep_plan works with (0: no windowed kernel; -1: no cuts at all, MCL_NO_BUCKET_CUTS)
            clr.bbox_play = h->env_no_bucket_cuts ? -1 : (sweep ? sweep_play(h) : 0);
            if (cell && h->pc_ready) {         // the resampling kernel left the constants and zeroed the per-particle scratch
                clr.logw_acc = nullptr; clr.far_flags = nullptr;
                // ... and, from the second update of a configuration on, the few words that are not per particle as well
                const mcl::PrepClear &pc0 = h->prep_passed;
                const bool done = h->prep_folded && clr.fix_count == pc0.fix_count && clr.fix_words == pc0.fix_words &&
                                  clr.fix_over == pc0.fix_over && clr.exact_count == pc0.exact_count && clr.far_count == pc0.far_count &&
                                  clr.bbox == pc0.bbox && clr.bbox_play == pc0.bbox_play && !clr.hist && !pc0.hist;
                if (!done) hipLaunchKernelGGL(mcl::k_prep_small, dim3(1), dim3(256), 0, h->stream, clr);
                if (!h->capturing) { h->prep_cache = clr; h->prep_cache_valid = true; h->prep_cache_n = n; }
            } else {
                hipLaunchKernelGGL(mcl::k_particle_prep, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, x, y, th, n, h->ox, h->oy,
                                   h->res, h->d_pc, h->d_angle, h->B, cell ? (short4 *)nullptr : h->d_qr, clr);
            }
        }
        if (cell) {
            // order the particles by (tile, cell, heading): bounding box -> bucket histogram (the atomic's return value
            // is the rank inside the bucket) -> exclusive scan -> scatter of pc / qr / index
            const unsigned nb256 = (unsigned)((n + 255) / 256);
            const int nparts = (int)(mcl::kSortKeySpace / mcl::kHistTile);
            const int bstride = n >= (1 << 20) ? 16 : 1;
            // occupied tiles of the map are numbered compactly when the map has at most kSortMaxTiles of them (bbox[5] says so)
            const int ntx_abs = ((h->Wp * mcl::kSortSub - 1) >> 5) + 1, nty_abs = ((h->Hp * mcl::kSortSub - 1) >> 5) + 1;
            const bool tiles_ok = (int64_t)ntx_abs * nty_abs <= mcl::kSortMaxTiles;
            const bool stale_layout = h->layout_stale_used;
            if (!stale_layout) {
                hipLaunchKernelGGL(mcl::k_cell_bbox, dim3((unsigned)std::min<int64_t>((n / bstride + 255) / 256, 128)), dim3(256), 0,
                                   h->stream, h->d_pc, n, bstride, h->Wp, h->Hp, h->d_bbox, tiles_ok ? h->d_tilemark : (int *)nullptr, ntx_abs);
                if (tiles_ok)
                    hipLaunchKernelGGL(mcl::k_tile_compact, dim3(1), dim3(1024), 0, h->stream, h->d_bbox, h->d_tilemark, h->d_tilemap, ntx_abs * nty_abs);
            }
            // Two ways to the same kind of order (which lanes share a wave; never results).  Counting sort with per-XCD
            // histograms: one returning L2 atomic per particle (~1 per clock and XCD) + a scatter.  From 3M particles a radix
            // sort of (key, index) pairs is quicker -- keys only, rocPRIM's device sort (a plain library sort), then a gather:
            // 4M x 1081 6.57 -> 6.47 ms per update; below, its fixed cost of a dozen launches loses (2M +0.03, 262 144 +0.05 ms).
            const bool radix = h->env_sort_radix >= 0 ? h->env_sort_radix != 0 : n >= 3000000;
            if (radix) {
                if (!h->d_skey2) {
                    HIPCHK(h, hipMalloc(&h->d_skey2, (size_t)h->cap * 4));
                    HIPCHK(h, hipMalloc(&h->d_sval2, (size_t)h->cap * 4));
                }
                size_t tb = 0;
                HIPCHK(h, rocprim::radix_sort_pairs(nullptr, tb, h->d_skey, h->d_skey2, h->d_srank, h->d_sval2, (size_t)n, 0, mcl::kSortKeyLog2, h->stream));
                if (tb > h->sort_tmp_bytes) {
                    dfree(h->d_sort_tmp);
                    h->sort_tmp_bytes = 0;
                    HIPCHK(h, hipMalloc(&h->d_sort_tmp, tb));
                    h->sort_tmp_bytes = tb;
                }
                if (!(stale_layout && h->keys_done))       // (else the resampling kernel wrote the pairs)
                    hipLaunchKernelGGL(mcl::k_sort_keys, dim3(nb256), dim3(256), 0, h->stream, h->d_pc, th, n, h->Wp, h->Hp, h->d_bbox, h->d_skey, h->d_srank,
                                       h->d_tilemap, ntx_abs);
                tb = h->sort_tmp_bytes;
                HIPCHK(h, rocprim::radix_sort_pairs(h->d_sort_tmp, tb, h->d_skey, h->d_skey2, h->d_srank, h->d_sval2, (size_t)n, 0, mcl::kSortKeyLog2, h->stream));
                hipLaunchKernelGGL(mcl::k_sort_gather, dim3(nb256), dim3(256), 0, h->stream, h->d_pc, th, n, h->d_sval2, h->d_pcs, h->d_ths, h->d_perm,
                                   sweep ? h->d_skey2 : (const uint32_t *)nullptr, h->d_bbox, h->d_cut_start, h->d_cut_end);
            } else {
            hipLaunchKernelGGL(mcl::k_sort_hist, dim3(nb256), dim3(256), 0, h->stream, h->d_pc, th, n, h->Wp, h->Hp, h->d_bbox, h->d_hist,
                               h->d_skey, h->d_srank, h->d_tile_used, h->d_tilemap, ntx_abs);
            hipLaunchKernelGGL(mcl::k_hist_partials, dim3(nparts), dim3(256), 0, h->stream, h->d_hist, h->d_histpart, h->d_tile_used);
            hipLaunchKernelGGL(mcl::k_hist_spine, dim3(1), dim3(1024), 0, h->stream, h->d_histpart, nparts);
            hipLaunchKernelGGL(mcl::k_hist_final, dim3(nparts), dim3(256), 0, h->stream, h->d_hist, h->d_histpart, h->d_tile_used);
            hipLaunchKernelGGL(mcl::k_sort_scatter, dim3(nb256), dim3(256), 0, h->stream, h->d_pc, th, n, h->d_skey, h->d_srank,
                               h->d_hist, h->d_pcs, h->d_ths, h->d_perm);
            }
            if (sweep) {
                // units of the sorted order (cut at tile borders when the set is ordered by whole tiles), from the bucket
                // offsets the scatter has just used -- before they are cleared
                const size_t mu = (size_t)max_sweep_units(n);
                if (mu > h->unit_sums_capacity) {
                    dfree(h->d_unit_sums); dfree(h->d_unit_begin);
                    h->unit_sums_capacity = 0;
                    HIPCHK(h, hipMalloc(&h->d_unit_sums, mu * 2 * sizeof(double4)));
                    HIPCHK(h, hipMalloc(&h->d_unit_begin, (mu + 1) * sizeof(uint32_t)));
                    h->unit_sums_capacity = mu;
                }
                if (!h->d_nunits) HIPCHK(h, hipMalloc(&h->d_nunits, sizeof(int)));
                hipLaunchKernelGGL(mcl::k_unit_table, dim3(1), dim3(1024), 0, h->stream, h->d_bbox, n, h->d_hist, h->d_histpart, h->d_tile_used,
                                   radix ? h->d_cut_start : (uint32_t *)nullptr, h->d_cut_end, h->d_unit_begin, h->d_nunits, (int)mu);
            }
            if (!radix) hipLaunchKernelGGL(mcl::k_hist_clear, dim3(nparts), dim3(256), 0, h->stream, h->d_hist, h->d_tile_used);
            if (sweep) {
                hipLaunchKernelGGL(mcl::k_unit_sums, dim3((unsigned)std::min<int64_t>(max_sweep_units(n), (n + mcl::kSwUnit - 1) / mcl::kSwUnit + 256)), dim3(256), 0, h->stream, h->d_pcs, h->d_unit_begin, h->d_nunits,
                                   h->d_unit_sums);
            } else {
                if ((size_t)nsl > h->slice_mean_capacity) {
                    dfree(h->d_slice_mean);
                    HIPCHK(h, hipMalloc(&h->d_slice_mean, (size_t)nsl * sizeof(double2)));
                    h->slice_mean_capacity = nsl;
                }
                hipLaunchKernelGGL(mcl::k_slice_means, dim3((unsigned)nsl), dim3(256), 0, h->stream, h->d_pcs, n, (n + nsl - 1) / nsl, h->d_slice_mean);
            }
            a.pcs = h->d_pcs; a.ths = h->d_ths; a.perm = h->d_perm; a.slice_mean = h->d_slice_mean;
            a.distw = h->d_distw; a.distw_stride = (size_t)h->Hp * h->Wps;
        }
        if (sweep) {
            if (!h->d_Ltd) return fail(h, MCL_ERR_HIP, "k_rays_sweep: table not allocated (internal)");
            if (!h->ltd_ready) build_ltd(h);          // a caller whose table decision was made for another particle count
            const int rc_plan = launch_sweep_plan(h, n, nseg, sweep_g);
            if (rc_plan) return rc_plan;
            a.sweep_g = sweep_g; a.Ltd = h->d_Ltd; a.ltd_cols = h->ltd_cols;
            a.split16 = h->env_sw_split16 >= 0 ? h->env_sw_split16 : 0;
            a.beam_csx = h->sweep_global ? h->d_beam_csxg : h->d_beam_csx; a.beam_pad = h->beam_pad; a.beam_margin = h->beam_margin;
            a.beam_csi = h->d_beam_csi; a.beam_err = h->d_beam_err; a.rec_k = 2.0 * h->rec_c;
            a.distg = h->d_distg; a.distg_stride = h->distg_stride; a.distg_pitch = h->distg_pitch;
            a.items = h->d_items; a.centres = h->d_centres; a.nitems = 0; a.nitems_ptr = h->d_nitems; a.unit_sums = h->d_unit_sums; a.unit_begin = h->d_unit_begin; a.slot_space = 1;
            if (!h->d_far_list) {
                HIPCHK(h, hipMalloc(&h->d_far_list, (size_t)h->cap * sizeof(uint32_t)));
                HIPCHK(h, hipMalloc(&h->d_far_sorted, (size_t)h->cap * sizeof(uint32_t)));
                HIPCHK(h, hipMalloc(&h->d_far_cnt, ((size_t)h->cap / mcl::kFarTile + 2) * sizeof(uint32_t)));
            }
            // the windowed far pass (four launches that stand down on the device when little is flagged) is only launched when
            // there is reason to expect work for it: the previous ray stage flagged a fair number of slots, or the particle set
            // is fresh (set / initialised since).  A misjudgement costs time, never results: without it k_rays_far takes all.
            a.far_sorted = h->d_far_sorted;
            a.far_windowed = (h->far_fresh || h->h_result[15] >= mcl::kFarWindowedMin / 2) ? 1 : 0;
            h->far_fresh = false;
            a.far_list = h->d_far_list; a.far_count = h->d_result + 15;      // word 15 of the result block, zeroed below
        }
        const bool sweep_glob = sweep && h->sweep_global;       // probes in global memory: no window in LDS
        // REC: the walk turns the beam direction by the scan's increment instead of fetching it (evenly spaced scans; the beams'
        // offsets from the grid sit in LDS behind the window: 8 bytes per table column)
        // (... as long as two workgroups still fit a CU's 160 KB: up to ~1800 table columns; more beams than that fetch their directions)
        const bool sweep_rec = sweep && h->rec_ok && h->sweep_rec_layout_ok && h->d_beam_err != nullptr &&
                               (size_t)mcl::kSwSide * mcl::kSwSide + (size_t)h->ltd_cols * 8 <= 80 * 1024 - 64;
        // PAIRS (two rays per lane) where the walk waits for memory: always in the global-field form; in LDS windows for a set
        // that was set / initialised since the last update (the spread cloud of a re-localisation: -11 % on its first update, where
        // the tracking cloud gains nothing and the levine stand-in loses 2 %); MCL_SWEEP_PAIRS=0 / 1 overrides
        const bool sweep_pairs = sweep_rec && (h->env_sweep_pairs >= 0 ? h->env_sweep_pairs != 0 : (sweep_glob || a.far_windowed != 0));
        size_t qlds = sweep ? (sweep_glob ? 0 : (size_t)mcl::kSwSide * mcl::kSwSide) + (sweep_rec ? (size_t)h->ltd_cols * 8 : 0) : (size_t)h->qside * h->qside;
        h->last_sweep_global = sweep_glob ? 1 : 0; h->last_sweep_rec = sweep_rec ? 1 : 0; h->last_sweep_pairs = (sweep_rec && (sweep_glob || sweep_pairs)) ? 1 : 0;
        dim3 qg((unsigned)nseg);   // persistent: 2 workgroups per CU
        unsigned long long *d_dbg = nullptr;
        const char *dbgpath = h->env_debug_wg.empty() ? nullptr : h->env_debug_wg.c_str();
        if (dbgpath) { HIPCHK(h, hipMalloc(&d_dbg, (size_t)qg.x * 32)); HIPCHK(h, hipMemset(d_dbg, 0, (size_t)qg.x * 32)); a.dbg = d_dbg; }
        // k_rays_far is bound by global-memory latency: 4 workgroups per CU worth of blocks (2 resident at a time)
        dim3 gfar((unsigned)std::max<int64_t>(1, std::min<int64_t>(4 * (int64_t)h->num_cu, (n + 15) / 16)));
        const int fix_split = std::max(1, std::min(16, (8 * h->num_cu) / std::max(nseg, 1)));   // ~8 workgroups of k_rays_fix per CU (2 .. 16 per segment: no difference, round 4)
        if (!sweep) HIPCHK(h, hipEventRecord(h->ev[EV_K0], h->stream));
        if (count) {
            if (sweep_glob && sweep_rec) hipExtLaunchKernelGGL((mcl::k_rays_sweep<true, true, true, true>), qg, b, qlds, h->stream, nullptr, h->ev[EV_K1], 0, a);
            else if (sweep_glob) hipExtLaunchKernelGGL((mcl::k_rays_sweep<true, true>), qg, b, qlds, h->stream, nullptr, h->ev[EV_K1], 0, a);
            else if (sweep && sweep_rec && sweep_pairs) hipExtLaunchKernelGGL((mcl::k_rays_sweep<true, false, true, true>), qg, b, qlds, h->stream, nullptr, h->ev[EV_K1], 0, a);
            else if (sweep && sweep_rec) hipExtLaunchKernelGGL((mcl::k_rays_sweep<true, false, true>), qg, b, qlds, h->stream, nullptr, h->ev[EV_K1], 0, a);
            else if (sweep) hipExtLaunchKernelGGL((mcl::k_rays_sweep<true>), qg, b, qlds, h->stream, nullptr, h->ev[EV_K1], 0, a);
            else if (cell) hipLaunchKernelGGL((mcl::k_rays_cell<true>), qg, b, qlds, h->stream, a);
            else hipLaunchKernelGGL((mcl::k_rays_quad<true>), qg, b, qlds, h->stream, a);
            if (!sweep) HIPCHK(h, hipEventRecord(h->ev[EV_K1], h->stream));
            if (sweep && a.far_windowed) { const int rcw = launch_far_windowed(h, a, n, true); if (rcw) return rcw; }
            hipLaunchKernelGGL((mcl::k_rays_far<true>), gfar, b, 0, h->stream, a);
            hipLaunchKernelGGL((mcl::k_rays_fix<true>), dim3(nseg * fix_split), dim3(256), 0, h->stream, a);
            hipLaunchKernelGGL((mcl::k_rays_exact<true>), dim3(2 * h->num_cu), dim3(256), 0, h->stream, a);
        } else {
            if (sweep_glob && sweep_rec) hipExtLaunchKernelGGL((mcl::k_rays_sweep<false, true, true, true>), qg, b, qlds, h->stream, nullptr, h->ev[EV_K1], 0, a);
            else if (sweep_glob) hipExtLaunchKernelGGL((mcl::k_rays_sweep<false, true>), qg, b, qlds, h->stream, nullptr, h->ev[EV_K1], 0, a);
            else if (sweep && sweep_rec && sweep_pairs) hipExtLaunchKernelGGL((mcl::k_rays_sweep<false, false, true, true>), qg, b, qlds, h->stream, nullptr, h->ev[EV_K1], 0, a);
            else if (sweep && sweep_rec) hipExtLaunchKernelGGL((mcl::k_rays_sweep<false, false, true>), qg, b, qlds, h->stream, nullptr, h->ev[EV_K1], 0, a);
            else if (sweep) hipExtLaunchKernelGGL((mcl::k_rays_sweep<false>), qg, b, qlds, h->stream, nullptr, h->ev[EV_K1], 0, a);
            else if (cell) hipLaunchKernelGGL((mcl::k_rays_cell<false>), qg, b, qlds, h->stream, a);
            else hipLaunchKernelGGL((mcl::k_rays_quad<false>), qg, b, qlds, h->stream, a);
            if (!sweep) HIPCHK(h, hipEventRecord(h->ev[EV_K1], h->stream));
            if (sweep && a.far_windowed) { const int rcw = launch_far_windowed(h, a, n, false); if (rcw) return rcw; }
            hipLaunchKernelGGL((mcl::k_rays_far<false>), gfar, b, 0, h->stream, a);
            hipLaunchKernelGGL((mcl::k_rays_fix<false>), dim3(nseg * fix_split), dim3(256), 0, h->stream, a);
            hipLaunchKernelGGL((mcl::k_rays_exact<false>), dim3(2 * h->num_cu), dim3(256), 0, h->stream, a);
        }
        if (sweep) {
            // the slot accumulators (k_rays_sweep's per-wedge sums + what the far / fix / exact kernels added) -> d_logw in particle
            // order, the per-workgroup maxima, and the overflow flag of the fix-up lists (k_fix_overflow's job for the other kernels)
            hipExtLaunchKernelGGL(mcl::k_combine_logw, dim3(mcl::kRedBlocks), dim3(256), 0, h->stream, nullptr, h->ev[EV_RAYS], 0, n, h->d_perm, h->d_logw_acc,
                                  h->d_logw, h->d_maxpart, h->d_fix_count, nseg, segcap, h->d_fix_over);
            h->ev_rays_bound = true;               // (the stage's last kernel: EV_RAYS is its stop event)
            h->max_partials_ready = true;
        } else {
            hipLaunchKernelGGL(mcl::k_fix_overflow, dim3(1), dim3(256), 0, h->stream, h->d_fix_count, nseg, segcap, h->d_fix_over);
            hipLaunchKernelGGL(mcl::k_gather_logw, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, h->d_logw_acc, n, h->d_logw);
        }
        if (d_dbg) {
            std::vector<unsigned long long> hd((size_t)qg.x * 4);
            HIPCHK(h, hipStreamSynchronize(h->stream));
            HIPCHK(h, hipMemcpy(hd.data(), d_dbg, hd.size() * 8, hipMemcpyDeviceToHost));
            if (FILE *f = fopen(dbgpath, "wb")) { fwrite(hd.data(), 8, hd.size(), f); fclose(f); }
            (void)hipFree(d_dbg);
        }
        h->last_quad = true;
    } else if (count) {
        hipLaunchKernelGGL((mcl::k_rays_skip<1, true>), g, b, lds, h->stream, a);
    } else {
        switch (R) {
        case 1: hipLaunchKernelGGL((mcl::k_rays_skip<1, false>), g, b, lds, h->stream, a); break;
        case 2: hipLaunchKernelGGL((mcl::k_rays_skip<2, false>), g, b, lds, h->stream, a); break;
        case 3: hipLaunchKernelGGL((mcl::k_rays_skip<3, false>), g, b, lds, h->stream, a); break;
        default: hipLaunchKernelGGL((mcl::k_rays_skip<4, false>), g, b, lds, h->stream, a); break;
        }
    }
    if (!windows && !h->capturing && !direct_table) HIPCHK(h, hipEventRecord(h->ev[EV_K1], h->stream));
    h->last_mode = mode;
    h->prep_folded = false;
    h->layout_stale_used = false; h->keys_done = false;
    if (!h->capturing) h->pc_ready = false;
    HIPCHK(h, hipGetLastError());
    return MCL_OK;
}

// obs -> obs_idx upload + per-update transposed log table
void stage_observation(mcl_engine *h, const float *obs, int stride)
{
    h->ltd_ready = false;
    for (int j = 0; j < h->B; ++j) h->h_obs[j] = obs[(size_t)j * stride];   // cpp:316-320 when stride = ANGLE_STEP
}

// k_rays_sweep's fp64 table of the observation whose table rows are in d_obs_idx
void build_ltd(mcl_engine *h)
{
    dim3 gd((h->ltd_cols + 255) / 256, mcl::sweep_table_rows(h->P));
    hipLaunchKernelGGL(mcl::k_build_ltd, gd, dim3(256), 0, h->stream, h->d_L, h->d_obs_idx, h->B, h->ltd_cols, h->P, h->beam_margin, h->d_Ltd);
    h->ltd_ready = true;
}

// the pinned staging buffer -> obs_idx + per-update transposed log table
int upload_observation(mcl_engine *h)
{
    HIPCHK(h, hipMemcpyAsync(h->d_obs, h->h_obs, (size_t)h->B * sizeof(float), hipMemcpyHostToDevice, h->stream));
    int rc = h->capturing ? MCL_OK : ensure_lt(h);          // no allocation while a graph is being captured (sizes are warm)
    if (rc) return rc;
    dim3 g((h->bpad + 255) / 256, h->P + 1);
    hipLaunchKernelGGL(mcl::k_obs_build_lt, g, dim3(256), 0, h->stream, h->d_obs, h->res, h->P, h->d_L, h->B, h->bpad, h->d_obs_idx, h->d_Lt,
                       h->d_Lt + (size_t)(h->P + 1) * h->bpad);
    if (choose_ray_mode(h, h->N, false) == 5) build_ltd(h);
    HIPCHK(h, hipGetLastError());
    return MCL_OK;
}

int prepare_observation(mcl_engine *h, const float *obs, int stride)
{
    stage_observation(h, obs, stride);
    return upload_observation(h);
}

void graph_reset(mcl_engine *h)
{
    for (int k = 0; k < 2; ++k)
        if (h->graph_exec[k]) { (void)hipGraphExecDestroy(h->graph_exec[k]); h->graph_exec[k] = nullptr; }
    h->graph_warm = false;
    h->prep_cache_valid = false; h->prep_folded = false;
    h->layout_valid = false; h->layout_stale_used = false; h->keys_done = false;
    h->pc_ready = false;                    // whatever changed (map, beams, particles, a buffer): the ray stage makes its own constants
}

int sensor_and_weights(mcl_engine *h, const double *d_global_max, bool defer_sums = false)
{
    // d_logw holds the log-weights of the current particle set
    if (h->cfg.weight_mode == MCL_WEIGHT_PRODUCT) {
        if (!h->cfg.keep_ray_steps) return fail(h, MCL_ERR_UNSUPPORTED, "weight_mode PRODUCT needs keep_ray_steps");
        int64_t n = h->N;
        hipLaunchKernelGGL(mcl::k_product_weights, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, h->P > 255 ? (const uint8_t *)nullptr : h->d_steps,
                           h->P > 255 ? reinterpret_cast<const uint16_t *>(h->d_steps) : (const uint16_t *)nullptr, h->d_obs_idx, n, h->B, h->d_table, h->P + 1, 1.0 / h->cfg.squash_factor, h->d_w);
        HIPCHK(h, hipGetLastError());
        return weight_stats(h, false, nullptr, defer_sums);
    }
    return weight_stats(h, true, d_global_max, defer_sums);
}

// weights, sums and the CDF of the current log-weights (the tail of an update).  Small updates take one launch.
int weights_and_cdf(mcl_engine *h, bool result_to_host = false)
{
    const int64_t n = h->N;
    if (h->cfg.weight_mode == MCL_WEIGHT_LOG && h->cfg.resample_neff_permille == 0 && n <= mcl::kTinyTailMax) {
        // d_pc holds (cos, sin) of the current headings whenever a ray kernel other than the literal march ran on them
        const double4 *pc = h->last_mode >= 2 ? h->d_pc : nullptr;
        hipLaunchKernelGGL(mcl::k_tiny_tail, dim3(1), dim3(1024), (size_t)n * sizeof(uint64_t), h->stream, h->d_logw, h->d_x[h->cur],
                           h->d_y[h->cur], h->d_th[h->cur], pc, n, h->d_w, h->d_q, h->d_cdf, h->d_scalars,
                           result_to_host ? h->h_result : (unsigned long long *)nullptr, ++h->result_seq);
        HIPCHK(h, hipGetLastError());
        h->max_partials_ready = false;
        h->carry_pending = false;
        h->blocktot_for = nullptr;             // no spine / leaders for this CDF: the resampling search bisects it directly
        h->compact_n = -1; h->compact_pending = false; h->list_epoch++;
        return MCL_OK;
    }
    int rc = sensor_and_weights(h, nullptr, true);             // (the sums are finished by the scan's spine)
    if (rc) return rc;
    return scan_weights(h, h->d_q, h->d_cdf, n, 0, nullptr);   // CDF for the next resample / visualize
}

bool ready(mcl_engine *h, bool need_particles)
{
    return h && h->have_map && h->B > 0 && (!need_particles || h->have_particles);
}

float elapsed(hipEvent_t a, hipEvent_t b)
{
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, a, b);
    return ms;
}

}  // namespace

extern "C" {

int mcl_abi_version(void) { return MCL_ABI_VERSION; }

void mcl_default_config(mcl_config_t *c)
{
    if (!c) return;
    std::memset(c, 0, sizeof(*c));
    c->max_particles = 2000;       // cpp:24, yaml:6
    c->device = 0;
    c->seed = 0;
    c->max_range_m = 12.0;         // cpp:27
    c->z_hit = 0.80; c->z_short = 0.01; c->z_max = 0.07; c->z_rand = 0.12; c->sigma_hit = 8.0;   // cpp:30-34
    c->squash_factor = 2.2;        // cpp:26
    c->motion_dispersion_x = 0.05; c->motion_dispersion_y = 0.025; c->motion_dispersion_theta = 0.25;   // cpp:35-37
    c->resample_mode = MCL_RESAMPLE_MULTINOMIAL;
    c->weight_mode = MCL_WEIGHT_LOG;
    c->ray_kernel = MCL_RAYS_AUTO;
}

const char *mcl_last_error(const mcl_engine_t *h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int mcl_create(const mcl_config_t *cfg, mcl_engine_t **out)
{
    if (cfg && (cfg->resample_neff_permille < 0 || cfg->resample_neff_permille > 1000)) return MCL_ERR_INVALID_ARG;
    g_create_error.clear();
    if (!cfg || !out) { g_create_error = "null argument"; return MCL_ERR_INVALID_ARG; }
    *out = nullptr;
    // weights are quantised to 2^-36 and summed in uint64 (E5/E6): N * 2^36 must stay below 2^64 with a bit to spare
    if (cfg->max_particles <= 0 || cfg->max_particles >= MCL_MAX_TOTAL_PARTICLES || cfg->squash_factor <= 0 || cfg->max_range_m <= 0) {
        g_create_error = "bad config (max_particles must be in [1, 2^27) / squash_factor / max_range_m)";
        return MCL_ERR_INVALID_ARG;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || cfg->device < 0 || cfg->device >= ndev) {
        g_create_error = "no HIP device (this engine has no CPU path)";
        return MCL_ERR_NO_DEVICE;
    }
    hipDeviceProp_t prop;
    if (hipSetDevice(cfg->device) != hipSuccess || hipGetDeviceProperties(&prop, cfg->device) != hipSuccess) {
        g_create_error = "hipSetDevice/hipGetDeviceProperties failed";
        return MCL_ERR_NO_DEVICE;
    }
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        g_create_error = std::string("device is ") + prop.gcnArchName + ", engine is built for gfx950 only";
        return MCL_ERR_NO_DEVICE;
    }
    mcl_engine *h = new mcl_engine();
    h->cfg = *cfg;
    // tuning knobs of the environment, read once here (never on the update path)
    if (const char *e = getenv("MCL_CELL_MIN")) h->env_cell_min = atoll(e);
    if (const char *e = getenv("MCL_CELL_SLICE")) h->env_cell_slice = atoll(e);
    if (const char *e = getenv("MCL_QSLICES_PER_CU")) h->env_qslices_per_cu = atoi(e);
    if (const char *e = getenv("MCL_QSIDE")) h->env_qside = atoi(e);
    if (const char *e = getenv("MCL_SWEEP_G")) h->env_sweep_g = atoi(e);
    if (const char *e = getenv("MCL_TINY_POLL")) h->env_tiny_poll = atoi(e);
    if (const char *e = getenv("MCL_NO_COMPACT")) h->env_no_compact = atoi(e);
    if (const char *e = getenv("MCL_SORT")) h->env_sort_radix = std::strcmp(e, "radix") == 0 ? 1 : (std::strcmp(e, "hist") == 0 ? 0 : -1);
    if (const char *e = getenv("MCL_DEBUG_WG")) h->env_debug_wg = e;
    h->env_no_bucket_cuts = getenv("MCL_NO_BUCKET_CUTS") != nullptr;
    if (const char *e = getenv("MCL_SWEEP_GLOBAL")) h->env_sweep_global = atoi(e) != 0;
    if (const char *e = getenv("MCL_SW_SPLIT16")) h->env_sw_split16 = atoi(e) != 0;
    if (const char *e = getenv("MCL_SWEEP_PAIRS")) h->env_sweep_pairs = atoi(e) != 0 ? 1 : 0;
    h->env_no_obs_overlap = getenv("MCL_NO_OBS_OVERLAP") != nullptr;
    h->env_no_prep_fold = getenv("MCL_NO_PREP_FOLD") != nullptr;
    h->env_no_stale_layout = getenv("MCL_NO_STALE_LAYOUT") != nullptr;
    h->env_comm_no_pregather = getenv("MCL_COMM_NO_PREGATHER") != nullptr;
    h->env_comm_no_lists = getenv("MCL_COMM_NO_LISTS") != nullptr;
    h->num_cu = prop.multiProcessorCount;
    h->cap = cfg->max_particles;
    auto bail = [&](const char *what) {
        g_create_error = std::string(what) + ": " + h->err;
        mcl_destroy(h);
        return MCL_ERR_HIP;
    };
#define CRT(call)                                                         \
    do {                                                                  \
        hipError_t e_ = (call);                                           \
        if (e_ != hipSuccess) { h->err = hipGetErrorString(e_); return bail(#call); } \
    } while (0)
    CRT(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    CRT(hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking));
    CRT(hipEventCreateWithFlags(&h->ev_obs, hipEventDisableTiming));
    for (int i = 0; i < EV_COUNT; ++i) CRT(hipEventCreate(&h->ev[i]));
    const size_t nb = (size_t)h->cap * sizeof(double);
    for (int b = 0; b < 2; ++b) {
        CRT(hipMalloc(&h->d_x[b], nb)); CRT(hipMalloc(&h->d_y[b], nb)); CRT(hipMalloc(&h->d_th[b], nb));
        CRT(hipMalloc(&h->d_pack[b], (size_t)h->cap * sizeof(double4)));
    }
    CRT(hipMalloc(&h->d_w, nb)); CRT(hipMalloc(&h->d_logw, nb)); CRT(hipMalloc(&h->d_tmp, nb * 3));
    CRT(hipMalloc(&h->d_logw_acc, nb));
    CRT(hipMalloc(&h->d_carry[0], nb)); CRT(hipMalloc(&h->d_carry[1], nb));
    CRT(hipMalloc(&h->d_q, (size_t)h->cap * 8)); CRT(hipMalloc(&h->d_cdf, (size_t)h->cap * 8));
    h->blocktot_capacity = (size_t)h->cap / mcl::kScanTile + 2;
    CRT(hipMalloc(&h->d_blocktot, h->blocktot_capacity * 8));
    CRT(hipMalloc(&h->d_blockcnt, h->blocktot_capacity * 4));
    h->compact_cap = std::max<int64_t>(4096, ((h->cap / 4 + 63) / 64) * 64);
    CRT(hipMalloc(&h->d_ccdf, (size_t)h->compact_cap * 8));
    CRT(hipMalloc(&h->d_ctop, ((size_t)h->compact_cap / 64 + 1) * 8));
    CRT(hipMalloc(&h->d_cidx, (size_t)h->compact_cap * 4));
    CRT(hipMalloc(&h->d_crec, (size_t)h->compact_cap * sizeof(double4)));
    CRT(hipMalloc(&h->d_idx, (size_t)h->cap * 4));
    CRT(hipMalloc(&h->d_part, (size_t)mcl::kRedBlocks * 8 * sizeof(double)));
    CRT(hipMalloc(&h->d_maxpart, (size_t)mcl::kRedBlocks * sizeof(double)));
    CRT(hipMalloc(&h->d_result, 32 * 8));
    CRT(hipMemset(h->d_result, 0, 32 * 8));
    CRT(hipHostMalloc(&h->h_result, 48 * 8));
    std::memset(h->h_result, 0, 48 * 8);
    h->d_scalars = reinterpret_cast<double *>(h->d_result);
    h->d_counters = h->d_result + 8;
    h->d_fix_over = h->d_result + 12;
    CRT(hipMalloc(&h->d_exact_list, (size_t)kExactCap * 8));
    CRT(hipMalloc(&h->d_inject, nb * 4));
    CRT(hipMalloc(&h->d_pc, (size_t)h->cap * sizeof(double4)));
    CRT(hipMalloc(&h->d_qr, (size_t)h->cap * sizeof(short4)));
    CRT(hipMalloc(&h->d_far, (size_t)h->cap * 4));
    CRT(hipMalloc(&h->d_pcs, (size_t)h->cap * sizeof(double4)));
    CRT(hipMalloc(&h->d_ths, (size_t)h->cap * sizeof(double)));
    CRT(hipMalloc(&h->d_perm, (size_t)h->cap * 4));
    CRT(hipMalloc(&h->d_skey, (size_t)h->cap * 4));
    CRT(hipMalloc(&h->d_srank, (size_t)h->cap * 4));
    CRT(hipMalloc(&h->d_hist, (size_t)mcl::kSortBuckets * 4));
    CRT(hipMalloc(&h->d_histpart, (size_t)(mcl::kSortKeySpace / mcl::kHistTile) * 4));
    CRT(hipMalloc(&h->d_tile_used, (size_t)(mcl::kSortKeySpace / mcl::kHistTile) * 4));
    CRT(hipMemset(h->d_hist, 0, (size_t)mcl::kSortBuckets * 4));          // kept all-zero between sorts (k_hist_clear)
    CRT(hipMemset(h->d_tile_used, 0, (size_t)(mcl::kSortKeySpace / mcl::kHistTile) * 4));
    CRT(hipMalloc(&h->d_bbox, 8 * sizeof(int)));
    CRT(hipMalloc(&h->d_bbox_nx, 8 * sizeof(int)));
    CRT(hipMalloc(&h->d_tilemap_nx, (size_t)mcl::kSortMaxTiles * sizeof(int)));
    CRT(hipMalloc(&h->d_tilemark_nx, (size_t)mcl::kSortMaxTiles * sizeof(int)));
    CRT(hipMemset(h->d_tilemark_nx, 0, (size_t)mcl::kSortMaxTiles * sizeof(int)));
    CRT(hipEventCreateWithFlags(&h->ev_children, hipEventDisableTiming));
    CRT(hipEventCreateWithFlags(&h->ev_layout, hipEventDisableTiming));
    CRT(hipEventCreateWithFlags(&h->ev_ext_in, hipEventDisableTiming));
    CRT(hipEventCreateWithFlags(&h->ev_ext_out, hipEventDisableTiming));
    CRT(hipMalloc(&h->d_cut_start, (size_t)mcl::kSwMaxCuts * sizeof(uint32_t)));
    CRT(hipMalloc(&h->d_cut_end, (size_t)mcl::kSwMaxCuts * sizeof(uint32_t)));
    CRT(hipMemset(h->d_cut_start, 0, (size_t)mcl::kSwMaxCuts * sizeof(uint32_t)));
    CRT(hipMemset(h->d_cut_end, 0, (size_t)mcl::kSwMaxCuts * sizeof(uint32_t)));
    CRT(hipMalloc(&h->d_tilemap, (size_t)mcl::kSortMaxTiles * sizeof(int)));
    CRT(hipMalloc(&h->d_tilemark, (size_t)mcl::kSortMaxTiles * sizeof(int)));
    CRT(hipMemset(h->d_tilemark, 0, (size_t)mcl::kSortMaxTiles * sizeof(int)));
    CRT(hipMemset(h->d_fix_over, 0, 16));
    CRT(hipMemset(h->d_scalars, 0, 8 * sizeof(double)));
    CRT(hipMemset(h->d_counters, 0, 4 * sizeof(unsigned long long)));
    CRT(hipDeviceSynchronize());        // the memsets above ran on the null stream; everything later uses h->stream
    CRT(hipFuncSetAttribute(reinterpret_cast<const void *>(&mcl::k_rays_skip<1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CRT(hipFuncSetAttribute(reinterpret_cast<const void *>(&mcl::k_rays_skip<1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CRT(hipFuncSetAttribute(reinterpret_cast<const void *>(&mcl::k_rays_skip<1, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CRT(hipFuncSetAttribute(reinterpret_cast<const void *>(&mcl::k_rays_skip<1, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CRT(hipFuncSetAttribute(reinterpret_cast<const void *>(&mcl::k_rays_skip<2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CRT(hipFuncSetAttribute(reinterpret_cast<const void *>(&mcl::k_rays_skip<3, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CRT(hipFuncSetAttribute(reinterpret_cast<const void *>(&mcl::k_rays_skip<4, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CRT(hipFuncSetAttribute(reinterpret_cast<const void *>(&mcl::k_rays_quad<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    CRT(hipFuncSetAttribute(reinterpret_cast<const void *>(&mcl::k_rays_quad<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    CRT(hipFuncSetAttribute(reinterpret_cast<const void *>(&mcl::k_rays_cell<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    CRT(hipFuncSetAttribute(reinterpret_cast<const void *>(&mcl::k_rays_cell<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    CRT(hipFuncSetAttribute(reinterpret_cast<const void *>(&mcl::k_tiny_tail), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    CRT(hipFuncSetAttribute(reinterpret_cast<const void *>(&mcl::k_sweep_plan), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    CRT(hipFuncSetAttribute(reinterpret_cast<const void *>(&mcl::k_rays_sweep<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    CRT(hipFuncSetAttribute(reinterpret_cast<const void *>(&mcl::k_rays_sweep<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    CRT(hipFuncSetAttribute(reinterpret_cast<const void *>(&mcl::k_rays_sweep<false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024 - 64));
    CRT(hipFuncSetAttribute(reinterpret_cast<const void *>(&mcl::k_rays_sweep<true, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024 - 64));
    CRT(hipFuncSetAttribute(reinterpret_cast<const void *>(&mcl::k_rays_sweep<false, false, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024 - 64));
    CRT(hipFuncSetAttribute(reinterpret_cast<const void *>(&mcl::k_rays_sweep<true, false, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024 - 64));
    {   // The hand-written probe loops address their LDS window from a raw offset: k_rays_sweep / k_rays_cell / k_rays_quad
        // from kQLdsBase (their static LDS must end exactly there), k_rays_skip from 0 (it must have no static LDS).  A
        // toolchain that lays a kernel out differently takes that kernel out of choose_ray_mode's choices -- the engine then
        // runs the next one down, same results -- so that no configuration can reach a malformed launch.
        auto static_lds_is = [](const void *fn, size_t want) {
            hipFuncAttributes fa{};
            return hipFuncGetAttributes(&fa, fn) == hipSuccess && fa.sharedSizeBytes == want;
        };
        const size_t qb = (size_t)mcl::kQLdsBase;
        h->sweep_layout_ok = static_lds_is(reinterpret_cast<const void *>(&mcl::k_rays_sweep<false>), qb) && static_lds_is(reinterpret_cast<const void *>(&mcl::k_rays_sweep<true>), qb);
        {
            // (the global form has one static word less: its dynamic LDS -- 16-byte aligned -- still starts at kQLdsBase)
            auto static_lds_fits = [](const void *fn, size_t lo, size_t hi) {
                hipFuncAttributes fa{};
                return hipFuncGetAttributes(&fa, fn) == hipSuccess && fa.sharedSizeBytes > lo && fa.sharedSizeBytes <= hi;
            };
            h->sweep_rec_layout_ok = static_lds_is(reinterpret_cast<const void *>(&mcl::k_rays_sweep<false, false, true>), qb) &&
                                     static_lds_is(reinterpret_cast<const void *>(&mcl::k_rays_sweep<true, false, true>), qb) &&
                                     static_lds_is(reinterpret_cast<const void *>(&mcl::k_rays_sweep<false, false, true, true>), qb) &&
                                     static_lds_is(reinterpret_cast<const void *>(&mcl::k_rays_sweep<true, false, true, true>), qb) &&
                                     static_lds_fits(reinterpret_cast<const void *>(&mcl::k_rays_sweep<false, true, true, true>), 0, qb) &&
                                     static_lds_fits(reinterpret_cast<const void *>(&mcl::k_rays_sweep<true, true, true, true>), 0, qb);
        }
        h->cell_layout_ok = static_lds_is(reinterpret_cast<const void *>(&mcl::k_rays_cell<false>), qb) && static_lds_is(reinterpret_cast<const void *>(&mcl::k_rays_cell<true>), qb);
        h->quad_layout_ok = static_lds_is(reinterpret_cast<const void *>(&mcl::k_rays_quad<false>), qb) && static_lds_is(reinterpret_cast<const void *>(&mcl::k_rays_quad<true>), qb);
        h->skip_layout_ok = static_lds_is(reinterpret_cast<const void *>(&mcl::k_rays_skip<1, false>), 0) && static_lds_is(reinterpret_cast<const void *>(&mcl::k_rays_skip<1, true>), 0) &&
                            static_lds_is(reinterpret_cast<const void *>(&mcl::k_rays_skip<1, false, true>), 0) && static_lds_is(reinterpret_cast<const void *>(&mcl::k_rays_skip<1, true, true>), 0) &&
                            static_lds_is(reinterpret_cast<const void *>(&mcl::k_rays_skip<2, false>), 0) && static_lds_is(reinterpret_cast<const void *>(&mcl::k_rays_skip<3, false>), 0) &&
                            static_lds_is(reinterpret_cast<const void *>(&mcl::k_rays_skip<4, false>), 0);
        (void)hipGetLastError();
    }
#undef CRT
    *out = h;
    return MCL_OK;
}

void mcl_destroy(mcl_engine_t *h)
{
    if (!h) return;
    (void)hipSetDevice(h->cfg.device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->comm) { comm_free(h->comm); h->comm = nullptr; }
    graph_reset(h);
    for (int b = 0; b < 2; ++b) { dfree(h->d_x[b]); dfree(h->d_y[b]); dfree(h->d_th[b]); }
    dfree(h->d_w); dfree(h->d_logw); dfree(h->d_tmp); dfree(h->d_logw_acc); dfree(h->d_carry[0]); dfree(h->d_carry[1]); dfree(h->d_q); dfree(h->d_cdf); dfree(h->d_blocktot); dfree(h->d_bm); dfree(h->d_bm_pop); dfree(h->d_bm_pref);
    dfree(h->d_gcdf); dfree(h->d_gtop);
    dfree(h->d_blockcnt); dfree(h->d_ccdf); dfree(h->d_ctop); dfree(h->d_cidx); dfree(h->d_crec);
    dfree(h->d_idx); dfree(h->d_steps); dfree(h->d_part); dfree(h->d_maxpart); dfree(h->d_result); if (h->h_result) { (void)hipHostFree(h->h_result); h->h_result = nullptr; } dfree(h->d_inject); dfree(h->d_pc); dfree(h->d_qr); dfree(h->d_far); dfree(h->d_far_list); dfree(h->d_far_sorted); dfree(h->d_far_cnt); dfree(h->d_pcs); dfree(h->d_ths); dfree(h->d_distw); dfree(h->d_distg); dfree(h->d_leaders); dfree(h->d_pack[0]); dfree(h->d_pack[1]); dfree(h->d_perm); dfree(h->d_skey); dfree(h->d_srank); dfree(h->d_skey2); dfree(h->d_sval2); dfree(h->d_sort_tmp); dfree(h->d_hist); dfree(h->d_histpart); dfree(h->d_tile_used); dfree(h->d_bbox); dfree(h->d_cut_start); dfree(h->d_cut_end); dfree(h->d_tilemap); dfree(h->d_tilemark); dfree(h->d_slice_mean); dfree(h->d_fix_list); dfree(h->d_fix_count); dfree(h->d_exact_list);
    dfree(h->d_grid); dfree(h->d_dist); dfree(h->d_dist4); dfree(h->d_L); dfree(h->d_table);
    for (int q = 0; q < 4; ++q) dfree(h->d_distq[q]);
    dfree(h->d_angle); dfree(h->d_beam_cs); dfree(h->d_beam_csx); dfree(h->d_beam_csxg); dfree(h->d_beam_csi); dfree(h->d_beam_err); dfree(h->d_obs_idx); dfree(h->d_Lt); dfree(h->d_Ltd); dfree(h->d_items); dfree(h->d_centres); dfree(h->d_nitems); dfree(h->d_unit_sums); dfree(h->d_unit_begin); dfree(h->d_nunits); dfree(h->d_obs); dfree(h->d_free);
    if (h->h_obs) (void)hipHostFree(h->h_obs);
    for (int i = 0; i < EV_COUNT; ++i)
        if (h->ev[i]) (void)hipEventDestroy(h->ev[i]);
    if (h->ev_obs) (void)hipEventDestroy(h->ev_obs);
    if (h->ev_children) (void)hipEventDestroy(h->ev_children);
    if (h->ev_layout) (void)hipEventDestroy(h->ev_layout);
    if (h->ev_ext_in) (void)hipEventDestroy(h->ev_ext_in);
    if (h->ev_ext_out) (void)hipEventDestroy(h->ev_ext_out);
    dfree(h->d_bbox_nx); dfree(h->d_tilemap_nx); dfree(h->d_tilemark_nx);
    if (h->stream2) { (void)hipStreamSynchronize(h->stream2); (void)hipStreamDestroy(h->stream2); }
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

int mcl_set_map(mcl_engine_t *h, const int8_t *data, uint32_t width, uint32_t height, float resolution, double origin_x,
                double origin_y)
{
    if (!h) return MCL_ERR_INVALID_ARG;
    if (!data || width == 0 || height == 0 || width > 200000 || height > 200000) return fail(h, MCL_ERR_INVALID_ARG, "bad map dimensions");
    graph_reset(h);
    if (!(resolution > 0.0f)) return fail(h, MCL_ERR_INVALID_ARG, "invalid map resolution");   // cpp:236-240
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const double res = (double)resolution;                    // cpp:191
    const int P = (int)(h->cfg.max_range_m / res);            // cpp:195
    // cpp:195 has no bound; the engine's tables are (P + 1)^2 doubles and its step indices 16 bits.  Up to 243 px the windowed
    // kernels apply, up to 281 px k_rays_skip's LDS window, beyond that k_rays_skip's global-field path (same results)
    if (P < 1 || P > 2047) return fail(h, MCL_ERR_UNSUPPORTED, "MAX_RANGE_PX must be in [1, 2047]");
    h->W = (int)width; h->H = (int)height; h->P = P;
    h->res = res; h->ox = origin_x; h->oy = origin_y;
    h->Wp = h->W + 1; h->Hp = h->H + 1; h->Wps = (h->Wp + 7) & ~7;
    // LDS window: as large as 160 KiB allows (nibbles), multiple of 8 cells
    h->tw_cells = 568;
    // k_rays_quad: byte window of side S in half the LDS (S*S <= 80 KiB, S % 8 == 0); usable when the extent
    // budget S - (P+2) - 3 is at least 48 cells, otherwise k_rays_skip's full-LDS nibble window is used
    { const int S = h->env_qside > 0 ? h->env_qside : 280; h->qside = (S - (P + 2) - 3 >= 32) ? S : 0; }
    // k_rays_sweep: the 256-cell LDS windows hold ranges up to 243 px; beyond that (or with MCL_SWEEP_GLOBAL=1) the same walk
    // probes mirrored copies of the wedge fields in global memory, a position's cell dwords being absolute offsets there (no
    // bound on the range from the position format)
    h->sweep_global = h->env_sweep_global || !mcl::sweep_window_fits(P);
    h->distg_why_not = nullptr;
    const bool want_wedges = h->qside > 0 || h->sweep_global;     // wedge + quadrant fields (k_rays_far reads the latter)
    build_sensor_table(h->cfg, P, h->table);
    const int tw = P + 1;
    std::vector<float> L((size_t)tw * tw);
    const double inv_squash = 1.0 / h->cfg.squash_factor;     // cpp:53
    for (int r = 0; r < tw; ++r)
        for (int d = 0; d < tw; ++d) L[(size_t)r * tw + d] = (float)(std::log(h->table[(size_t)d * tw + r]) * inv_squash);
    std::vector<uint8_t> dist;
    build_distance_field(data, h->W, h->H, h->Wp, h->Hp, h->Wps, dist);
    h->have_map = false;                 // until every buffer below exists again: a failure leaves "map not set", never dangling pointers
    dfree(h->d_grid); dfree(h->d_dist); dfree(h->d_dist4); dfree(h->d_L); dfree(h->d_table);
    for (int q = 0; q < 4; ++q) dfree(h->d_distq[q]);
    HIPCHK(h, hipMalloc(&h->d_grid, (size_t)h->W * h->H));
    HIPCHK(h, hipMalloc(&h->d_dist, dist.size()));
    HIPCHK(h, hipMalloc(&h->d_L, L.size() * sizeof(float)));
    HIPCHK(h, hipMalloc(&h->d_table, h->table.size() * sizeof(double)));
    HIPCHK(h, hipMemcpy(h->d_grid, data, (size_t)h->W * h->H, hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(h->d_dist, dist.data(), dist.size(), hipMemcpyHostToDevice));
    {
        std::vector<uint8_t> d4(dist.size() / 2);          // Wps is a multiple of 8
        for (size_t k = 0; k < d4.size(); ++k)
            d4[k] = (uint8_t)(std::min<int>(dist[2 * k], 15) | (std::min<int>(dist[2 * k + 1], 15) << 4));
        HIPCHK(h, hipMalloc(&h->d_dist4, d4.size()));
        HIPCHK(h, hipMemcpy(h->d_dist4, d4.data(), d4.size(), hipMemcpyHostToDevice));
    }
    if (want_wedges) {
        static const int qsx[4] = {1, -1, -1, 1}, qsy[4] = {1, 1, -1, -1};
        std::vector<uint8_t> dq;
        for (int q = 0; q < 4; ++q) {
            build_directional_field(data, h->W, h->H, h->Wp, h->Hp, h->Wps, qsx[q], qsy[q], dq);
            HIPCHK(h, hipMalloc(&h->d_distq[q], dq.size()));
            HIPCHK(h, hipMemcpy(h->d_distq[q], dq.data(), dq.size(), hipMemcpyHostToDevice));
        }
    }
    dfree(h->d_distw); dfree(h->d_distg);
    if (want_wedges) {
        // wedge fields for k_rays_cell (mcl_wedge.h), built on the device from the isotropic field's stop cells
        const size_t fsz = (size_t)h->Hp * h->Wps, ncell = (size_t)h->Hp * h->Wp;
        int32_t *d_nxt = nullptr, *d_prv = nullptr;
        mcl::WedgeRow *d_rows = nullptr;
        std::vector<mcl::WedgeRow> rows(2 * mcl::kWedgeR + 1);
        HIPCHK(h, hipMalloc(&h->d_distw, fsz * mcl::kWedges));
        HIPCHK(h, hipMemsetAsync(h->d_distw, 0, fsz * mcl::kWedges, h->stream));   // same stream as the kernels that fill it
        HIPCHK(h, hipMalloc(&d_nxt, ncell * 4));
        HIPCHK(h, hipMalloc(&d_prv, ncell * 4));
        HIPCHK(h, hipMalloc(&d_rows, rows.size() * sizeof(mcl::WedgeRow)));
        hipLaunchKernelGGL(mcl::k_row_tables, dim3((h->Hp + 63) / 64), dim3(64), 0, h->stream, h->d_dist, h->Wp, h->Hp, h->Wps, d_nxt, d_prv);
        int rc_w = MCL_OK;
        for (int k = 0; k < mcl::kWedges && rc_w == MCL_OK; ++k) {
            mcl::wedge_rows(k, rows.data());
            if (hipMemcpyAsync(d_rows, rows.data(), rows.size() * sizeof(mcl::WedgeRow), hipMemcpyHostToDevice, h->stream) != hipSuccess ||
                hipStreamSynchronize(h->stream) != hipSuccess) { rc_w = MCL_ERR_HIP; break; }
            hipLaunchKernelGGL(mcl::k_wedge_field, dim3((h->Wp + 255) / 256, h->Hp), dim3(256), 0, h->stream, d_nxt, d_prv, h->Wp, h->Hp, h->Wps,
                               d_rows, h->d_distw + (size_t)k * fsz);
            if (hipStreamSynchronize(h->stream) != hipSuccess) rc_w = MCL_ERR_HIP;   // rows[] is reused by the next wedge
        }
        (void)hipFree(d_nxt); (void)hipFree(d_prv); (void)hipFree(d_rows);
        if (rc_w != MCL_OK) return fail(h, rc_w, "building the wedge fields failed");
        if (h->sweep_global) {
            // the copies k_rays_sweep<.., GLOBAL> probes: every field mirrored for its quadrant, with a two-cell ring of stop bytes
            // and a TAIL of stop rows (k_ring_field; the allocation is stop-filled first).  Memory safety does not rest on the rays
            // being valid: a walk starts in a cell of the ringed grid, runs towards +x, +y only and advances by at most P samples
            // in total whatever bytes it reads (mcl::sweep_global_layout: the largest offset it can form lies inside its own
            // field's tail; tests/test_sweep_addressing.py enumerates the corners).
            const mcl::SweepGlobalLayout gl = mcl::sweep_global_layout(h->Wp, h->Hp, P);
            if (!gl.ok) {
                h->distg_why_not = "k_rays_sweep's global-field form needs the sixteen ringed fields (with their tails) in less than 2^32 bytes and rows / pitch below 2^24: this map is too large for it";
            } else {
                h->distg_pitch = gl.pitch;
                h->distg_stride = gl.stride;
                HIPCHK(h, hipMalloc(&h->d_distg, gl.alloc));
                HIPCHK(h, hipMemsetAsync(h->d_distg, 0xFF, gl.alloc, h->stream));
                for (int k = 0; k < mcl::kWedges; ++k) {
                    const int q = k >> mcl::kWedgeShift;
                    hipLaunchKernelGGL(mcl::k_ring_field, dim3((h->distg_pitch + 255) / 256, h->Hp + 4), dim3(256), 0, h->stream, h->d_distw + (size_t)k * fsz,
                                       h->Wp, h->Hp, h->Wps, h->distg_pitch, (q == 0 || q == 3) ? 1 : 0, (q == 0 || q == 1) ? 1 : 0,
                                       h->d_distg + (size_t)k * h->distg_stride);
                }
                HIPCHK(h, hipGetLastError());
                HIPCHK(h, hipStreamSynchronize(h->stream));
            }
        }
    }
    HIPCHK(h, hipMemcpy(h->d_L, L.data(), L.size() * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(h->d_table, h->table.data(), h->table.size() * sizeof(double), hipMemcpyHostToDevice));
    h->lt_capacity = 0; dfree(h->d_Lt); h->ltd_capacity = 0; dfree(h->d_Ltd); h->ltd_ready = false;
    {   // free-space list for initialize_global (cpp:199-213, 411-421): row-major order of data == 0
        std::vector<uint32_t> fr;
        for (size_t i = 0; i < (size_t)h->W * h->H; ++i)
            if (data[i] == 0) fr.push_back((uint32_t)i);
        dfree(h->d_free);
        h->n_free = fr.size();
        if (h->n_free) {
            HIPCHK(h, hipMalloc(&h->d_free, fr.size() * sizeof(uint32_t)));
            HIPCHK(h, hipMemcpy(h->d_free, fr.data(), fr.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        }
    }
    h->have_map = true;
    return MCL_OK;
}

int mcl_get_max_range_px(const mcl_engine_t *h, int32_t *out)
{
    if (!h || !out) return MCL_ERR_INVALID_ARG;
    if (!h->have_map) return MCL_ERR_NOT_READY;
    *out = h->P;
    return MCL_OK;
}

int mcl_get_sensor_table(const mcl_engine_t *h, double *out, size_t n)
{
    if (!h || !out) return MCL_ERR_INVALID_ARG;
    if (!h->have_map) return MCL_ERR_NOT_READY;
    if (n != h->table.size()) return MCL_ERR_INVALID_ARG;
    std::memcpy(out, h->table.data(), n * sizeof(double));
    return MCL_OK;
}

int mcl_set_beam_angles(mcl_engine_t *h, const float *angles, int32_t n_beams)
{
    if (h) graph_reset(h);
    if (!h) return MCL_ERR_INVALID_ARG;
    if (!angles || n_beams <= 0 || n_beams > 65536) return fail(h, MCL_ERR_INVALID_ARG, "bad beam angles");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    h->B = 0;                            // until every buffer below exists again (ready() tests B > 0)
    h->bpad = (n_beams + 63) & ~63;
    h->quad_ok = n_beams < 16384;
    for (int j = 1; j < n_beams && h->quad_ok; ++j)
        if (!(angles[j] > angles[j - 1])) h->quad_ok = false;
    if (h->quad_ok && !((double)angles[n_beams - 1] - (double)angles[0] < 2.0 * M_PI - 1e-3)) h->quad_ok = false;
    h->angles.assign(angles, angles + n_beams);
    // padded so that k_rays_skip never clamps its beam index, with at least one entry after the last beam: k_rays_sweep's
    // beam walk requests the NEXT beam's direction on every trip, the last one included
    const int ncs = (n_beams + 1 + 255) & ~255;
    std::vector<double2> cs(ncs);
    for (int j = 0; j < ncs; ++j) {
        double a = (double)angles[j < n_beams ? j : n_beams - 1];   // cpp:533 widens the float angle
        cs[j] = make_double2(std::cos(a), std::sin(a));
    }
    // k_rays_sweep pads the lanes whose scan begins or ends inside a wedge with virtual beams -- the angular grid of the scan
    // continued beyond its ends -- up to the beams of a full wedge.  Only for an evenly spaced scan (every angle within a quarter
    // of the spacing of the grid through the first and the last) that leaves a wedge of the turn uncovered (no second range).
    h->beam_pad = 0; h->beam_margin = 0;
    const double span = n_beams > 1 ? (double)angles[n_beams - 1] - (double)angles[0] : 0.0;
    const double inc = n_beams > 1 ? span / (double)(n_beams - 1) : 0.0;
    const double wedge = 2.0 * M_PI / (double)mcl::kWedges;
    if (h->quad_ok && inc > 0.0 && span + wedge < 2.0 * M_PI - 1e-3 && !getenv("MCL_NO_BEAM_PAD")) {
        bool regular = true;
        for (int j = 0; j < n_beams && regular; ++j)
            regular = std::fabs((double)angles[j] - ((double)angles[0] + (double)j * inc)) < 0.25 * inc;
        const double per_wedge = wedge / inc;
        if (regular && per_wedge >= 2.0 && per_wedge <= 120.0) {
            h->beam_pad = (int)std::ceil(per_wedge - 1e-9);          // (beam_pad - 1) spacings are less than a wedge
            h->beam_margin = (h->beam_pad + 2 + 7) & ~7;
        }
    }
    const int ncsx = n_beams + 2 * h->beam_margin + 8;
    // The LDS form continues the scan's angular grid beyond its ends (a virtual ray then looks like its neighbours' real ones).
    // Such a ray may leave the lane's wedge by a beam or so, where the wedge's skip field does not hold for it: harmless in an
    // LDS window (it can only read the window), NOT in the global-field form, where a ray that jumps a wall near the map border
    // would leave the field -- there a virtual beam repeats the first / last real beam (inside the wedge by construction).
    std::vector<double2> csx(ncsx), csxg(ncsx);
    for (int i = 0; i < ncsx; ++i) {
        const int j = i - h->beam_margin;
        const double ac = (double)angles[j < 0 ? 0 : (j < n_beams ? j : n_beams - 1)];
        double a;
        if (j >= 0 && j < n_beams) a = (double)angles[j];
        else if (h->beam_margin > 0) a = (double)angles[0] + (double)j * inc;
        else a = ac;
        csx[i] = make_double2(std::cos(a), std::sin(a));
        csxg[i] = make_double2(std::cos(ac), std::sin(ac));
    }
    dfree(h->d_angle); dfree(h->d_beam_cs); dfree(h->d_beam_csx); dfree(h->d_beam_csxg); dfree(h->d_obs_idx); dfree(h->d_obs);
    dfree(h->d_beam_csi); dfree(h->d_beam_err);
    h->rec_ok = false;
    if (h->beam_margin > 0 && !getenv("MCL_SWEEP_NO_REC")) {
        // REC: the walk turns the direction by the grid increment instead of fetching it (mcl_rays_sweep.h).  Every beam must lie
        // within 2e-6 rad of the grid through the first and the last (the second-order term of its offset, e^2 / 2 * 2^32 units,
        // then stays below 1e-2 unit of the guard's budget): the float rounding of a0 + j inc, which is what a lidar driver
        // publishes, is a few 1e-8.
        const int ecols = (n_beams + 2 * h->beam_margin + 64) & ~63;          // = ltd_cols (ensure_lt)
        std::vector<double2> csi(ncsx);
        std::vector<double> err((size_t)ecols, 0.0);
        double worst = 0.0;
        for (int i = 0; i < ncsx; ++i) {
            const int j = i - h->beam_margin;
            const double ag = (double)angles[0] + (double)j * inc;
            csi[i] = make_double2(std::cos(ag), std::sin(ag));
            if (j >= 0 && j < n_beams && i < ecols) { err[i] = (double)angles[j] - ag; worst = std::max(worst, std::fabs(err[i])); }
        }
        if (worst <= 2e-6 && ncsx <= ecols + 8) {
            HIPCHK(h, hipMalloc(&h->d_beam_csi, (size_t)ncsx * sizeof(double2)));
            HIPCHK(h, hipMemcpy(h->d_beam_csi, csi.data(), (size_t)ncsx * sizeof(double2), hipMemcpyHostToDevice));
            HIPCHK(h, hipMalloc(&h->d_beam_err, (size_t)ecols * sizeof(double)));
            HIPCHK(h, hipMemcpy(h->d_beam_err, err.data(), (size_t)ecols * sizeof(double), hipMemcpyHostToDevice));
            h->rec_c = std::cos(inc); h->rec_s = std::sin(inc);
            h->rec_ok = true;
        }
    }
    if (h->h_obs) { (void)hipHostFree(h->h_obs); h->h_obs = nullptr; }
    HIPCHK(h, hipMalloc(&h->d_obs, (size_t)n_beams * sizeof(float)));
    HIPCHK(h, hipHostMalloc(&h->h_obs, (size_t)n_beams * sizeof(float)));
    HIPCHK(h, hipMalloc(&h->d_angle, (size_t)n_beams * sizeof(float)));
    HIPCHK(h, hipMalloc(&h->d_beam_cs, (size_t)ncs * sizeof(double2)));
    HIPCHK(h, hipMalloc(&h->d_obs_idx, (size_t)n_beams * sizeof(int32_t)));
    HIPCHK(h, hipMemcpy(h->d_angle, angles, (size_t)n_beams * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(h->d_beam_cs, cs.data(), (size_t)ncs * sizeof(double2), hipMemcpyHostToDevice));
    HIPCHK(h, hipMalloc(&h->d_beam_csx, (size_t)ncsx * sizeof(double2)));
    HIPCHK(h, hipMemcpy(h->d_beam_csx, csx.data(), (size_t)ncsx * sizeof(double2), hipMemcpyHostToDevice));
    HIPCHK(h, hipMalloc(&h->d_beam_csxg, (size_t)ncsx * sizeof(double2)));
    HIPCHK(h, hipMemcpy(h->d_beam_csxg, csxg.data(), (size_t)ncsx * sizeof(double2), hipMemcpyHostToDevice));
    h->lt_capacity = 0; dfree(h->d_Lt); h->ltd_capacity = 0; dfree(h->d_Ltd); h->ltd_ready = false;
    h->B = n_beams;
    return MCL_OK;
}

// weight_scale: the weight the fixed-point values are scaled by (null: this call's own maximum).  A shard of a larger set
// must use the maximum over the WHOLE set, or the shards' fixed-point weights are not comparable.
static int set_particles_impl(mcl_engine_t *h, const double *xyz, const double *weights, int64_t n, const double *weight_scale)
{
    if (h) graph_reset(h);
    if (!h) return MCL_ERR_INVALID_ARG;
    if (!xyz || !weights || n <= 0 || n > h->cap) return fail(h, MCL_ERR_INVALID_ARG, "bad particle arrays / count");
    if (weight_scale && !(*weight_scale > 0.0)) return fail(h, MCL_ERR_INVALID_ARG, "weight scale must be positive");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const size_t nb = (size_t)n * sizeof(double);
    const int c = h->cur;
    HIPCHK(h, hipMemcpyAsync(h->d_x[c], xyz, nb, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_y[c], xyz + n, nb, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_th[c], xyz + 2 * n, nb, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_w, weights, nb, hipMemcpyHostToDevice, h->stream));
    h->N = n;
    if (weight_scale) {
        std::memcpy(&h->h_result[kResultStage], weight_scale, sizeof(double));        // pinned: stays valid until the copy has run
        HIPCHK(h, hipMemcpyAsync(h->d_scalars, &h->h_result[kResultStage], sizeof(double), hipMemcpyHostToDevice, h->stream));
    }
    int rc = weight_stats(h, false, weight_scale ? h->d_scalars : nullptr, true);
    if (rc) return rc;
    rc = scan_weights(h, h->d_q, h->d_cdf, n, 0, nullptr);
    if (rc) return rc;
    rc = fetch_scalars(h);
    if (rc) return rc;
    h->have_particles = true;
    h->far_fresh = true;
    h->pack_valid[0] = h->pack_valid[1] = false;
    h->have_idx = h->have_steps = h->have_logw = false;
    comm_forget(h->comm);                   // a sharded set: the other shards' lists are unknown again
    h->stage_kept = false;
    return MCL_OK;
}

int mcl_set_particles(mcl_engine_t *h, const double *xyz, const double *weights, int64_t n)
{
    return set_particles_impl(h, xyz, weights, n, nullptr);
}

int mcl_set_particles_shard(mcl_engine_t *h, const double *xyz, const double *weights, int64_t n, double max_weight_of_the_whole_set)
{
    return set_particles_impl(h, xyz, weights, n, &max_weight_of_the_whole_set);
}

static int finish_init(mcl_engine *h, int64_t n, int64_t n_total)
{
    graph_reset(h);
    h->N = n;
    hipLaunchKernelGGL(mcl::k_fill, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, h->d_w, n, 1.0 / (double)n_total);
    HIPCHK(h, hipGetLastError());
    int rc = weight_stats(h, false, nullptr, true);
    if (rc) return rc;
    rc = scan_weights(h, h->d_q, h->d_cdf, n, 0, nullptr);
    if (rc) return rc;
    rc = fetch_scalars(h);
    if (rc) return rc;
    h->have_particles = true;
    h->far_fresh = true;
    h->pack_valid[0] = h->pack_valid[1] = false;
    h->have_idx = h->have_steps = h->have_logw = false;
    comm_forget(h->comm);                   // a sharded set: the other shards' lists are unknown again
    h->stage_kept = false;
    h->init_idx++;
    return MCL_OK;
}

int mcl_init_particles_pose(mcl_engine_t *h, const double pose[3], int64_t n, int64_t first_global_index, int64_t n_total)
{
    if (!h) return MCL_ERR_INVALID_ARG;
    if (!pose || n <= 0 || n > h->cap || first_global_index < 0 || n_total < n || n_total >= MCL_MAX_TOTAL_PARTICLES)
        return fail(h, MCL_ERR_INVALID_ARG, "bad init arguments (the sharded total must stay below 2^27)");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const int c = h->cur;
    hipLaunchKernelGGL(mcl::k_init_pose, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, pose[0], pose[1], pose[2], n,
                       first_global_index, (uint32_t)h->cfg.seed, (uint32_t)(h->cfg.seed >> 32), h->init_idx, h->d_x[c], h->d_y[c], h->d_th[c]);
    HIPCHK(h, hipGetLastError());
    return finish_init(h, n, n_total);
}

int mcl_init_global(mcl_engine_t *h, int64_t n, int64_t first_global_index, int64_t n_total)
{
    if (!h) return MCL_ERR_INVALID_ARG;
    if (!h->have_map) return fail(h, MCL_ERR_NOT_READY, "map not set");                 // cpp:403
    if (h->n_free == 0) return fail(h, MCL_ERR_NOT_READY, "No free space found in map!");   // cpp:423-427
    if (n <= 0 || n > h->cap || first_global_index < 0 || n_total < n || n_total >= MCL_MAX_TOTAL_PARTICLES)
        return fail(h, MCL_ERR_INVALID_ARG, "bad init arguments (the sharded total must stay below 2^27)");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const int c = h->cur;
    hipLaunchKernelGGL(mcl::k_init_global, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, h->d_free, h->n_free, h->W, h->res,
                       h->ox, h->oy, n, first_global_index, (uint32_t)h->cfg.seed, (uint32_t)(h->cfg.seed >> 32), h->init_idx, h->d_x[c],
                       h->d_y[c], h->d_th[c]);
    HIPCHK(h, hipGetLastError());
    return finish_init(h, n, n_total);
}

int mcl_get_particles(mcl_engine_t *h, double *xyz, int64_t n)
{
    if (!h || !xyz) return MCL_ERR_INVALID_ARG;
    if (!h->have_particles) return MCL_ERR_NOT_READY;
    if (n != h->N) return fail(h, MCL_ERR_INVALID_ARG, "n != active particle count");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const size_t nb = (size_t)n * sizeof(double);
    const int c = h->cur;
    HIPCHK(h, hipMemcpyAsync(xyz, h->d_x[c], nb, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(xyz + n, h->d_y[c], nb, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(xyz + 2 * n, h->d_th[c], nb, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return MCL_OK;
}

int mcl_get_weights(mcl_engine_t *h, double *weights, int64_t n)
{
    if (!h || !weights) return MCL_ERR_INVALID_ARG;
    if (!h->have_particles) return MCL_ERR_NOT_READY;
    if (n != h->N) return fail(h, MCL_ERR_INVALID_ARG, "n != active particle count");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    hipLaunchKernelGGL(mcl::k_normalized, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, h->d_w, n, h->global_sums[0], h->d_tmp);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(weights, h->d_tmp, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return MCL_OK;
}

int mcl_sample_particles(mcl_engine_t *h, int32_t k, const double *uniforms, double *out)
{
    if (!h || !out || k <= 0 || k > 65536) return MCL_ERR_INVALID_ARG;
    if (!h->have_particles) return MCL_ERR_NOT_READY;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    double *d_u = nullptr;
    double *d_out = h->d_tmp;
    if ((int64_t)k * 4 > h->cap * 3) return fail(h, MCL_ERR_INVALID_ARG, "k too large for this engine");
    if (uniforms) {
        d_u = h->d_tmp + (size_t)3 * k;
        HIPCHK(h, hipMemcpyAsync(d_u, uniforms, (size_t)k * sizeof(double), hipMemcpyHostToDevice, h->stream));
    }
    const int c = h->cur;
    hipLaunchKernelGGL(mcl::k_sample, dim3((k + 63) / 64), dim3(64), 0, h->stream, h->d_x[c], h->d_y[c], h->d_th[c], h->d_cdf, h->N,
                       h->q_total, d_u, (uint32_t)h->cfg.seed, (uint32_t)(h->cfg.seed >> 32), h->update_idx, k, d_out);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(out, d_out, (size_t)3 * k * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return MCL_OK;
}

int mcl_particle_mean(mcl_engine_t *h, double out[3])
{
    if (!h || !out) return MCL_ERR_INVALID_ARG;
    if (!h->have_particles) return MCL_ERR_NOT_READY;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const int c = h->cur;
    const int nb = 256;
    hipLaunchKernelGGL(mcl::k_colsum, dim3(nb), dim3(mcl::kRedThreads), 0, h->stream, h->d_x[c], h->d_y[c], h->d_th[c], h->N, h->d_part);
    HIPCHK(h, hipGetLastError());
    std::vector<double> part((size_t)nb * 4);
    HIPCHK(h, hipMemcpyAsync(part.data(), h->d_part, part.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    double s[3] = {0, 0, 0};
    for (int b = 0; b < nb; ++b) { s[0] += part[b * 4]; s[1] += part[b * 4 + 1]; s[2] += part[b * 4 + 2]; }
    for (int k = 0; k < 3; ++k) out[k] = s[k] / (double)h->N;
    return MCL_OK;
}


// What the resampling kernel does for the ray stage that follows it in the same update (mcl_update and the staged flow alike):
// the per-particle constants come out of it (k_rays_skip: one launch less per small update; k_rays_cell / k_rays_sweep: also the
// per-particle scratch their stage wants zeroed, a pass over the children less), and from the second update of a configuration on
// the (key, index) pairs of the ordering and the few words launch_rays would clear.
static void resample_ray_extras(mcl_engine *h, int64_t n, mcl::ResampleArgs &a)
{
    const int rmode = choose_ray_mode(h, n, false);
    if (!(rmode == 2 || rmode >= 4)) return;
    a.pc_out = h->d_pc; a.ox = h->ox; a.oy = h->oy; a.res = h->res;
    if (rmode >= 4) { a.clr_logw_acc = h->d_logw_acc; a.clr_far_flags = reinterpret_cast<uint32_t *>(h->d_far); }
    h->pc_ready = true;
    // The ordering of the ray stage works from the layout (bounding box, occupied tiles) of the PREVIOUS update's children
    // when there is one: the set moves by a cell or so per update and the order only decides which rays share a wave.
    // Radix ordering: this kernel then writes the (key, index) pairs too and the sort starts right after it.
    const int ntx_abs = ((h->Wp * mcl::kSortSub - 1) >> 5) + 1, nty_abs = ((h->Hp * mcl::kSortSub - 1) >> 5) + 1;
    const bool tiles_ok = (int64_t)ntx_abs * nty_abs <= mcl::kSortMaxTiles;
    if (rmode >= 4 && h->layout_valid && h->layout_n == n && !h->env_no_stale_layout) {
        h->layout_stale_used = true;
        const bool radix = h->env_sort_radix >= 0 ? h->env_sort_radix != 0 : n >= 3000000;
        if (radix && h->d_skey2) {
            a.key_out = h->d_skey; a.val_out = h->d_srank; a.key_bbox = h->d_bbox; a.key_tilemap = tiles_ok ? h->d_tilemap : nullptr;
            a.key_ntx = ntx_abs; a.key_Wp = h->Wp; a.key_Hp = h->Hp;
            h->keys_done = true;
        }
    }
    if (rmode >= 4 && h->prep_cache_valid && h->prep_cache_n == n && !h->env_no_prep_fold) {
        a.prep = h->prep_cache; a.prep_on = 1;      // launch_rays checks that this is what it would have cleared
        if (h->layout_stale_used) a.prep.bbox = nullptr;     // the layout in d_bbox is in use: not reset
        h->prep_passed = a.prep;
        h->prep_folded = true;
    }
}

// After the resampling kernel: the layout of THESE children, for the next update, on the second stream beside the sort and the ray stage.
// (in two steps: the event right behind the resampling kernel, the launches once the ray stage has been submitted -- the
//  ordering kernels of the main stream are on the critical path of a small update, these are not)
static int layout_mark(mcl_engine *h, int64_t n)
{
    h->layout_wanted = false;
    if (choose_ray_mode(h, n, false) < 4 || h->env_no_stale_layout) return MCL_OK;
    HIPCHK(h, hipEventRecord(h->ev_children, h->stream));
    h->layout_wanted = true;
    // (launched right here the layout kernels share the device with the sort: measured 0.04-0.06 ms per update slower at 1M and
    //  4M particles than behind the ray stage, where they run in the shadow of the persistent kernel's tail)
    return MCL_OK;
}

static int next_layout_launch(mcl_engine *h, int64_t n)
{
    if (!h->layout_wanted) return MCL_OK;
    h->layout_wanted = false;
    const int ntx_abs = ((h->Wp * mcl::kSortSub - 1) >> 5) + 1, nty_abs = ((h->Hp * mcl::kSortSub - 1) >> 5) + 1;
    const bool tiles_ok = (int64_t)ntx_abs * nty_abs <= mcl::kSortMaxTiles;
    const int bstride = n >= (1 << 20) ? 16 : 1;
    const int play = h->env_no_bucket_cuts ? -1 : (choose_ray_mode(h, n, false) == 5 ? sweep_play(h) : 0);
    HIPCHK(h, hipStreamWaitEvent(h->stream2, h->ev_children, 0));
    hipLaunchKernelGGL(mcl::k_bbox_init, dim3(1), dim3(64), 0, h->stream2, h->d_bbox_nx, play);
    hipLaunchKernelGGL(mcl::k_cell_bbox, dim3((unsigned)std::min<int64_t>((n / bstride + 255) / 256, 128)), dim3(256), 0, h->stream2, h->d_pc, n,
                       bstride, h->Wp, h->Hp, h->d_bbox_nx, tiles_ok ? h->d_tilemark_nx : (int *)nullptr, ntx_abs);
    if (tiles_ok)
        hipLaunchKernelGGL(mcl::k_tile_compact, dim3(1), dim3(1024), 0, h->stream2, h->d_bbox_nx, h->d_tilemark_nx, h->d_tilemap_nx, ntx_abs * nty_abs);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipEventRecord(h->ev_layout, h->stream2));
    h->layout_pending = true;
    return MCL_OK;
}

// End of an update: the layout made beside its ray stage becomes the one the next update orders by.
static int layout_adopt(mcl_engine *h, int64_t n)
{
    if (!h->layout_pending) return MCL_OK;
    HIPCHK(h, hipEventSynchronize(h->ev_layout));
    std::swap(h->d_bbox, h->d_bbox_nx); std::swap(h->d_tilemap, h->d_tilemap_nx);
    h->layout_valid = true; h->layout_n = n; h->layout_pending = false;
    return MCL_OK;
}

static int do_update(mcl_engine_t *h, const double action[3], const float *obs, int32_t n_beams, const double *normals,
                     const double *uniforms, bool resample_and_move, int obs_stride = 1)
{
    if (!h) return MCL_ERR_INVALID_ARG;
    if (!ready(h, true)) return fail(h, MCL_ERR_NOT_READY, "map, beam angles and particles must be set first");
    if (!obs || n_beams != h->B || (resample_and_move && !action)) return fail(h, MCL_ERR_INVALID_ARG, "bad action/observation");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    auto t0 = std::chrono::steady_clock::now();
    const int64_t n = h->N;
    const double *d_norm = nullptr, *d_uni = nullptr;
    if (resample_and_move) {
        if (normals) {
            HIPCHK(h, hipMemcpyAsync(h->d_inject, normals, (size_t)n * 3 * sizeof(double), hipMemcpyHostToDevice, h->stream));
            d_norm = h->d_inject;
        }
        if (uniforms) {
            HIPCHK(h, hipMemcpyAsync(h->d_inject + (size_t)3 * h->cap, uniforms, (size_t)n * sizeof(double), hipMemcpyHostToDevice, h->stream));
            d_uni = h->d_inject + (size_t)3 * h->cap;
        }
    }
    // adaptive resampling (off by default): keep the particles when the previous update left an effective sample
    // size (sum w)^2 / sum w^2 of at least r * N; their weights then multiply, i.e. the log-weights add
    bool keep = false;
    if (resample_and_move && h->cfg.resample_neff_permille > 0 && h->carry_valid && !uniforms) {
        const double sw = h->h_scalars[1], sww = h->h_scalars[7];
        keep = sww > 0.0 && sw * sw >= ((double)h->cfg.resample_neff_permille / 1000.0) * (double)n * sww;
    }
    if (h->cfg.resample_neff_permille > 0 && h->cfg.weight_mode == MCL_WEIGHT_PRODUCT)
        return fail(h, MCL_ERR_UNSUPPORTED, "resample_neff_permille needs weight_mode LOG");
    if (!resample_and_move) HIPCHK(h, hipMemsetAsync(h->d_counters, 0, 4 * sizeof(unsigned long long), h->stream));   // else: the resampling kernel
    h->pc_ready = false;
    h->layout_stale_used = false; h->keys_done = false;      // (set below when this update orders by the previous update's layout)
    // A small update (k_rays_skip, the whole tail in one workgroup) is three launches and no copy: resampling + motion +
    // per-particle constants + table rows of the scan | rays against the static table | weights, sums, CDF and the result
    // block written straight to pinned host memory.  Every buffer exists once a regular update has run (graph_warm).
    const bool tiny = resample_and_move && h->cfg.graph_mode != 1 && h->graph_warm && n <= mcl::kTinyTailMax && !keep &&
                      choose_ray_mode(h, n, false) == 2 && h->cfg.weight_mode == MCL_WEIGHT_LOG && h->cfg.resample_neff_permille == 0;
    if (!tiny) HIPCHK(h, hipEventRecord(h->ev[EV_START], h->stream));
    bool obs_early = false;
    if (resample_and_move) {
        const int c = h->cur, nx = c ^ 1;
        mcl::ResampleArgs a{};
        a.px = h->d_x[c]; a.py = h->d_y[c]; a.pth = h->d_th[c];
        a.cdf = h->d_cdf; a.n_parents = n; a.q_total = h->q_total;
        a.tile_excl = (h->blocktot_for == h->d_cdf && h->blocktot_n == n) ? h->d_blocktot : nullptr;   // spine of the scan that produced d_cdf
        a.leaders = a.tile_excl ? h->d_leaders : nullptr;
        // parents: the compact list the last scan left (the particles that carry weight: a few per cent after an update
        // with many beams) when it exists, else the full CDF and the packed records
        const bool compact = !keep && h->compact_n > 0;
        if (compact) {
            a.ccdf = h->d_ccdf; a.ctop = h->d_ctop; a.n_compact = h->compact_n; a.cidx = h->d_cidx; a.crec = h->d_crec;
            a.cpack = nullptr;               // the next update most likely draws from a compact list again: no record per child
        } else {
            if (!h->pack_valid[c] && n > 65536 && !keep) {       // records first: one fetch per gathered parent instead of three
                hipLaunchKernelGGL(mcl::k_pack_records, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, h->d_x[c], h->d_y[c], h->d_th[c], n, h->d_pack[c]);
                h->pack_valid[c] = true;
            }
            a.ppack = h->pack_valid[c] ? h->d_pack[c] : nullptr;
            a.cpack = h->d_pack[nx];
        }
        a.cx = h->d_x[nx]; a.cy = h->d_y[nx]; a.cth = h->d_th[nx];
        a.idx_out = h->d_idx;
        a.n_children = n; a.child_first = 0; a.n_children_total = n;
        a.mode = h->cfg.resample_mode;
        a.uniforms = d_uni; a.normals = d_norm;
        a.seed_lo = (uint32_t)h->cfg.seed; a.seed_hi = (uint32_t)(h->cfg.seed >> 32);
        a.update_idx = h->update_idx;
        a.k0 = 0;
        if (a.mode == MCL_RESAMPLE_SYSTEMATIC) {
            // one 32-bit offset per update: Philox stream 3 (host restatement of the same function)
            uint32_t c0 = 0, c1 = h->update_idx, c2 = 3, c3 = 0, k0 = a.seed_lo, k1 = a.seed_hi;
            for (int r = 0; r < 10; ++r) {
                uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
                uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
                c0 = n0; c1 = n1; c2 = n2; c3 = n3; k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
            }
            a.k0 = c0;
            if (uniforms) a.k0 = (uint32_t)(std::min(std::max(uniforms[0], 0.0), 0.99999999976716936) * 4294967296.0);
        }
        motion_scalars(action, a.dt, a.v, a.w);
        a.disp_x = h->cfg.motion_dispersion_x; a.disp_y = h->cfg.motion_dispersion_y; a.disp_th = h->cfg.motion_dispersion_theta;
        a.do_resample = keep ? 0 : 1; a.do_motion = 1;
        a.clear_counters = h->d_counters;
        if (tiny) {
            // the scan goes from the pinned staging buffer to table rows inside this kernel (no copy node, no table build)
            stage_observation(h, obs, obs_stride);
            a.obs_src = h->h_obs; a.obs_idx_out = h->d_obs_idx; a.obs_B = h->B; a.obs_P = h->P; a.res = h->res;
        }
        resample_ray_extras(h, n, a);
        size_t cdf_lds = 0;
        if (!a.tile_excl && a.do_resample && n <= mcl::kTinyTailMax) { a.cdf_lds_entries = (int)n; cdf_lds = (size_t)n * sizeof(uint64_t); }
        h->ev_resample_bound = false;
        if (!tiny && !h->capturing) {
            hipExtLaunchKernelGGL(mcl::k_resample_motion, dim3((unsigned)((n + 255) / 256)), dim3(256), cdf_lds, h->stream, nullptr, h->ev[EV_RESAMPLE], 0, a);
            h->ev_resample_bound = true;
        } else {
            hipLaunchKernelGGL(mcl::k_resample_motion, dim3((unsigned)((n + 255) / 256)), dim3(256), cdf_lds, h->stream, a);
        }
        HIPCHK(h, hipGetLastError());
        if (a.pc_out) { const int rc_l = layout_mark(h, n); if (rc_l) return rc_l; }
        // The tables of this update's scan (table rows of the observed ranges, Lt, Ltd) depend on nothing the resampling and ordering
        // kernels produce: with a windowed ray kernel they are built on a second stream beside those, and the ray stage waits for
        // them (a copy and two or three small launches off the critical path of an update: ~15 us).  Enqueued AFTER the resampling
        // kernel: that one is on the critical path, and a small update is bound by the order the host submits in.
        if (!tiny && !h->env_no_obs_overlap && choose_ray_mode(h, n, false) >= 3) {
            std::swap(h->stream, h->stream2);
            const int rc_obs = prepare_observation(h, obs, obs_stride);
            hipError_t ee = rc_obs ? hipSuccess : hipEventRecord(h->ev_obs, h->stream);
            std::swap(h->stream, h->stream2);
            if (rc_obs) return rc_obs;
            HIPCHK(h, ee);
            obs_early = true;
        }
        h->cur = nx;                       // cpp:689 as a pointer swap
        h->resampled_last = !keep;
        h->pack_valid[nx] = a.cpack != nullptr;
        h->compact_used = compact;
        h->have_idx = true;
    }
    int rc;
    if (tiny) {
        rc = launch_rays(h, h->d_x[h->cur], h->d_y[h->cur], h->d_th[h->cur], n, false, true);
        if (!rc) rc = weights_and_cdf(h, true);
        if (rc) return rc;
        // The result block lands in pinned memory stamped with this update's sequence number: spin on the stamp instead of
        // the stream's completion signal (the signal's path through the runtime costs several microseconds at this size).
        // Everything later on this engine is ordered behind the kernels by the stream; a stamp that never comes (a faulted
        // kernel) ends in the ordinary synchronisation, which reports the error.
        bool seen = false;
        if (h->env_tiny_poll) {
            const volatile unsigned long long *stamp = h->h_result + kResultStamp;
            const auto give_up = t0 + std::chrono::milliseconds(20);
            for (unsigned spin = 0; !seen; ++spin) {
                seen = __atomic_load_n(stamp, __ATOMIC_ACQUIRE) == h->result_seq;
                if (!seen && (spin & 1023u) == 1023u && std::chrono::steady_clock::now() > give_up) break;
            }
        }
        if (!seen) HIPCHK(h, hipStreamSynchronize(h->stream));
        unpack_result(h);
        h->have_logw = true;
        h->have_steps = h->cfg.keep_ray_steps != 0;
        h->update_idx++;
        // one unit, reported as the ray-cast stage (resampling, query prep and the tail are inside it), so that the six
        // stages still add up to the total the host uses for delay compensation
        h->timings[0] = 0.0; h->timings[1] = 0.0; h->timings[2] = 0.0; h->timings[4] = 0.0;
        h->timings[5] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        h->timings[3] = h->timings[5];
        h->ray_ms = h->timings[3];
        h->ray_ms_is_graph_tail = true;
        return MCL_OK;
    }
    if (!(resample_and_move && h->ev_resample_bound)) HIPCHK(h, hipEventRecord(h->ev[EV_RESAMPLE], h->stream));
    // Small updates are launch-bound (about twenty launches for ~0.06 ms of kernels): once a regular update has run with
    // these sizes on the k_rays_skip path, everything after the resampling kernel is replayed as one hipGraph per
    // particle buffer (observation upload, table build, rays, weights, CDF, result read-back: all arguments are fixed).
    // Eligibility is a pure function of the configuration and the sizes (choose_ray_mode), never of what the previous
    // update happened to run: k_rays_skip chosen outright has no work lists, no allocation and no fallback.
    bool graph_ok = h->cfg.graph_mode != 1 && h->graph_warm && resample_and_move && choose_ray_mode(h, n, false) == 2 && !keep && !h->cfg.debug_count_probes &&
                    h->cfg.weight_mode == MCL_WEIGHT_LOG && h->cfg.resample_neff_permille == 0;
    if (graph_ok) {
        stage_observation(h, obs, obs_stride);
        const int gi = h->cur;
        if (!h->graph_exec[gi]) {
            // nothing executes during capture, so a failure here simply falls back to the launch-by-launch path below
            hipGraph_t graph = nullptr;
            if (hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
                graph_ok = false;
            } else {
                h->capturing = true;
                rc = upload_observation(h);
                if (!rc) rc = launch_rays(h, h->d_x[gi], h->d_y[gi], h->d_th[gi], n);
                if (!rc) rc = weights_and_cdf(h);
                hipError_t ce = hipSuccess;
                if (!rc) ce = hipMemcpyAsync(h->h_result, h->d_result, kResultWords * 8, hipMemcpyDeviceToHost, h->stream);
                h->capturing = false;
                const hipError_t ee = hipStreamEndCapture(h->stream, &graph);
                hipError_t ie = hipErrorUnknown;
                if (!rc && ce == hipSuccess && ee == hipSuccess && graph && h->last_mode == 2)
                    ie = hipGraphInstantiate(&h->graph_exec[gi], graph, nullptr, nullptr, 0);
                if (graph) (void)hipGraphDestroy(graph);
                if (ie != hipSuccess) {
                    h->graph_exec[gi] = nullptr;
                    graph_reset(h);
                    (void)hipGetLastError();
                    h->cfg.graph_mode = 1;          // do not try again on this engine
                    graph_ok = false;
                }
            }
        }
    }
    if (graph_ok) {
        const int gi = h->cur;
        HIPCHK(h, hipGraphLaunch(h->graph_exec[gi], h->stream));
        h->pc_ready = false;
        HIPCHK(h, hipEventRecord(h->ev[EV_SENSOR], h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        h->compact_pending = h->d_ccdf != nullptr && !h->env_no_compact && n > mcl::kTinyTailMax;   // the captured scan wrote a list
        unpack_result(h);
        h->carry_pending = false;
        h->have_logw = true;
        h->have_steps = h->cfg.keep_ray_steps != 0;
        if (resample_and_move) h->update_idx++;
        // the captured tail is one unit: its time is reported as the ray-cast stage (query prep and table evaluation
        // are inside it), so that the six stages still add up to the total the host uses for delay compensation
        h->timings[0] = elapsed(h->ev[EV_START], h->ev[EV_RESAMPLE]);
        h->timings[1] = 0.0; h->timings[2] = 0.0; h->timings[4] = 0.0;
        h->timings[3] = elapsed(h->ev[EV_RESAMPLE], h->ev[EV_SENSOR]);      // the graph as a whole
        h->timings[5] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        h->ray_ms = h->timings[3];
        h->ray_ms_is_graph_tail = true;
        return MCL_OK;
    }
    h->ray_ms_is_graph_tail = false;
    if (obs_early) {
        HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_obs, 0));
    } else {
        rc = prepare_observation(h, obs, obs_stride);
        if (rc) return rc;
    }
    // (the tables were made beside the ordering: no query-preparation stage on this stream, nothing to time)
    h->ev_query_skipped = obs_early;
    if (!obs_early) HIPCHK(h, hipEventRecord(h->ev[EV_QUERY], h->stream));
    h->ev_rays_bound = false;
    rc = launch_rays(h, h->d_x[h->cur], h->d_y[h->cur], h->d_th[h->cur], n);
    if (rc) return rc;
    rc = next_layout_launch(h, n);          // (second stream; behind the ray stage in submission order, beside it on the device)
    if (rc) return rc;
    if (keep) {
        hipLaunchKernelGGL(mcl::k_add_carry, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, h->d_logw, h->d_carry[h->carry_idx], n);
        h->max_partials_ready = false;
    }
    if (keep || !h->ev_rays_bound) HIPCHK(h, hipEventRecord(h->ev[EV_RAYS], h->stream));
    h->bind_sensor_event = true; h->ev_sensor_bound = false;
    rc = weights_and_cdf(h);
    h->bind_sensor_event = false;
    if (rc) return rc;
    if (!h->ev_sensor_bound) HIPCHK(h, hipEventRecord(h->ev[EV_SENSOR], h->stream));
    rc = fetch_scalars(h);                 // one D2H copy (scalars, counters, overflow flag); synchronises the stream
    if (rc) return rc;
    if (h->last_quad && h->h_fix_count != 0) {
        // more undecided rays than the work list holds (only with debug_force_exact at large sizes or a
        // pathological map): redo the ray stage with the self-contained k_rays_skip
        HIPCHK(h, hipMemsetAsync(h->d_counters, 0, 4 * sizeof(unsigned long long), h->stream));
        rc = launch_rays(h, h->d_x[h->cur], h->d_y[h->cur], h->d_th[h->cur], n, true);
        if (rc) return rc;
        if (keep) {
            hipLaunchKernelGGL(mcl::k_add_carry, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, h->d_logw, h->d_carry[h->carry_idx], n);
            h->max_partials_ready = false;
        }
        rc = weights_and_cdf(h);
        if (rc) return rc;
        rc = fetch_scalars(h);
        if (rc) return rc;
    }
    if (h->carry_pending) { h->carry_idx ^= 1; h->carry_valid = true; h->carry_pending = false; }   // this update's logw - max
    { const int rc_l = layout_adopt(h, n); if (rc_l) return rc_l; }
    h->graph_warm = true;                  // every buffer this configuration needs exists now
    h->have_logw = true;
    h->have_steps = h->cfg.keep_ray_steps != 0;
    if (resample_and_move) h->update_idx++;
    h->timings[0] = elapsed(h->ev[EV_START], h->ev[EV_RESAMPLE]);
    h->timings[1] = 0.0;                   // motion is fused into the resample/gather kernel
    h->timings[2] = h->ev_query_skipped ? 0.0 : elapsed(h->ev[EV_RESAMPLE], h->ev[EV_QUERY]);
    h->timings[3] = elapsed(h->ev[h->ev_query_skipped ? EV_RESAMPLE : EV_QUERY], h->ev[EV_RAYS]);
    h->timings[4] = elapsed(h->ev[EV_RAYS], h->ev[EV_SENSOR]);
    h->timings[5] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    h->ray_ms = elapsed(h->ev[EV_K0], h->ev[EV_K1]);
    return MCL_OK;
}

int mcl_update(mcl_engine_t *h, const double action[3], const float *obs, int32_t n_beams, const double *normals_nx3,
               const double *uniforms_n)
{
    return do_update(h, action, obs, n_beams, normals_nx3, uniforms_n, true);
}

int mcl_sensor_update(mcl_engine_t *h, const float *obs, int32_t n_beams)
{
    return do_update(h, nullptr, obs, n_beams, nullptr, nullptr, false);
}

int mcl_update_scan(mcl_engine_t *h, const double action[3], const float *ranges, int32_t n_ranges, int32_t angle_step)
{
    if (!h) return MCL_ERR_INVALID_ARG;
    if (!ranges || angle_step <= 0 || n_ranges <= 0) return fail(h, MCL_ERR_INVALID_ARG, "bad scan");
    int32_t nb = (n_ranges + angle_step - 1) / angle_step;            // cpp:317: i = 0, ANGLE_STEP, ...
    return do_update(h, action, ranges, nb, nullptr, nullptr, true, angle_step);
}

int mcl_expected_pose(mcl_engine_t *h, double out[3])
{
    if (!h || !out) return MCL_ERR_INVALID_ARG;
    if (!h->have_particles) return MCL_ERR_NOT_READY;
    const double s = h->global_sums[0];
    // cpp:704-713 sums w_i*x_i with normalised w_i; here the division by sum(w) comes last
    double k = (s > 0.0) ? 1.0 / s : 1.0;
    out[0] = h->global_sums[1] * k;
    out[1] = h->global_sums[2] * k;
    out[2] = std::atan2(h->global_sums[3] * k, h->global_sums[4] * k);
    return MCL_OK;
}

int mcl_get_stage_timings(const mcl_engine_t *h, double ms[6])
{
    if (!h || !ms) return MCL_ERR_INVALID_ARG;
    std::memcpy(ms, h->timings, sizeof(h->timings));
    return MCL_OK;
}

int mcl_get_resample_indices(mcl_engine_t *h, int32_t *idx, int64_t n)
{
    if (!h || !idx) return MCL_ERR_INVALID_ARG;
    if (!h->have_idx) return MCL_ERR_NOT_READY;
    if (n != h->N) return MCL_ERR_INVALID_ARG;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    HIPCHK(h, hipMemcpyAsync(idx, h->d_idx, (size_t)n * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return MCL_OK;
}

int mcl_get_ray_steps(mcl_engine_t *h, uint8_t *steps, size_t n)
{
    if (!h || !steps) return MCL_ERR_INVALID_ARG;
    if (!h->cfg.keep_ray_steps) return fail(h, MCL_ERR_UNSUPPORTED, "engine created without keep_ray_steps");
    if (h->P > 255) return fail(h, MCL_ERR_UNSUPPORTED, "MAX_RANGE_PX > 255: step indices do not fit bytes, use mcl_get_ray_steps16");
    if (!h->have_steps) return MCL_ERR_NOT_READY;
    if (n != (size_t)h->N * h->B) return MCL_ERR_INVALID_ARG;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    HIPCHK(h, hipMemcpyAsync(steps, h->d_steps, n, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return MCL_OK;
}

int mcl_get_ray_steps16(mcl_engine_t *h, uint16_t *steps, size_t n)
{
    if (!h || !steps) return MCL_ERR_INVALID_ARG;
    if (!h->cfg.keep_ray_steps) return fail(h, MCL_ERR_UNSUPPORTED, "engine created without keep_ray_steps");
    if (!h->have_steps) return MCL_ERR_NOT_READY;
    if (n != (size_t)h->N * h->B) return MCL_ERR_INVALID_ARG;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    if (h->P > 255) {
        HIPCHK(h, hipMemcpyAsync(steps, h->d_steps, n * 2, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
    } else {
        std::vector<uint8_t> b(n);
        HIPCHK(h, hipMemcpyAsync(b.data(), h->d_steps, n, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        for (size_t k = 0; k < n; ++k) steps[k] = b[k];
    }
    return MCL_OK;
}

int mcl_get_log_weights(mcl_engine_t *h, double *logw, int64_t n)
{
    if (!h || !logw) return MCL_ERR_INVALID_ARG;
    if (!h->have_logw) return MCL_ERR_NOT_READY;
    if (n != h->N) return MCL_ERR_INVALID_ARG;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    HIPCHK(h, hipMemcpyAsync(logw, h->d_logw, (size_t)n * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return MCL_OK;
}

int mcl_get_counters(mcl_engine_t *h, uint64_t out[4])
{
    if (!h || !out) return MCL_ERR_INVALID_ARG;
    for (int i = 0; i < 4; ++i) out[i] = h->h_counters[i];
    return MCL_OK;
}

int mcl_get_compact_list(const mcl_engine_t *h, int64_t *n_entries, int32_t *used_by_last_update)
{
    if (!h || !n_entries || !used_by_last_update) return MCL_ERR_INVALID_ARG;
    *n_entries = h->compact_n;
    *used_by_last_update = h->compact_used ? 1 : 0;
    return MCL_OK;
}

int mcl_set_debug_count_probes(mcl_engine_t *h, int32_t on)
{
    if (!h) return MCL_ERR_INVALID_ARG;
    graph_reset(h);                        // a captured tail holds the non-counting kernel
    h->cfg.debug_count_probes = on ? 1 : 0;
    return MCL_OK;
}

int mcl_get_ray_kernel_ms(const mcl_engine_t *h, double *ms)
{
    if (!h || !ms) return MCL_ERR_INVALID_ARG;
    *ms = h->ray_ms;
    return MCL_OK;
}

int mcl_get_ray_kernel_id(const mcl_engine_t *h, int32_t *kernel)
{
    if (!h || !kernel) return MCL_ERR_INVALID_ARG;
    *kernel = h->last_mode;
    return MCL_OK;
}

int mcl_get_ray_kernel_variant(const mcl_engine_t *h, int32_t out[3])
{
    if (!h || !out) return MCL_ERR_INVALID_ARG;
    out[0] = h->last_sweep_global; out[1] = h->last_sweep_rec; out[2] = h->last_sweep_pairs;
    return MCL_OK;
}

int mcl_get_planned_ray_kernel(const mcl_engine_t *h, int64_t n_particles, int32_t *kernel, const char **reason)
{
    if (!h || !kernel) return MCL_ERR_INVALID_ARG;
    if (!h->have_map || h->B <= 0) return MCL_ERR_NOT_READY;
    const int64_t n = n_particles > 0 ? n_particles : (h->have_particles ? h->N : h->cap);
    const char *why = "";
    *kernel = choose_ray_mode(h, n, false, &why);
    if (reason) *reason = why ? why : "";
    return MCL_OK;
}

int mcl_get_effective_sample_size(const mcl_engine_t *h, double *n_eff, int32_t *resampled_last_update)
{
    if (!h || !n_eff || !resampled_last_update) return MCL_ERR_INVALID_ARG;
    if (!h->have_particles) return MCL_ERR_NOT_READY;
    const double sw = h->h_scalars[1], sww = h->h_scalars[7];
    *n_eff = sww > 0.0 ? sw * sw / sww : 0.0;
    *resampled_last_update = h->resampled_last ? 1 : 0;
    return MCL_OK;
}

int mcl_host_sensor_table(const mcl_config_t *cfg, int32_t P, double *out, size_t n)
{
    if (!cfg || !out || P < 1 || n != (size_t)(P + 1) * (P + 1)) return MCL_ERR_INVALID_ARG;
    std::vector<double> t;
    build_sensor_table(*cfg, P, t);
    std::memcpy(out, t.data(), n * sizeof(double));
    return MCL_OK;
}

int mcl_host_skip_field_dir(const int8_t *data, uint32_t width, uint32_t height, int32_t quadrant, uint8_t *out, size_t n)
{
    if (!data || !out || width == 0 || height == 0 || quadrant < 0 || quadrant > 3 || n != (size_t)(width + 1) * (height + 1))
        return MCL_ERR_INVALID_ARG;
    static const int qsx[4] = {1, -1, -1, 1}, qsy[4] = {1, 1, -1, -1};
    const int Wp = (int)width + 1, Hp = (int)height + 1, Wps = (Wp + 7) & ~7;
    std::vector<uint8_t> d;
    build_directional_field(data, (int)width, (int)height, Wp, Hp, Wps, qsx[quadrant], qsy[quadrant], d);
    for (int y = 0; y < Hp; ++y) std::memcpy(out + (size_t)y * Wp, d.data() + (size_t)y * Wps, Wp);
    return MCL_OK;
}

int mcl_host_skip_field_wedge(const int8_t *data, uint32_t width, uint32_t height, int32_t wedge, uint8_t *out, size_t n)
{
    if (!data || !out || width == 0 || height == 0 || wedge < 0 || wedge >= mcl::kWedges || n != (size_t)(width + 1) * (height + 1))
        return MCL_ERR_INVALID_ARG;
    const int W = (int)width, H = (int)height, Wp = W + 1, Hp = H + 1;
    std::vector<int32_t> nxt((size_t)Wp * Hp), prv((size_t)Wp * Hp);
    for (int y = 0; y < Hp; ++y) {
        auto stop = [&](int x) { return data[(size_t)std::max(y - 1, 0) * W + std::max(x - 1, 0)] > 50; };
        int last = -1;
        for (int x = 0; x < Wp; ++x) { if (stop(x)) last = x; prv[(size_t)y * Wp + x] = last; }
        int next = Wp;
        for (int x = Wp - 1; x >= 0; --x) { if (stop(x)) next = x; nxt[(size_t)y * Wp + x] = next; }
    }
    std::vector<mcl::WedgeRow> rows(2 * mcl::kWedgeR + 1);
    mcl::wedge_rows(wedge, rows.data());
    for (int y = 0; y < Hp; ++y)
        for (int x = 0; x < Wp; ++x) out[(size_t)y * Wp + x] = (uint8_t)mcl::wedge_skip_cell(nxt.data(), prv.data(), Wp, Hp, x, y, rows.data());
    return MCL_OK;
}

int mcl_host_skip_field(const int8_t *data, uint32_t width, uint32_t height, uint8_t *out, size_t n)
{
    if (!data || !out || width == 0 || height == 0 || n != (size_t)(width + 1) * (height + 1)) return MCL_ERR_INVALID_ARG;
    const int Wp = (int)width + 1, Hp = (int)height + 1, Wps = (Wp + 7) & ~7;
    std::vector<uint8_t> d;
    build_distance_field(data, (int)width, (int)height, Wp, Hp, Wps, d);
    for (int y = 0; y < Hp; ++y) std::memcpy(out + (size_t)y * Wp, d.data() + (size_t)y * Wps, Wp);
    return MCL_OK;
}

int mcl_host_sweep_global_layout(uint32_t width, uint32_t height, int32_t max_range_px, int64_t out[6])
{
    if (!out || width == 0 || height == 0 || max_range_px < 1) return MCL_ERR_INVALID_ARG;
    const int Wp = (int)width + 1, Hp = (int)height + 1;
    const mcl::SweepGlobalLayout g = mcl::sweep_global_layout(Wp, Hp, max_range_px);
    out[0] = g.ok ? 1 : 0; out[1] = g.pitch; out[2] = g.rows; out[3] = (int64_t)g.stride; out[4] = (int64_t)g.alloc;
    out[5] = (int64_t)mcl::sweep_global_max_offset(g, Wp, Hp, max_range_px, mcl::kWedges - 1);
    return MCL_OK;
}

// ---------------------------------------------------------------------------------------------
// multi-GPU staging
// ---------------------------------------------------------------------------------------------
int mcl_device_ptr(mcl_engine_t *h, int32_t which, void **p)
{
    if (!h || !p) return MCL_ERR_INVALID_ARG;
    switch (which) {
    case MCL_BUF_X: *p = h->d_x[h->cur]; break;
    case MCL_BUF_Y: *p = h->d_y[h->cur]; break;
    case MCL_BUF_THETA: *p = h->d_th[h->cur]; break;
    case MCL_BUF_QWEIGHT: *p = h->d_q; break;
    case MCL_BUF_LOGW: *p = h->d_logw; break;
    case MCL_BUF_SCALARS: *p = h->d_scalars; break;
    default: return MCL_ERR_INVALID_ARG;
    }
    return MCL_OK;
}

int mcl_export_state(mcl_engine_t *h, double *d_x, double *d_y, double *d_theta, uint64_t *d_q)
{
    if (!h) return MCL_ERR_INVALID_ARG;
    if (!h->have_particles) return MCL_ERR_NOT_READY;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const size_t nb = (size_t)h->N * 8;
    const int c = h->cur;
    if (d_x) HIPCHK(h, hipMemcpyAsync(d_x, h->d_x[c], nb, hipMemcpyDeviceToDevice, h->stream));
    if (d_y) HIPCHK(h, hipMemcpyAsync(d_y, h->d_y[c], nb, hipMemcpyDeviceToDevice, h->stream));
    if (d_theta) HIPCHK(h, hipMemcpyAsync(d_theta, h->d_th[c], nb, hipMemcpyDeviceToDevice, h->stream));
    if (d_q) HIPCHK(h, hipMemcpyAsync(d_q, h->d_q, nb, hipMemcpyDeviceToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return MCL_OK;
}

int mcl_export_records(mcl_engine_t *h, void *d_records)
{
    if (!h || !d_records) return MCL_ERR_INVALID_ARG;
    if (!h->have_particles) return MCL_ERR_NOT_READY;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const int c = h->cur;
    const int64_t n = h->N;
    if (h->pack_valid[c]) {
        HIPCHK(h, hipMemcpyAsync(d_records, h->d_pack[c], (size_t)n * sizeof(double4), hipMemcpyDeviceToDevice, h->stream));
    } else {
        hipLaunchKernelGGL(mcl::k_pack_records, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, h->d_x[c], h->d_y[c], h->d_th[c], n,
                           reinterpret_cast<double4 *>(d_records));
        HIPCHK(h, hipGetLastError());
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return MCL_OK;
}

int mcl_export_records_at(mcl_engine_t *h, const int64_t *d_index, int64_t count, void *d_out)
{
    if (!h || count < 0 || (count > 0 && (!d_index || !d_out))) return MCL_ERR_INVALID_ARG;
    if (!h->have_particles) return MCL_ERR_NOT_READY;
    if (count == 0) return MCL_OK;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const int c = h->cur;
    const int64_t n = h->N;
    if (!h->pack_valid[c]) {
        hipLaunchKernelGGL(mcl::k_pack_records, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, h->d_x[c], h->d_y[c], h->d_th[c], n, h->d_pack[c]);
        h->pack_valid[c] = true;
    }
    hipLaunchKernelGGL(mcl::k_gather_records, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, h->stream, h->d_pack[c], d_index, count, n,
                       reinterpret_cast<double4 *>(d_out));
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return MCL_OK;
}

int mcl_stage_distinct_parents(mcl_engine_t *h, const int32_t *d_parent, int64_t n_children, int64_t n_total, int64_t *d_distinct,
                               int32_t *d_slot, int64_t *count)
{
    if (!h || !d_parent || !d_distinct || !d_slot || !count || n_children <= 0 || n_total <= 0) return MCL_ERR_INVALID_ARG;
    if (n_total > MCL_MAX_TOTAL_PARTICLES) return fail(h, MCL_ERR_INVALID_ARG, "n_total exceeds MCL_MAX_TOTAL_PARTICLES");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const int64_t nwords = (n_total + 31) / 32;
    if ((size_t)nwords > h->bm_capacity) {
        dfree(h->d_bm); dfree(h->d_bm_pop); dfree(h->d_bm_pref);
        h->bm_capacity = 0;
        HIPCHK(h, hipMalloc(&h->d_bm, (size_t)nwords * 4));
        HIPCHK(h, hipMalloc(&h->d_bm_pop, (size_t)nwords * 8));
        HIPCHK(h, hipMalloc(&h->d_bm_pref, (size_t)nwords * 8));
        h->bm_capacity = (size_t)nwords;
    }
    size_t need = (size_t)nwords / mcl::kScanTile + 2;
    if (need > h->blocktot_capacity) {
        graph_reset(h);
        dfree(h->d_blocktot);
        HIPCHK(h, hipMalloc(&h->d_blocktot, need * 8));
        h->blocktot_capacity = need;
    }
    HIPCHK(h, hipMemsetAsync(h->d_bm, 0, (size_t)nwords * 4, h->stream));
    hipLaunchKernelGGL(mcl::k_bm_mark, dim3((unsigned)((n_children + 255) / 256)), dim3(256), 0, h->stream, d_parent, n_children, n_total, h->d_bm);
    hipLaunchKernelGGL(mcl::k_bm_pop, dim3((unsigned)((nwords + 255) / 256)), dim3(256), 0, h->stream, h->d_bm, nwords, h->d_bm_pop);
    int rc = scan_weights(h, h->d_bm_pop, h->d_bm_pref, nwords, 0, nullptr);
    if (rc) return rc;
    h->blocktot_for = nullptr;                  // the spine now describes the bitmap's prefix, not a CDF
    hipLaunchKernelGGL(mcl::k_bm_expand, dim3((unsigned)((nwords + 255) / 256)), dim3(256), 0, h->stream, h->d_bm, h->d_bm_pref, nwords, d_distinct);
    hipLaunchKernelGGL(mcl::k_bm_slot, dim3((unsigned)((n_children + 255) / 256)), dim3(256), 0, h->stream, d_parent, n_children, n_total, h->d_bm,
                       h->d_bm_pref, d_slot);
    HIPCHK(h, hipGetLastError());
    uint64_t c = 0;
    HIPCHK(h, hipMemcpyAsync(&c, h->d_bm_pref + (nwords - 1), 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    *count = (int64_t)c;
    return MCL_OK;
}

int mcl_get_scalars(mcl_engine_t *h, double out[8])
{
    if (!h || !out) return MCL_ERR_INVALID_ARG;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    HIPCHK(h, hipMemcpyAsync(out, h->d_scalars, 8 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return MCL_OK;
}

int mcl_get_host_scalars(const mcl_engine_t *h, double out[8])
{
    if (!h || !out) return MCL_ERR_INVALID_ARG;
    std::memcpy(out, h->h_scalars, 8 * sizeof(double));
    return MCL_OK;
}

// Where the parents of a staged resample come from.
struct ParentSource {
    const double *px = nullptr, *py = nullptr, *pth = nullptr;     // gathered columns, n_parents entries each
    const void *records = nullptr;                                  // gathered (or compacted) packed records
    const double4 *rank_records[mcl::kMaxShards] = {};              // one record array per shard (peer pointers, same process)
    int64_t n_per_rank = 0;
    int self_rank = 0;
    unsigned long long *remote_count = nullptr;
    const unsigned char *cchunks = nullptr;                         // the shards' compact lists, gathered as chunks
    int64_t cchunk_entries = 0;
    const uint64_t *gcdf = nullptr, *gtop = nullptr;                // their merged CDF (k_compact_merge)
    const int32_t *idx_in = nullptr;                                // parents decided by an earlier index-only pass (into `records`)
    int32_t *idx_only_out = nullptr;                                // index-only pass: parents go here, nothing else happens
    bool keep = false;                                              // adaptive resampling kept the set: every particle is its own parent (motion only)
};

// Launches the staged resample (+ motion) on the engine's stream; no synchronisation.  An index-only pass leaves the
// engine untouched; otherwise the children land in the other particle buffer, which becomes current.
static int stage_resample_launch(mcl_engine_t *h, const ParentSource &src, const uint64_t *d_cdf, int64_t n_parents, uint64_t q_total,
                                 int64_t child_first, int64_t n_children_total, const double action[3])
{
    if (!h) return MCL_ERR_INVALID_ARG;
    if (!ready(h, true)) return fail(h, MCL_ERR_NOT_READY, "map, beam angles and particles must be set first");
    const bool have_parents = src.records || src.n_per_rank > 0 || (src.px && src.py && src.pth) || src.idx_only_out || src.gcdf;
    if (!have_parents || (!d_cdf && !src.idx_in && !src.gcdf && !src.keep) || (!action && !src.idx_only_out) || n_parents <= 0 ||
        n_parents >= MCL_MAX_TOTAL_PARTICLES || n_children_total >= MCL_MAX_TOTAL_PARTICLES)
        return fail(h, MCL_ERR_INVALID_ARG, "bad stage_resample arguments (totals must stay below 2^27)");
    if (h->cfg.weight_mode != MCL_WEIGHT_LOG)
        return fail(h, MCL_ERR_UNSUPPORTED, "the staged (sharded) flow needs weight_mode LOG");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const int64_t n = h->N;
    const bool index_only = src.idx_only_out != nullptr;
    if (!index_only) {
        HIPCHK(h, hipMemsetAsync(h->d_counters, 0, 4 * sizeof(unsigned long long), h->stream));
        HIPCHK(h, hipEventRecord(h->ev[EV_START], h->stream));
    }
    const int nx = h->cur ^ 1;
    mcl::ResampleArgs a{};
    a.px = src.px; a.py = src.py; a.pth = src.pth; a.cdf = d_cdf; a.n_parents = n_parents; a.q_total = q_total;
    a.ppack = reinterpret_cast<const double4 *>(src.records);
    for (int r = 0; r < mcl::kMaxShards; ++r) a.ppack_rank[r] = src.rank_records[r];
    a.n_per_rank = src.n_per_rank; a.self_rank = src.self_rank; a.remote_count = src.remote_count;
    a.idx_in = src.idx_in; a.index_only = index_only ? 1 : 0;
    a.cpack = h->d_pack[nx];
    if (src.gcdf) {
        a.ccdf = src.gcdf; a.ctop = src.gtop; a.n_compact = n_parents; a.cchunks = src.cchunks;
        a.ccap = src.cchunk_entries; a.cchunk_bytes = src.cchunk_entries * 44;
        a.cpack = nullptr;
    }
    a.tile_excl = (d_cdf && h->blocktot_for == d_cdf && h->blocktot_n == n_parents) ? h->d_blocktot : nullptr;   // spine of the scan that produced d_cdf
    a.leaders = a.tile_excl ? h->d_leaders : nullptr;
    a.cx = h->d_x[nx]; a.cy = h->d_y[nx]; a.cth = h->d_th[nx];
    a.idx_out = src.idx_in ? nullptr : h->d_idx;       // the global parents of an earlier index-only pass stay in d_idx
    a.n_children = n; a.child_first = child_first; a.n_children_total = n_children_total;
    a.mode = h->cfg.resample_mode;
    a.seed_lo = (uint32_t)h->cfg.seed; a.seed_hi = (uint32_t)(h->cfg.seed >> 32);
    a.update_idx = h->update_idx;
    if (a.mode == MCL_RESAMPLE_SYSTEMATIC) {
        uint32_t c0 = 0, c1 = h->update_idx, c2 = 3, c3 = 0, k0 = a.seed_lo, k1 = a.seed_hi;
        for (int r = 0; r < 10; ++r) {
            uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
            uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
            c0 = n0; c1 = n1; c2 = n2; c3 = n3; k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
        }
        a.k0 = c0;
    }
    if (action) motion_scalars(action, a.dt, a.v, a.w);
    a.disp_x = h->cfg.motion_dispersion_x; a.disp_y = h->cfg.motion_dispersion_y; a.disp_th = h->cfg.motion_dispersion_theta;
    a.do_resample = src.keep ? 0 : 1; a.do_motion = 1;
    a.idx_out_base = src.keep ? child_first : 0;
    h->layout_stale_used = false; h->keys_done = false; h->pc_ready = false;
    // as in mcl_update: the ray stage's per-particle constants, its zeroed scratch and (by the previous update's layout) the sort
    // keys come out of this kernel; the layout of these children is made on the second stream for the next update
    if (!index_only) resample_ray_extras(h, n, a);
    if (!index_only) hipExtLaunchKernelGGL(mcl::k_resample_motion, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, nullptr, h->ev[EV_RESAMPLE], 0, a);
    else hipLaunchKernelGGL(mcl::k_resample_motion, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, a);
    HIPCHK(h, hipGetLastError());
    if (a.pc_out) { const int rc_l = layout_mark(h, n); if (rc_l) return rc_l; }
    if (index_only) {
        HIPCHK(h, hipMemcpyAsync(src.idx_only_out, h->d_idx, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToDevice, h->stream));
        h->have_idx = true;
        return MCL_OK;
    }
    h->cur = nx;
    h->pack_valid[nx] = a.cpack != nullptr;
    h->compact_used = src.gcdf != nullptr;
    h->have_idx = true;
    h->have_logw = false;
    h->resampled_last = !src.keep;
    h->stage_kept = src.keep;              // mcl_stage_rays adds the previous update's log-weights (minus their maximum)
    h->update_idx++;
    return MCL_OK;
}

static int stage_resample_sync(mcl_engine_t *h, const ParentSource &src, const uint64_t *d_cdf, int64_t n_parents, uint64_t q_total,
                               int64_t child_first, int64_t n_children_total, const double action[3])
{
    int rc = stage_resample_launch(h, src, d_cdf, n_parents, q_total, child_first, n_children_total, action);
    if (rc) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));           // the children (or the indices) are final: the host may export / gather them now
    return MCL_OK;
}

int mcl_stage_resample(mcl_engine_t *h, const double *d_px, const double *d_py, const double *d_pth, const uint64_t *d_cdf,
                       int64_t n_parents, uint64_t q_total, int64_t child_first, int64_t n_children_total, const double action[3])
{
    ParentSource src; src.px = d_px; src.py = d_py; src.pth = d_pth;
    if (!d_px || !d_py || !d_pth) return h ? fail(h, MCL_ERR_INVALID_ARG, "bad stage_resample arguments") : MCL_ERR_INVALID_ARG;
    return stage_resample_sync(h, src, d_cdf, n_parents, q_total, child_first, n_children_total, action);
}

int mcl_stage_resample_records(mcl_engine_t *h, const void *d_records, const uint64_t *d_cdf, int64_t n_parents, uint64_t q_total,
                               int64_t child_first, int64_t n_children_total, const double action[3])
{
    ParentSource src; src.records = d_records;
    if (!d_records) return h ? fail(h, MCL_ERR_INVALID_ARG, "bad stage_resample arguments") : MCL_ERR_INVALID_ARG;
    return stage_resample_sync(h, src, d_cdf, n_parents, q_total, child_first, n_children_total, action);
}

int mcl_stage_resample_indices(mcl_engine_t *h, const uint64_t *d_cdf, int64_t n_parents, uint64_t q_total, int64_t child_first,
                               int64_t n_children_total, int32_t *d_parent_idx)
{
    ParentSource src; src.idx_only_out = d_parent_idx;
    if (!d_parent_idx || !d_cdf) return h ? fail(h, MCL_ERR_INVALID_ARG, "bad stage_resample_indices arguments") : MCL_ERR_INVALID_ARG;
    return stage_resample_sync(h, src, d_cdf, n_parents, q_total, child_first, n_children_total, nullptr);
}

int mcl_stage_motion_records(mcl_engine_t *h, const void *d_records, int64_t n_records, const int32_t *d_record_of_child, int64_t child_first,
                             int64_t n_children_total, const double action[3])
{
    ParentSource src; src.records = d_records; src.idx_in = d_record_of_child;
    if (!d_records || !d_record_of_child || n_records <= 0) return h ? fail(h, MCL_ERR_INVALID_ARG, "bad stage_motion_records arguments") : MCL_ERR_INVALID_ARG;
    return stage_resample_sync(h, src, nullptr, n_records, 0, child_first, n_children_total, action);
}

// ---- sharded sets: the shards exchange their compact lists instead of every weight --------------------------------
int mcl_compact_chunk_bytes(int64_t chunk_entries, int64_t *bytes)
{
    if (!bytes || chunk_entries <= 0 || (chunk_entries & 63)) return MCL_ERR_INVALID_ARG;
    *bytes = chunk_entries * 44;             // [ccdf: 8 | crec: 32 | cidx: 4] per entry, column by column
    return MCL_OK;
}

static int export_compact_launch(mcl_engine_t *h, void *d_chunk, int64_t chunk_entries, int dst_device, hipStream_t stream)
{
    if (!h || !d_chunk || chunk_entries <= 0 || (chunk_entries & 63)) return MCL_ERR_INVALID_ARG;
    if (h->compact_n < 0 || h->compact_n > chunk_entries) return fail(h, MCL_ERR_NOT_READY, "no compact list of that size (mcl_get_compact_list)");
    if (h->compact_n == 0) return MCL_OK;
    unsigned char *c = static_cast<unsigned char *>(d_chunk);
    const size_t n = (size_t)h->compact_n, cap = (size_t)chunk_entries;
    const int src = h->cfg.device;
    HIPCHK(h, hipMemcpyPeerAsync(c, dst_device, h->d_ccdf, src, n * 8, stream));
    HIPCHK(h, hipMemcpyPeerAsync(c + cap * 8, dst_device, h->d_crec, src, n * 32, stream));
    HIPCHK(h, hipMemcpyPeerAsync(c + cap * 40, dst_device, h->d_cidx, src, n * 4, stream));
    return MCL_OK;
}

int mcl_export_compact(mcl_engine_t *h, void *d_chunk, int64_t chunk_entries)
{
    if (!h) return MCL_ERR_INVALID_ARG;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const int rc = export_compact_launch(h, d_chunk, chunk_entries, h->cfg.device, h->stream);
    if (rc) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return MCL_OK;
}

// merge of the gathered chunks + staged resample from them, on the engine's stream; no synchronisation
static int stage_resample_compact_launch(mcl_engine_t *h, const void *d_chunks, int32_t n_shards, int64_t chunk_entries, const int64_t *counts,
                                         const uint64_t *totals, int64_t n_per_shard, int32_t self_shard, int64_t child_first,
                                         int64_t n_children_total, const double action[3], unsigned long long *remote_count)
{
    if (!h) return MCL_ERR_INVALID_ARG;
    if (!d_chunks || !counts || !totals || n_shards <= 0 || n_shards > mcl::kMaxShards || chunk_entries <= 0 || (chunk_entries & 63) || n_per_shard <= 0 ||
        self_shard < 0 || self_shard >= n_shards)
        return fail(h, MCL_ERR_INVALID_ARG, "bad stage_resample_compact arguments");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    mcl::MergeArgs m{};
    m.chunks = static_cast<const unsigned char *>(d_chunks); m.chunk_bytes = chunk_entries * 44; m.ccap = chunk_entries; m.n_shards = n_shards;
    uint64_t off = 0;
    for (int r = 0; r < n_shards; ++r) {
        if (counts[r] < 0 || counts[r] > chunk_entries) return fail(h, MCL_ERR_INVALID_ARG, "a list is longer than its chunk");
        m.count[r] = (uint32_t)counts[r]; m.off[r] = off; m.tot[r] = counts[r] > 0 ? totals[r] : 0ull;
        off += m.tot[r];
    }
    if (off == 0) return fail(h, MCL_ERR_INVALID_ARG, "the lists carry no weight");
    const size_t total = (size_t)n_shards * (size_t)chunk_entries;
    if (total >= (size_t)MCL_MAX_TOTAL_PARTICLES) return fail(h, MCL_ERR_INVALID_ARG, "gathered lists exceed 2^27 entries");
    if (total > h->gcdf_capacity) {
        dfree(h->d_gcdf); dfree(h->d_gtop);
        h->gcdf_capacity = 0;
        HIPCHK(h, hipMalloc(&h->d_gcdf, total * 8));
        HIPCHK(h, hipMalloc(&h->d_gtop, (total / 64 + 1) * 8));
        h->gcdf_capacity = total;
    }
    m.gcdf = h->d_gcdf; m.gtop = h->d_gtop;
    hipLaunchKernelGGL(mcl::k_compact_merge, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, h->stream, m);
    HIPCHK(h, hipGetLastError());
    ParentSource src;
    src.cchunks = m.chunks; src.cchunk_entries = chunk_entries; src.gcdf = h->d_gcdf; src.gtop = h->d_gtop;
    src.n_per_rank = n_per_shard; src.self_rank = self_shard; src.remote_count = remote_count;
    return stage_resample_launch(h, src, nullptr, (int64_t)total, off, child_first, n_children_total, action);
}

int mcl_stage_resample_compact(mcl_engine_t *h, const void *d_chunks, int32_t n_shards, int64_t chunk_entries, const int64_t *counts,
                               const uint64_t *totals, int64_t n_per_shard, int32_t self_shard, int64_t child_first, int64_t n_children_total,
                               const double action[3])
{
    const int rc = stage_resample_compact_launch(h, d_chunks, n_shards, chunk_entries, counts, totals, n_per_shard, self_shard, child_first,
                                                 n_children_total, action, nullptr);
    if (rc) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return MCL_OK;
}

static int stage_rays_launch(mcl_engine_t *h, const float *obs, int32_t n_beams, bool force_skip, double *d_max_out = nullptr)
{
    if (!h) return MCL_ERR_INVALID_ARG;
    if (!ready(h, true)) return fail(h, MCL_ERR_NOT_READY, "map, beam angles and particles must be set first");
    if (!obs || n_beams != h->B) return fail(h, MCL_ERR_INVALID_ARG, "bad observation");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const int64_t n = h->N;
    const int c = h->cur;
    int rc = MCL_OK;
    if (!force_skip) {
        rc = prepare_observation(h, obs, 1);
        if (rc) return rc;
        HIPCHK(h, hipEventRecord(h->ev[EV_QUERY], h->stream));
    } else {
        HIPCHK(h, hipMemsetAsync(h->d_counters, 0, 4 * sizeof(unsigned long long), h->stream));
    }
    if (!h->pc_ready) { h->layout_stale_used = false; h->keys_done = false; }     // no staged resampling before this call: nothing prepared
    h->ev_rays_bound = false;
    rc = launch_rays(h, h->d_x[c], h->d_y[c], h->d_th[c], n, force_skip);
    if (rc) return rc;
    rc = next_layout_launch(h, n);
    if (rc) return rc;
    if (h->stage_kept) {
        hipLaunchKernelGGL(mcl::k_add_carry, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, h->d_logw, h->d_carry[h->carry_idx], n);
        h->max_partials_ready = false;
        HIPCHK(h, hipEventRecord(h->ev[EV_RAYS], h->stream));
    } else if (!h->ev_rays_bound) HIPCHK(h, hipEventRecord(h->ev[EV_RAYS], h->stream));
    if (!h->max_partials_ready)
        hipLaunchKernelGGL(mcl::k_reduce_max, dim3(mcl::kRedBlocks), dim3(mcl::kRedThreads), 0, h->stream, h->d_logw, n, h->d_maxpart);
    hipLaunchKernelGGL(mcl::k_final_max, dim3(1), dim3(mcl::kRedThreads), 0, h->stream, h->d_maxpart, mcl::kRedBlocks, h->d_scalars, d_max_out);
    h->max_partials_ready = false;
    HIPCHK(h, hipGetLastError());
    if (!d_max_out)          // (the device-ordered flows read the result block once, after the weights)
        HIPCHK(h, hipMemcpyAsync(h->h_result, h->d_result, kResultWords * 8, hipMemcpyDeviceToHost, h->stream));
    return MCL_OK;
}

// the host-side notes of a finished ray stage (its stream has been waited for)
static void stage_rays_note(mcl_engine_t *h)
{
    h->have_logw = true;
    h->have_steps = h->cfg.keep_ray_steps != 0;
    h->timings[0] = elapsed(h->ev[EV_START], h->ev[EV_RESAMPLE]);
    h->timings[1] = 0.0;
    h->timings[2] = 0.0;
    h->timings[3] = elapsed(h->ev[EV_QUERY], h->ev[EV_RAYS]);
    h->ray_ms = elapsed(h->ev[EV_K0], h->ev[EV_K1]);
}

static int stage_rays_finish(mcl_engine_t *h, const float *obs, int32_t n_beams)
{
    HIPCHK(h, hipSetDevice(h->cfg.device));
    for (int attempt = 0; attempt < 2; ++attempt) {
        HIPCHK(h, hipStreamSynchronize(h->stream));
        std::memcpy(h->h_scalars, h->h_result, 8 * sizeof(double));       // [0] = local max log-weight (mcl_get_host_scalars)
        std::memcpy(h->h_counters, h->h_result + 8, 4 * sizeof(unsigned long long));
        h->h_fix_count = h->h_result[12];
        if (!(h->last_quad && h->h_fix_count != 0) || attempt == 1) break;
        // work-list overflow (only with debug_force_exact at large sizes): once more with the self-contained k_rays_skip
        int rc = stage_rays_launch(h, obs, n_beams, true);
        if (rc) return rc;
    }
    stage_rays_note(h);
    return MCL_OK;
}

int mcl_stage_rays(mcl_engine_t *h, const float *obs, int32_t n_beams)
{
    int rc = stage_rays_launch(h, obs, n_beams, false);
    if (rc) return rc;
    return stage_rays_finish(h, obs, n_beams);
}

int mcl_stage_propagate(mcl_engine_t *h, const double *d_px, const double *d_py, const double *d_pth, const uint64_t *d_cdf,
                        int64_t n_parents, uint64_t q_total, int64_t child_first, int64_t n_children_total,
                        const double action[3], const float *obs, int32_t n_beams)
{
    if (!h) return MCL_ERR_INVALID_ARG;
    if (!obs || n_beams != h->B) return fail(h, MCL_ERR_INVALID_ARG, "bad observation");
    int rc = mcl_stage_resample(h, d_px, d_py, d_pth, d_cdf, n_parents, q_total, child_first, n_children_total, action);
    if (rc) return rc;
    return mcl_stage_rays(h, obs, n_beams);
}

int mcl_set_reserved_cus(mcl_engine_t *h, int32_t n_cus)
{
    if (!h || n_cus < 0 || n_cus >= h->num_cu) return MCL_ERR_INVALID_ARG;
    h->reserved_cus = n_cus;
    return MCL_OK;
}

// d_global_max: the global maximum in device memory (the device-ordered flow: the log-weights need not have been seen by the
// host yet), else the host's value is staged
static int stage_weights_launch(mcl_engine_t *h, double global_max_logw, const double *d_global_max = nullptr)
{
    if (!h) return MCL_ERR_INVALID_ARG;
    if (!h->have_logw && !d_global_max) return MCL_ERR_NOT_READY;
    if (h->cfg.weight_mode != MCL_WEIGHT_LOG)
        return fail(h, MCL_ERR_UNSUPPORTED, "the staged (sharded) flow needs weight_mode LOG");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    if (d_global_max) {
        hipLaunchKernelGGL(mcl::k_copy_double, dim3(1), dim3(1), 0, h->stream, d_global_max, h->d_scalars);
    } else {
        h->h_result[kResultStage] = 0;
        std::memcpy(&h->h_result[kResultStage], &global_max_logw, sizeof(double));   // pinned: stays valid until the copy has run
        HIPCHK(h, hipMemcpyAsync(h->d_scalars, &h->h_result[kResultStage], sizeof(double), hipMemcpyHostToDevice, h->stream));
    }
    int rc = weight_stats(h, true, h->d_scalars, true);        // (the scan below finishes the sums)
    if (rc) return rc;
    // (adaptive resampling: the carry this wrote -- logw minus the GLOBAL maximum -- becomes current in mcl_stage_finish, once the
    //  host knows that the update is not redone from the ray stage on)
    // the shard's own CDF follows its new weights: mcl_sample_particles (visualize) and a later plain mcl_update search it
    h->bind_sensor_event = true; h->ev_sensor_bound = false;
    rc = scan_weights(h, h->d_q, h->d_cdf, h->N, 0, nullptr);
    h->bind_sensor_event = false;
    if (rc) return rc;
    if (!h->ev_sensor_bound) HIPCHK(h, hipEventRecord(h->ev[EV_SENSOR], h->stream));
    HIPCHK(h, hipMemcpyAsync(h->h_result, h->d_result, kResultWords * 8, hipMemcpyDeviceToHost, h->stream));
    return MCL_OK;
}

static int stage_weights_finish(mcl_engine_t *h)
{
    HIPCHK(h, hipSetDevice(h->cfg.device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    unpack_result(h);
    h->timings[4] = elapsed(h->ev[EV_RAYS], h->ev[EV_SENSOR]);
    return layout_adopt(h, h->N);
}

int mcl_stage_weights(mcl_engine_t *h, double global_max_logw)
{
    int rc = stage_weights_launch(h, global_max_logw);
    if (rc) return rc;
    return stage_weights_finish(h);
}

// end of a staged update: the carry its weights stage wrote becomes the current one
static void stage_commit_carry(mcl_engine_t *h)
{
    if (h->carry_pending) { h->carry_idx ^= 1; h->carry_valid = true; h->carry_pending = false; }
    h->stage_kept = false;
}

int mcl_stage_finish(mcl_engine_t *h, const double global_sums[5])
{
    if (!h || !global_sums) return MCL_ERR_INVALID_ARG;
    for (int i = 0; i < 5; ++i) h->global_sums[i] = global_sums[i];
    stage_commit_carry(h);
    return MCL_OK;
}

// Adaptive resampling in a sharded set (E9).  The DECISION is the host's: keep the set when (sum w)^2 >= r / 1000 * N * sum w^2 over
// the WHOLE set (the sums of the previous update: five in the exchanged vector, sum w^2 at its end; r = resample_neff_permille),
// the same numbers on every shard.  mcl_stage_keep then takes the place of the exchange and the resampling: every particle is its
// own parent (no collective at all), the motion model is applied with the same random streams, and mcl_stage_rays adds the
// previous update's log-weights minus their global maximum, as mcl_update does.  Launch only (no wait).
int mcl_stage_keep(mcl_engine_t *h, int64_t child_first, int64_t n_children_total, const double action[3])
{
    if (!h || !action) return MCL_ERR_INVALID_ARG;
    if (h->cfg.resample_neff_permille <= 0) return fail(h, MCL_ERR_UNSUPPORTED, "mcl_stage_keep needs resample_neff_permille > 0");
    if (!h->carry_valid) return fail(h, MCL_ERR_NOT_READY, "no log-weights of a previous update to carry");
    ParentSource src;
    const int c = h->cur;
    src.px = h->d_x[c]; src.py = h->d_y[c]; src.pth = h->d_th[c];
    src.keep = true;
    return stage_resample_launch(h, src, nullptr, h->N, 0, child_first, n_children_total, action);
}

// ---- the staged flow ordered on the device: nothing below waits for the stream except mcl_stage_complete ----------------
int mcl_stream_wait_external(mcl_engine_t *h, void *stream)
{
    if (!h) return MCL_ERR_INVALID_ARG;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    HIPCHK(h, hipEventRecord(h->ev_ext_in, static_cast<hipStream_t>(stream)));
    HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_ext_in, 0));
    return MCL_OK;
}

int mcl_external_wait_stream(mcl_engine_t *h, void *stream)
{
    if (!h) return MCL_ERR_INVALID_ARG;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    HIPCHK(h, hipEventRecord(h->ev_ext_out, h->stream));
    HIPCHK(h, hipStreamWaitEvent(static_cast<hipStream_t>(stream), h->ev_ext_out, 0));
    return MCL_OK;
}

int mcl_export_compact_async(mcl_engine_t *h, void *d_chunk, int64_t chunk_entries)
{
    if (!h) return MCL_ERR_INVALID_ARG;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    return export_compact_launch(h, d_chunk, chunk_entries, h->cfg.device, h->stream);
}

int mcl_stage_resample_compact_async(mcl_engine_t *h, const void *d_chunks, int32_t n_shards, int64_t chunk_entries, const int64_t *counts,
                                     const uint64_t *totals, int64_t n_per_shard, int32_t self_shard, int64_t child_first,
                                     int64_t n_children_total, const double action[3])
{
    return stage_resample_compact_launch(h, d_chunks, n_shards, chunk_entries, counts, totals, n_per_shard, self_shard, child_first,
                                         n_children_total, action, nullptr);
}

int mcl_stage_rays_async(mcl_engine_t *h, const float *obs, int32_t n_beams, double *d_local_max)
{
    if (!h || !d_local_max) return MCL_ERR_INVALID_ARG;
    const int rc = stage_rays_launch(h, obs, n_beams, false, d_local_max);
    if (rc) return rc;
    h->stage_async_rays = true;
    return MCL_OK;
}

int mcl_stage_weights_async(mcl_engine_t *h, const double *d_global_max, double *d_vec, int32_t n_shards, int32_t self_shard)
{
    if (!h || !d_global_max || !d_vec || n_shards <= 0 || n_shards > mcl::kMaxShards || self_shard < 0 || self_shard >= n_shards)
        return MCL_ERR_INVALID_ARG;
    if (!h->stage_async_rays) return fail(h, MCL_ERR_NOT_READY, "mcl_stage_rays_async first");
    const int rc = stage_weights_launch(h, 0.0, d_global_max);       // ends with the copy of the result block to pinned memory
    if (rc) return rc;
    hipLaunchKernelGGL(mcl::k_stage_pack, dim3(1), dim3(64), 0, h->stream, h->d_result, d_vec, n_shards, self_shard, h->compact_pending ? 1 : 0,
                       (unsigned long long)h->compact_cap);
    HIPCHK(h, hipGetLastError());
    h->stage_async_weights = true;
    return MCL_OK;
}

int mcl_stage_complete(mcl_engine_t *h, const double global_sums[5], int32_t *redo)
{
    if (!h || !global_sums || !redo) return MCL_ERR_INVALID_ARG;
    if (!h->stage_async_rays || !h->stage_async_weights) return fail(h, MCL_ERR_NOT_READY, "mcl_stage_rays_async and mcl_stage_weights_async first");
    h->stage_async_rays = h->stage_async_weights = false;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    unpack_result(h);
    stage_rays_note(h);
    h->timings[4] = elapsed(h->ev[EV_RAYS], h->ev[EV_SENSOR]);
    const int rc = layout_adopt(h, h->N);
    if (rc) return rc;
    // more undecided rays than the fix-up lists hold (debug_force_exact at large sizes, a pathological map): the log-weights of
    // this update are incomplete.  The caller runs the ray stage again through mcl_stage_rays (which falls back to the
    // self-contained kernel) and the two exchanges after it; the children are untouched
    *redo = (h->last_quad && h->h_fix_count != 0) ? 1 : 0;
    for (int i = 0; i < 5; ++i) h->global_sums[i] = global_sums[i];
    // (the carry of adaptive resampling is committed by mcl_stage_finish, which the host calls once the update stands)
    return MCL_OK;
}


// ---------------------------------------------------------------------------------------------
// One process per GPU, the exchange in native code: the engine holds an RCCL communicator and one call runs a whole sharded
// update -- the three collectives of an update (all-gather of the compact parent lists, all-reduce MAX of one double,
// all-reduce SUM of 5 + 3 G + 1 doubles) are enqueued on the ENGINE'S OWN STREAM between its kernels: no second stream, no
// event hop, no interpreter between the stages; the host waits once, for the summed vector.  RCCL is taken from the process
// at run time (dlopen: the library a torch process already carries, else the ROCm one): the engine does not link it, and a
// host without RCCL keeps every other entry point.  The rendezvous (128-byte id from rank 0 to every rank) is the host's.
// ---------------------------------------------------------------------------------------------
// The few RCCL types and enumerators the entry points below need, declared here: the library is taken with dlopen at run time
// and must build on a ROCm tree without the RCCL development headers (values as in rccl/rccl.h: they are RCCL's / NCCL's ABI).
extern "C" {
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclChar = 0, ncclUint64 = 5, ncclDouble = 8 } ncclDataType_t;
typedef enum { ncclSum = 0, ncclMax = 2 } ncclRedOp_t;
}

struct RcclApi {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string why;
};

static RcclApi &rccl_api()
{
    static RcclApi api;
    if (api.lib || !api.why.empty()) return api;
    const char *names[] = {"librccl.so.1", "librccl.so"};
    for (const char *nm : names) {
        api.lib = dlopen(nm, RTLD_NOW | RTLD_NOLOAD);        // the copy the process already has (a torch process: torch's)
        if (api.lib) break;
    }
    for (const char *nm : names) {
        if (api.lib) break;
        api.lib = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
    }
    if (!api.lib) { api.why = std::string("RCCL not found: ") + (dlerror() ? dlerror() : "dlopen failed"); return api; }
    auto sym = [&](const char *n) { void *p = dlsym(api.lib, n); if (!p && api.why.empty()) api.why = std::string("RCCL symbol missing: ") + n; return p; };
    api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
    api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
    api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
    api.CommAbort = reinterpret_cast<decltype(api.CommAbort)>(sym("ncclCommAbort"));
    api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(sym("ncclAllReduce"));
    api.AllGather = reinterpret_cast<decltype(api.AllGather)>(sym("ncclAllGather"));
    api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
    if (!api.why.empty()) api.lib = nullptr;
    return api;
}

struct mcl_comm {
    ncclComm_t comm = nullptr;
    int n_ranks = 0, rank = 0;
    unsigned char *d_chunk_local = nullptr, *d_chunk_all = nullptr;     // the lists: this shard's chunk, every shard's
    size_t chunk_capacity = 0;                                           // entries per chunk the buffers hold
    double *d_red = nullptr;                                             // [0] MAX exchange | [1 .. k] SUM exchange | [1 + k] error word (ranks that failed)
    double *h_red = nullptr;                                             // pinned copy of it
    uint64_t bytes_received = 0, bytes_payload = 0;                      // of the last update's list exchange
    int host_waits = 0;
    // dense exchange (an update without lists: the first after the particles were set): every shard's fixed-point weights,
    // their global CDF and every shard's packed records; allocated when first needed
    uint64_t *d_qall = nullptr, *d_cdfall = nullptr;
    double4 *d_recall = nullptr;
    size_t dense_capacity = 0;                                           // particles (all shards) the three arrays hold
    bool last_dense = false, last_kept = false;
    bool vec_valid = false;                                              // vec holds the sums of an update of the current particle set
    uint64_t dense_weights_bytes = 0, dense_records_bytes = 0;
    // what the shards' lists look like (the previous update's summed vector, or mcl_comm_set_lists after a dense update)
    bool lists_known = false;
    int64_t counts[mcl::kMaxShards] = {};
    uint64_t totals[mcl::kMaxShards] = {};
    double vec[5 + 3 * mcl::kMaxShards + 2] = {};                        // the last summed vector
    // failure protocol (mcl_comm_update): how long a host wait may last before the communicator is aborted; the update count;
    // MCL_COMM_FAIL = "<rank>:<update>:<stage>" (test switch: that rank reports a failure / stalls before that stage of that update)
    double timeout_ms = 30000.0;
    bool dead = false;                                                   // aborted (a wait ran out, or a collective call failed): create again
    unsigned long long updates = 0;
    int fail_rank = -1; long long fail_update = -1; std::string fail_stage;
    // the lists of the NEXT update, gathered right after this update's sums (beside whatever the host does between updates)
    bool gathered = false;
    unsigned long long gathered_epoch = 0;
    unsigned long long gathered_epoch_all = 0;                           // the update (count) after which EVERY rank pre-gathered the lists
    int64_t gathered_entries = 0;
    int64_t gathered_counts[mcl::kMaxShards] = {};
};

static void comm_forget(mcl_comm *c)
{
    if (!c) return;
    c->lists_known = false;
    c->gathered = false;
    c->gathered_entries = 0;
    c->vec_valid = false;
}

static void comm_free(mcl_comm *c)
{
    if (!c) return;
    if (c->comm && rccl_api().CommDestroy) (void)rccl_api().CommDestroy(c->comm);        // (an aborted communicator is gone already: comm == nullptr)
    if (c->d_chunk_local) (void)hipFree(c->d_chunk_local);
    if (c->d_chunk_all) (void)hipFree(c->d_chunk_all);
    if (c->d_red) (void)hipFree(c->d_red);
    if (c->h_red) (void)hipHostFree(c->h_red);
    if (c->d_qall) (void)hipFree(c->d_qall);
    if (c->d_cdfall) (void)hipFree(c->d_cdfall);
    if (c->d_recall) (void)hipFree(c->d_recall);
    delete c;
}

// ---- failure protocol of a sharded update --------------------------------------------------------------------------------------
// A rank that fails must not hang its peers (the reference's error model is log-and-skip-a-tick, cpp:236-240, 680, 756; a node
// blocked inside a collective for ever is not that).  Two layers:
//   * SOFT failure -- anything a rank finds wrong locally while its communicator still works (an engine stage that fails, list
//     state that does not match the exchange, MCL_COMM_FAIL): the rank stops its local work but STILL ISSUES EVERY COLLECTIVE of
//     the update, with the sizes every rank derives from the shared numbers, and raises the ERROR WORD that rides behind the
//     summed vector (one more double of the SUM all-reduce: the number of ranks that failed).  Every rank then returns from the
//     same update: the failing ones with their own status and message, the others with MCL_ERR_PEER.  The communicator stays
//     usable; the particle set must be set or initialised again on every rank (as after MCL_ERR_HIP from mcl_update), and the
//     next update is a dense one.
//   * HARD failure -- a collective call that fails, or a host wait that lasts longer than MCL_COMM_TIMEOUT_MS (default 30 000;
//     a peer that died or never arrived): the communicator is ABORTED (ncclCommAbort: the collective kernels in flight end), the
//     call returns MCL_ERR_TIMEOUT / MCL_ERR_HIP, and mcl_comm_create must run again (on every rank: their waits run out too).
// Nothing here re-executes a process; recovery that needs a new process is the host's (a fresh child, never an exec of a
// process that has touched the GPU).
static void comm_abort(mcl_engine_t *h, const char *why)
{
    mcl_comm *c = h->comm;
    if (!c) return;
    RcclApi &api = rccl_api();
    if (c->comm) {
        if (api.CommAbort) (void)api.CommAbort(c->comm);
        // (without ncclCommAbort in the library the communicator is leaked rather than destroyed: ncclCommDestroy waits for peers)
        c->comm = nullptr;
    }
    c->dead = true;
    comm_forget(c);
    (void)hipStreamSynchronize(h->stream);          // the aborted kernels and whatever else was enqueued drain
    (void)why;
}

#define NCCLCHK(h, call)                                                                                          \
    do {                                                                                                          \
        ncclResult_t r_ = (call);                                                                                 \
        if (r_ != ncclSuccess) {                                                                                  \
            const std::string m_ = std::string(#call) + ": " + rccl_api().GetErrorString(r_) + " (communicator aborted: mcl_comm_create again)"; \
            comm_abort(h, "collective call failed");                                                              \
            return fail(h, MCL_ERR_HIP, m_);                                                                      \
        }                                                                                                         \
    } while (0)

// the host wait of a sharded update, bounded: MCL_OK when the engine's stream has drained, else the communicator is aborted
static int comm_wait(mcl_engine_t *h)
{
    mcl_comm *c = h->comm;
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0;; ++spins) {
        const hipError_t e = hipStreamQuery(h->stream);
        if (e == hipSuccess) { c->host_waits += 1; return MCL_OK; }
        if (e != hipErrorNotReady) {
            const std::string m = std::string("hipStreamQuery: ") + hipGetErrorString(e);
            comm_abort(h, "stream error");
            return fail(h, MCL_ERR_HIP, m);
        }
        if ((spins & 63u) == 63u) {
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            if (ms > c->timeout_ms) break;
        }
    }
    comm_abort(h, "timeout");
    return fail(h, MCL_ERR_TIMEOUT, "a collective of the sharded update did not finish within MCL_COMM_TIMEOUT_MS (a peer failed or never arrived): "
                                    "communicator aborted, mcl_comm_create again");
}

int mcl_comm_available(const char **why)
{
    RcclApi &api = rccl_api();
    if (why) *why = api.lib ? "" : api.why.c_str();
    return api.lib ? MCL_OK : MCL_ERR_UNSUPPORTED;
}

int mcl_comm_unique_id(unsigned char id[128])
{
    RcclApi &api = rccl_api();
    if (!id) return MCL_ERR_INVALID_ARG;
    if (!api.lib) return MCL_ERR_UNSUPPORTED;
    static_assert(sizeof(ncclUniqueId) == 128, "RCCL unique id is 128 bytes");
    ncclUniqueId u;
    if (api.GetUniqueId(&u) != ncclSuccess) return MCL_ERR_HIP;
    std::memcpy(id, &u, 128);
    return MCL_OK;
}

int mcl_comm_create(mcl_engine_t *h, const unsigned char id[128], int32_t n_ranks, int32_t rank)
{
    if (!h || !id || n_ranks <= 0 || n_ranks > mcl::kMaxShards || rank < 0 || rank >= n_ranks) return MCL_ERR_INVALID_ARG;
    RcclApi &api = rccl_api();
    if (!api.lib) return fail(h, MCL_ERR_UNSUPPORTED, api.why);
    if (h->cfg.weight_mode != MCL_WEIGHT_LOG)
        return fail(h, MCL_ERR_UNSUPPORTED, "a sharded set needs weight_mode LOG");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    if (h->comm) { comm_free(h->comm); h->comm = nullptr; }
    mcl_comm *c = new mcl_comm();
    c->n_ranks = n_ranks; c->rank = rank;
    if (const char *e = getenv("MCL_COMM_TIMEOUT_MS")) { const double v = atof(e); if (v > 0.0) c->timeout_ms = v; }
    if (const char *e = getenv("MCL_COMM_FAIL")) {                    // "<rank>:<update>:<stage>", stage = resample | rays | weights | stall
        int fr = -1; long long fu = -1; char st[32] = {0};
        if (std::sscanf(e, "%d:%lld:%31s", &fr, &fu, st) == 3) { c->fail_rank = fr; c->fail_update = fu; c->fail_stage = st; }
    }
    ncclUniqueId u;
    std::memcpy(&u, id, 128);
    const ncclResult_t r = api.CommInitRank(&c->comm, n_ranks, u, rank);          // collective: every rank is in this call
    if (r != ncclSuccess) { c->comm = nullptr; comm_free(c); return fail(h, MCL_ERR_HIP, std::string("ncclCommInitRank: ") + api.GetErrorString(r)); }
    const size_t words = 1 + 5 + 3 * (size_t)n_ranks + 2 + 1;                     // MAX | summed vector | error word
    if (hipMalloc(&c->d_red, words * 8) != hipSuccess || hipHostMalloc(&c->h_red, words * 8) != hipSuccess) {
        comm_free(c);
        return fail(h, MCL_ERR_HIP, "mcl_comm_create: allocation failed");
    }
    h->comm = c;
    return MCL_OK;
}

// The three collectives of an update on known data, before any particle depends on them: all-reduce MAX of the rank, all-reduce
// SUM of ones, all-gather of one 64-byte chunk per rank.  COLLECTIVE.  A host that finds a rank failing here keeps its other
// exchange (dist.py: torch's collectives) instead of learning it in the first update.
int mcl_comm_selftest(mcl_engine_t *h)
{
    if (!h) return MCL_ERR_INVALID_ARG;
    mcl_comm *c = h->comm;
    if (!c) return fail(h, MCL_ERR_NOT_READY, "mcl_comm_create first");
    if (c->dead) return fail(h, MCL_ERR_NOT_READY, "the communicator was aborted: mcl_comm_create again");
    RcclApi &api = rccl_api();
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const int G = c->n_ranks;
    unsigned char *d_buf = nullptr;
    HIPCHK(h, hipMalloc(&d_buf, 64 * (size_t)(G + 1)));
    unsigned char mine[64];
    for (int i = 0; i < 64; ++i) mine[i] = (unsigned char)(c->rank * 7 + i);
    double two[2] = {(double)c->rank, 1.0};
    auto run = [&]() -> int {
        HIPCHK(h, hipMemcpyAsync(d_buf, mine, 64, hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(c->d_red, two, 16, hipMemcpyHostToDevice, h->stream));
        NCCLCHK(h, api.AllReduce(c->d_red, c->d_red, 1, ncclDouble, ncclMax, c->comm, h->stream));
        NCCLCHK(h, api.AllReduce(c->d_red + 1, c->d_red + 1, 1, ncclDouble, ncclSum, c->comm, h->stream));
        NCCLCHK(h, api.AllGather(d_buf, d_buf + 64, 64, ncclChar, c->comm, h->stream));
        std::vector<unsigned char> all(64 * (size_t)G);
        HIPCHK(h, hipMemcpyAsync(c->h_red, c->d_red, 16, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipMemcpyAsync(all.data(), d_buf + 64, all.size(), hipMemcpyDeviceToHost, h->stream));
        const int rw = comm_wait(h);
        if (rw) return rw;
        if (c->h_red[0] != (double)(G - 1) || c->h_red[1] != (double)G) return fail(h, MCL_ERR_HIP, "mcl_comm_selftest: an all-reduce returned a wrong value");
        for (int r = 0; r < G; ++r)
            for (int i = 0; i < 64; ++i)
                if (all[(size_t)r * 64 + i] != (unsigned char)(r * 7 + i)) return fail(h, MCL_ERR_HIP, "mcl_comm_selftest: the all-gather returned wrong bytes");
        return MCL_OK;
    };
    const int rc = run();
    (void)hipStreamSynchronize(h->stream);
    (void)hipFree(d_buf);
    return rc;
}

int mcl_comm_destroy(mcl_engine_t *h)
{
    if (!h) return MCL_ERR_INVALID_ARG;
    if (h->comm) {
        (void)hipSetDevice(h->cfg.device);
        (void)hipStreamSynchronize(h->stream);
        comm_free(h->comm);
        h->comm = nullptr;
    }
    return MCL_OK;
}

// What a rank found wrong locally during one sharded update (the first thing): its local work stops, its collectives go on.
struct CommLocal {
    int code = MCL_OK;
    std::string msg;
    bool bad() const { return code != MCL_OK; }
    void note(mcl_engine_t *h, int rc) { if (rc != MCL_OK && code == MCL_OK) { code = rc; msg = h->err; } }
};

// MCL_COMM_FAIL: does this rank fail (or stall) before `stage` of this update?
static bool comm_injected(const mcl_comm *c, const char *stage)
{
    return c->fail_rank == c->rank && c->fail_update == (long long)c->updates && c->fail_stage == stage;
}

// rays -> local max -> all-reduce MAX -> weights, scan, list -> this shard's part of the sums -> all-reduce SUM (with the error
// word) -> pinned host.  Returns a HARD failure, or MCL_OK with *failed_ranks = the summed error word (0: the sums are good).
static int comm_rays_to_sums(mcl_engine_t *h, const float *obs, int32_t n_beams, bool sync_rays, CommLocal &loc, double *failed_ranks)
{
    mcl_comm *c = h->comm;
    RcclApi &api = rccl_api();
    const size_t k = 5 + 3 * (size_t)c->n_ranks + 2;
    if (!loc.bad() && comm_injected(c, "rays")) loc.note(h, fail(h, MCL_ERR_HIP, "injected failure before the ray stage (MCL_COMM_FAIL)"));
    if (!loc.bad()) loc.note(h, stage_rays_launch(h, obs, n_beams, false, sync_rays ? nullptr : c->d_red));
    if (!loc.bad() && sync_rays) {        // after an overflow: wait, let the synchronous stage fall back to the self-contained kernel
        loc.note(h, stage_rays_finish(h, obs, n_beams));
        c->host_waits += 1;
        if (!loc.bad()) hipLaunchKernelGGL(mcl::k_copy_double, dim3(1), dim3(1), 0, h->stream, h->d_scalars, c->d_red);
    }
    if (loc.bad()) hipLaunchKernelGGL(mcl::k_set_double, dim3(1), dim3(1), 0, h->stream, c->d_red, -INFINITY);      // (any finite-or-not value will do: the update is void)
    NCCLCHK(h, api.AllReduce(c->d_red, c->d_red, 1, ncclDouble, ncclMax, c->comm, h->stream));
    if (!loc.bad() && comm_injected(c, "weights")) loc.note(h, fail(h, MCL_ERR_HIP, "injected failure before the weights stage (MCL_COMM_FAIL)"));
    if (!loc.bad()) loc.note(h, stage_weights_launch(h, 0.0, c->d_red));
    if (!loc.bad()) {
        hipLaunchKernelGGL(mcl::k_stage_pack, dim3(1), dim3(64), 0, h->stream, h->d_result, c->d_red + 1, c->n_ranks, c->rank, h->compact_pending ? 1 : 0,
                           (unsigned long long)h->compact_cap);
        if (hipGetLastError() != hipSuccess) loc.note(h, fail(h, MCL_ERR_HIP, "k_stage_pack launch failed"));
    }
    if (loc.bad() && hipMemsetAsync(c->d_red + 1, 0, k * 8, h->stream) != hipSuccess) { comm_abort(h, "memset"); return fail(h, MCL_ERR_HIP, "hipMemsetAsync failed (communicator aborted)"); }
    if (comm_injected(c, "stall"))        // test switch: this rank's stream is busy for ~3 x the bound before the last collective (a peer that does not arrive)
        hipLaunchKernelGGL(mcl::k_spin_ms, dim3(1), dim3(64), 0, h->stream, std::min(3.0 * c->timeout_ms, 2000.0));
    hipLaunchKernelGGL(mcl::k_set_double, dim3(1), dim3(1), 0, h->stream, c->d_red + 1 + k, loc.bad() ? 1.0 : 0.0);
    NCCLCHK(h, api.AllReduce(c->d_red + 1, c->d_red + 1, k + 1, ncclDouble, ncclSum, c->comm, h->stream));
    if (hipMemcpyAsync(c->h_red, c->d_red, (1 + k + 1) * 8, hipMemcpyDeviceToHost, h->stream) != hipSuccess) { comm_abort(h, "memcpy"); return fail(h, MCL_ERR_HIP, "hipMemcpyAsync failed (communicator aborted)"); }
    const int rw = comm_wait(h);                                      // THE host wait of the update
    if (rw) return rw;
    *failed_ranks = c->h_red[1 + k];
    if (*failed_ranks != 0.0 || loc.bad()) return MCL_OK;
    unpack_result(h);
    stage_rays_note(h);
    h->timings[4] = elapsed(h->ev[EV_RAYS], h->ev[EV_SENSOR]);
    loc.note(h, layout_adopt(h, h->N));                               // (after the last collective: a failure here is this rank's alone)
    return MCL_OK;
}

// export of this shard's list + all-gather of the chunks, on the engine's stream
static int comm_gather_lists(mcl_engine_t *h, const int64_t *counts, CommLocal &loc)
{
    mcl_comm *c = h->comm;
    RcclApi &api = rccl_api();
    const int G = c->n_ranks;
    int64_t longest = 0, listed = 0;
    for (int r = 0; r < G; ++r) { longest = std::max(longest, counts[r]); listed += counts[r]; }
    const int64_t entries = std::max<int64_t>(64, (longest + 63) & ~(int64_t)63);
    c->gathered = false;
    if ((size_t)entries > c->chunk_capacity) {
        // (without buffers of the agreed size this rank cannot take part in the all-gather: a hard failure)
        bool ok = hipStreamSynchronize(h->stream) == hipSuccess;
        if (c->d_chunk_local) (void)hipFree(c->d_chunk_local);
        if (c->d_chunk_all) (void)hipFree(c->d_chunk_all);
        c->d_chunk_local = c->d_chunk_all = nullptr; c->chunk_capacity = 0;
        const size_t cap = (size_t)entries + (size_t)entries / 4;            // lists breathe from update to update
        ok = ok && hipMalloc(&c->d_chunk_local, cap * 44) == hipSuccess && hipMalloc(&c->d_chunk_all, cap * 44 * (size_t)G) == hipSuccess;
        if (!ok) { comm_abort(h, "alloc"); return fail(h, MCL_ERR_HIP, "mcl_comm_update: no memory for the list exchange (communicator aborted)"); }
        c->chunk_capacity = cap;
    }
    if (!loc.bad()) loc.note(h, export_compact_launch(h, c->d_chunk_local, entries, h->cfg.device, h->stream));
    NCCLCHK(h, api.AllGather(c->d_chunk_local, c->d_chunk_all, (size_t)entries * 44, ncclChar, c->comm, h->stream));
    c->bytes_received = (uint64_t)entries * 44u * (uint64_t)(G - 1);
    c->bytes_payload = (uint64_t)(listed - counts[c->rank]) * 44u;
    c->gathered = !loc.bad(); c->gathered_epoch = h->list_epoch; c->gathered_entries = entries;
    for (int r = 0; r < G; ++r) c->gathered_counts[r] = counts[r];
    return MCL_OK;
}

// the shards' list lengths and weight totals out of a summed vector
static void comm_note_lists(mcl_comm *c, const double *vec)
{
    const int G = c->n_ranks;
    for (int r = 0; r < G; ++r) {
        c->counts[r] = (int64_t)vec[5 + 3 * r] - 1;
        c->totals[r] = ((uint64_t)vec[6 + 3 * r] + ((uint64_t)vec[7 + 3 * r] << 32));      // exact: halves < 2^32
    }
    c->lists_known = true;
}

int mcl_comm_set_lists(mcl_engine_t *h, const int64_t *counts, const uint64_t *totals)
{
    if (!h || !h->comm || !counts || !totals) return MCL_ERR_INVALID_ARG;
    mcl_comm *c = h->comm;
    for (int r = 0; r < c->n_ranks; ++r) { c->counts[r] = counts[r]; c->totals[r] = totals[r]; }
    c->lists_known = true;
    return MCL_OK;
}

// An update without lists (the first after the particles were set or initialised, or a shard whose list outgrew its arrays):
// every shard's fixed-point weights and packed records are gathered whole (8 + 32 B per particle of the other shards), every
// rank scans the same global CDF and draws its own children from it -- the same thresholds as every other path.
// (A set without any weight is noted as this rank's failure -- on every rank, they read the same total -- and the update goes
//  on through its collectives like any other void update.)
static int comm_resample_dense(mcl_engine_t *h, const double action[3], CommLocal &loc)
{
    mcl_comm *c = h->comm;
    RcclApi &api = rccl_api();
    const int G = c->n_ranks;
    const int64_t n = h->N, nt = n * G;
    if ((size_t)nt > c->dense_capacity) {
        bool ok = hipStreamSynchronize(h->stream) == hipSuccess;
        if (c->d_qall) (void)hipFree(c->d_qall);
        if (c->d_cdfall) (void)hipFree(c->d_cdfall);
        if (c->d_recall) (void)hipFree(c->d_recall);
        c->d_qall = c->d_cdfall = nullptr; c->d_recall = nullptr; c->dense_capacity = 0;
        ok = ok && hipMalloc(&c->d_qall, (size_t)nt * 8) == hipSuccess && hipMalloc(&c->d_cdfall, (size_t)nt * 8) == hipSuccess &&
             hipMalloc(&c->d_recall, (size_t)nt * sizeof(double4)) == hipSuccess;
        if (!ok) { comm_abort(h, "alloc"); return fail(h, MCL_ERR_HIP, "mcl_comm_update: no memory for the dense exchange (communicator aborted)"); }
        c->dense_capacity = (size_t)nt;
    }
    if (!loc.bad() && (size_t)nt / mcl::kScanTile + 2 > h->blocktot_capacity) {          // spine scratch of the scan, sized for one shard so far
        graph_reset(h);
        bool ok = hipStreamSynchronize(h->stream) == hipSuccess;
        dfree(h->d_blocktot);
        h->blocktot_capacity = 0;
        ok = ok && hipMalloc(&h->d_blocktot, ((size_t)nt / mcl::kScanTile + 2) * 8) == hipSuccess;
        if (ok) h->blocktot_capacity = (size_t)nt / mcl::kScanTile + 2;
        else loc.note(h, fail(h, MCL_ERR_HIP, "mcl_comm_update: no memory for the scan of the whole set"));
    }
    const int cur = h->cur;
    if (!loc.bad() && !h->pack_valid[cur]) {
        hipLaunchKernelGGL(mcl::k_pack_records, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, h->d_x[cur], h->d_y[cur], h->d_th[cur], n,
                           h->d_pack[cur]);
        h->pack_valid[cur] = true;
    }
    NCCLCHK(h, api.AllGather(h->d_q, c->d_qall, (size_t)n, ncclUint64, c->comm, h->stream));
    NCCLCHK(h, api.AllGather(h->d_pack[cur], c->d_recall, (size_t)n * sizeof(double4), ncclChar, c->comm, h->stream));
    if (!loc.bad()) loc.note(h, scan_weights(h, c->d_qall, c->d_cdfall, nt, 0, nullptr));
    // the draw needs the global fixed-point total on the host (a launch argument): one more wait, in an update that has no lists
    uint64_t q_total = 0;
    if (!loc.bad() && hipMemcpyAsync(&c->h_red[0], c->d_cdfall + (nt - 1), 8, hipMemcpyDeviceToHost, h->stream) != hipSuccess)
        loc.note(h, fail(h, MCL_ERR_HIP, "hipMemcpyAsync of the weight total failed"));
    const int rw = comm_wait(h);
    if (rw) return rw;
    if (loc.bad()) return MCL_OK;
    std::memcpy(&q_total, &c->h_red[0], 8);
    if (q_total == 0) { loc.note(h, fail(h, MCL_ERR_NOT_READY, "the particle set carries no weight")); return MCL_OK; }
    ParentSource src;
    src.records = c->d_recall;
    c->last_dense = true;
    c->dense_weights_bytes = (uint64_t)n * 8u * (uint64_t)(G - 1);
    c->dense_records_bytes = (uint64_t)n * 32u * (uint64_t)(G - 1);
    if (comm_injected(c, "resample")) { loc.note(h, fail(h, MCL_ERR_HIP, "injected failure before the resampling stage (MCL_COMM_FAIL)")); return MCL_OK; }
    loc.note(h, stage_resample_launch(h, src, c->d_cdfall, nt, q_total, (int64_t)c->rank * n, nt, action));
    return MCL_OK;
}

int mcl_comm_update(mcl_engine_t *h, const double action[3], const float *obs, int32_t n_beams, double pose_out[3])
{
    if (!h || !action || !obs || !pose_out) return MCL_ERR_INVALID_ARG;
    mcl_comm *c = h->comm;
    // (the checks up to here must come out the same on every rank -- a host that calls with different arguments on different
    //  ranks has a bug no protocol repairs; what CAN differ between ranks goes through CommLocal below)
    if (!c) return fail(h, MCL_ERR_NOT_READY, "mcl_comm_create first");
    if (c->dead) return fail(h, MCL_ERR_NOT_READY, "the communicator was aborted: mcl_comm_create again");
    if (n_beams != h->B) return fail(h, MCL_ERR_INVALID_ARG, "bad observation");
    const int G = c->n_ranks;
    const int64_t n_per_shard = h->N;
    if (n_per_shard * G >= MCL_MAX_TOTAL_PARTICLES) return fail(h, MCL_ERR_INVALID_ARG, "particle total must stay below 2^27");
    const auto t0 = std::chrono::steady_clock::now();
    HIPCHK(h, hipSetDevice(h->cfg.device));
    c->host_waits = 0;
    c->last_dense = false;
    c->updates += 1;
    CommLocal loc;
    // "particles not set on this rank" is a local condition: it goes through the protocol like any other
    if (!ready(h, true)) loc.note(h, fail(h, MCL_ERR_NOT_READY, "map, beam angles and particles must be set first"));
    const int64_t *counts = c->counts;
    const uint64_t *totals = c->totals;
    // Lists or not is decided from what EVERY rank knows alike (the previous update's summed vector): all ranks take the same branch.
    bool lists = c->lists_known && !h->env_comm_no_lists;
    uint64_t weight = 0;
    for (int r = 0; r < G && lists; ++r) {
        lists = counts[r] >= 0;
        weight += counts[r] > 0 ? totals[r] : 0ull;
    }
    lists = lists && weight != 0;
    int rc;
    // Adaptive resampling (E9): the set is kept when the effective sample size of the WHOLE set (the previous update's summed
    // vector: every rank has the same numbers) is at least r / 1000 of it -- no exchange at all then
    bool keep = false;
    if (h->cfg.resample_neff_permille > 0 && c->vec_valid && h->carry_valid) {
        const double sw = c->vec[0], sww = c->vec[5 + 3 * G + 1];
        keep = sww > 0.0 && sw * sw >= ((double)h->cfg.resample_neff_permille / 1000.0) * (double)(n_per_shard * G) * sww;
    }
    c->last_kept = keep;
    if (keep) {
        c->gathered = false;
        c->lists_known = false;
        if (!loc.bad()) loc.note(h, mcl_stage_keep(h, (int64_t)c->rank * n_per_shard, n_per_shard * G, action));
    } else if (lists) {
        if (!loc.bad() && counts[c->rank] != h->compact_n)
            loc.note(h, fail(h, MCL_ERR_NOT_READY, "this engine's list is not the one the last exchange described (state changed on one rank only?)"));
        // (1) the lists: already here when the previous update gathered them (same lists, same lengths), else now.  Whether they
        // were pre-gathered is the same on every rank: every rank gathers after a good update and forgets after a void one.
        bool have = c->gathered_entries > 0 && c->gathered_epoch_all == c->updates - 1;
        for (int r = 0; r < G && have; ++r) have = c->gathered_counts[r] == counts[r];
        if (have && !(c->gathered && c->gathered_epoch == h->list_epoch) && !loc.bad())
            loc.note(h, fail(h, MCL_ERR_NOT_READY, "the gathered lists are stale on this rank (particle state changed on one rank only?)"));
        if (!have) { rc = comm_gather_lists(h, counts, loc); if (rc) return rc; }
        const int64_t entries = c->gathered_entries;
        c->gathered = false;
        c->lists_known = false;                // (known again once this update's vector is here)
        if (!loc.bad() && comm_injected(c, "resample")) loc.note(h, fail(h, MCL_ERR_HIP, "injected failure before the resampling stage (MCL_COMM_FAIL)"));
        if (!loc.bad())
            loc.note(h, stage_resample_compact_launch(h, c->d_chunk_all, G, entries, counts, totals, n_per_shard, c->rank, (int64_t)c->rank * n_per_shard,
                                                      n_per_shard * G, action, nullptr));
    } else {
        c->gathered = false;
        c->lists_known = false;
        rc = comm_resample_dense(h, action, loc);
        if (rc) return rc;
    }
    // (2) + (3)
    double failed = 0.0;
    rc = comm_rays_to_sums(h, obs, n_beams, false, loc, &failed);
    if (rc) return rc;
    const size_t k = 5 + 3 * (size_t)G + 2;
    if (failed == 0.0 && !loc.bad() && c->h_red[1 + k - 2] != 0.0) {
        // some shard's fix-up lists overflowed (debug_force_exact at size, a pathological map): every rank once more from the ray stage on
        rc = comm_rays_to_sums(h, obs, n_beams, true, loc, &failed);
        if (rc) return rc;
    }
    if (failed != 0.0 || loc.bad()) {
        // a void update, on every rank alike: the lists and sums are forgotten, the particle set must be set or initialised again
        comm_forget(c);
        c->gathered_entries = 0;
        h->have_particles = false;
        if (loc.bad()) return fail(h, loc.code, loc.msg + " [sharded update void on every rank]");
        char m[160];
        std::snprintf(m, sizeof m, "%d rank(s) of the sharded set reported a failure in this update: it is void on every rank (set or initialise the particles again)", (int)failed);
        return fail(h, MCL_ERR_PEER, m);
    }
    for (size_t i = 0; i < k; ++i) c->vec[i] = c->h_red[1 + i];
    for (int i = 0; i < 5; ++i) h->global_sums[i] = c->vec[i];
    c->vec_valid = true;
    stage_commit_carry(h);
    comm_note_lists(c, c->vec);
    bool next_keeps = false;
    if (h->cfg.resample_neff_permille > 0) {
        const double sw = c->vec[0], sww = c->vec[5 + 3 * G + 1];
        next_keeps = sww > 0.0 && sw * sw >= ((double)h->cfg.resample_neff_permille / 1000.0) * (double)(n_per_shard * G) * sww;
    }
    // the lists this update wrote are final: gather them for the next update now, beside the host's work between updates
    // (every rank reads the same vector, so every rank takes the same decision)
    c->gathered_entries = 0;
    {
        bool all = true;
        uint64_t wsum = 0;
        for (int r = 0; r < G; ++r) { all = all && c->counts[r] >= 0; wsum |= c->counts[r] > 0 ? c->totals[r] : 0ull; }
        if (all && wsum != 0 && !h->env_comm_no_pregather && !next_keeps) {
            CommLocal pre;                         // (a local failure of the export shows in the next update: `gathered` stays false here)
            rc = comm_gather_lists(h, c->counts, pre);
            if (rc) return rc;
            c->gathered_epoch_all = c->updates;
        }
    }
    const double sw = c->vec[0], kk = sw > 0.0 ? 1.0 / sw : 1.0;          // expected_pose (cpp:702-716) over the whole set
    pose_out[0] = c->vec[1] * kk; pose_out[1] = c->vec[2] * kk; pose_out[2] = std::atan2(c->vec[3] * kk, c->vec[4] * kk);
    h->timings[5] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return MCL_OK;
}

int mcl_comm_get_vector(const mcl_engine_t *h, double *vec_out, int32_t n)
{
    if (!h || !h->comm || !vec_out || n != 5 + 3 * h->comm->n_ranks + 2) return MCL_ERR_INVALID_ARG;
    for (int i = 0; i < n; ++i) vec_out[i] = h->comm->vec[i];
    return MCL_OK;
}

int mcl_comm_last_exchange(const mcl_engine_t *h, int32_t *dense, uint64_t *weights_bytes, uint64_t *records_bytes)
{
    if (!h || !h->comm) return MCL_ERR_INVALID_ARG;
    if (dense) *dense = h->comm->last_kept ? 2 : h->comm->last_dense ? 1 : 0;       // 0 lists, 1 dense, 2 none (the set was kept)
    if (weights_bytes) *weights_bytes = h->comm->last_dense ? h->comm->dense_weights_bytes : 0;
    if (records_bytes) *records_bytes = h->comm->last_dense ? h->comm->dense_records_bytes : 0;
    return MCL_OK;
}

int mcl_comm_stats(const mcl_engine_t *h, uint64_t *list_bytes_received, uint64_t *list_payload_bytes, int32_t *host_waits)
{
    if (!h || !h->comm) return MCL_ERR_INVALID_ARG;
    if (list_bytes_received) *list_bytes_received = h->comm->bytes_received;
    if (list_payload_bytes) *list_payload_bytes = h->comm->bytes_payload;
    if (host_waits) *host_waits = h->comm->host_waits;
    return MCL_OK;
}

int mcl_scan_weights(mcl_engine_t *h, const uint64_t *d_q, uint64_t *d_cdf, int64_t n, uint64_t offset)
{
    if (!h || !d_q || !d_cdf || n <= 0) return MCL_ERR_INVALID_ARG;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    size_t need = (size_t)n / mcl::kScanTile + 2;
    if (need > h->blocktot_capacity) {   // spine scratch is sized for cap; grow it for gathered arrays
        graph_reset(h);
        dfree(h->d_blocktot);
        HIPCHK(h, hipMalloc(&h->d_blocktot, need * 8));
        h->blocktot_capacity = need;
    }
    int rc = scan_weights(h, d_q, d_cdf, n, offset, nullptr);
    if (rc) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return MCL_OK;
}


// ---------------------------------------------------------------------------------------------
// Several GPUs driven by ONE host process (the reference is a single ROS 2 process, cpp:1019-1025): a group owns one engine
// per device, shards the particles contiguously and runs every update phase on all devices before the next phase starts,
// so the devices work concurrently although one host thread issues the calls.  Per update and device:
//   * the other shards' fixed-point weights arrive by peer copies (8 B per particle of the other shards), every device
//     scans the same exact global CDF and draws its own children;
//   * a child's parent record is read where it lives (peer pointer): only SELECTED parents cross a link, no record is
//     gathered wholesale;
//   * max log-weight and the seven sums are combined on the host (a few doubles per device).
// Results are bit-identical to one engine holding all particles (exact integer CDF, exact fp64 log-weight sums, Philox
// keyed by the global particle index), which tests/test_gpu_group.py checks with two engines on one device.
// ---------------------------------------------------------------------------------------------
struct mcl_group {
    std::vector<mcl_engine *> eng;
    std::vector<uint64_t *> d_qall, d_cdfall;      // per device: all shards' weights and their global CDF
    std::vector<unsigned long long *> d_remote;    // per device, 4 words: children whose parent was fetched from a peer (last update) |
                                                   // this shard's max log-weight | the maximum over the shards (doubles)
    std::vector<unsigned char *> d_chunks;         // per device: every shard's compact parent list (grown on demand)
    std::vector<size_t> chunks_capacity;
    std::vector<hipEvent_t> ev_ready, ev_children, ev_rays; // per device, see mcl_group_update
    bool compact_last = false;
    int64_t n_per = 0, n_total = 0;
    uint64_t q_total = 0;
    bool have_q_total = false;
    double sums[5] = {0, 0, 0, 0, 0};
    double sum_ww = 0.0;                           // sum w^2 of the whole set (adaptive resampling)
    bool kept_last = false;
    double timings[6] = {0, 0, 0, 0, 0, 0};
    uint64_t bytes_weights = 0, bytes_parents = 0;
    std::string err;
};

static int gfail(mcl_group *g, int rc, const std::string &msg)
{
    if (g) g->err = msg;
    return rc;
}

static int group_sync_q_total(mcl_group *g)
{
    // after set_particles / init: every shard has its local fixed-point total on the host (fetch_scalars)
    uint64_t t = 0;
    for (auto *e : g->eng) t += e->q_total;
    g->q_total = t;
    g->have_q_total = true;
    double gs[5] = {0, 0, 0, 0, 0};
    for (auto *e : g->eng)
        for (int k = 0; k < 5; ++k) gs[k] += e->global_sums[k];
    for (int k = 0; k < 5; ++k) g->sums[k] = gs[k];
    for (auto *e : g->eng) mcl_stage_finish(e, gs);
    return MCL_OK;
}

const char *mcl_group_last_error(const mcl_group_t *g) { return g ? g->err.c_str() : g_create_error.c_str(); }

void mcl_group_destroy(mcl_group_t *g)
{
    if (!g) return;
    for (size_t d = 0; d < g->eng.size(); ++d) {
        if (!g->eng[d]) continue;
        (void)hipSetDevice(g->eng[d]->cfg.device);
        if (d < g->d_qall.size() && g->d_qall[d]) (void)hipFree(g->d_qall[d]);
        if (d < g->d_cdfall.size() && g->d_cdfall[d]) (void)hipFree(g->d_cdfall[d]);
        if (d < g->d_remote.size() && g->d_remote[d]) (void)hipFree(g->d_remote[d]);
        if (d < g->d_chunks.size() && g->d_chunks[d]) (void)hipFree(g->d_chunks[d]);
        if (d < g->ev_ready.size() && g->ev_ready[d]) (void)hipEventDestroy(g->ev_ready[d]);
        if (d < g->ev_children.size() && g->ev_children[d]) (void)hipEventDestroy(g->ev_children[d]);
        if (d < g->ev_rays.size() && g->ev_rays[d]) (void)hipEventDestroy(g->ev_rays[d]);
        mcl_destroy(g->eng[d]);
    }
    delete g;
}

int mcl_group_create(const mcl_config_t *cfg, const int32_t *devices, int32_t n_devices, mcl_group_t **out)
{
    g_create_error.clear();
    if (!cfg || !devices || !out || n_devices <= 0 || n_devices > mcl::kMaxShards) { g_create_error = "bad group arguments (1..16 devices)"; return MCL_ERR_INVALID_ARG; }
    *out = nullptr;
    if (cfg->weight_mode != MCL_WEIGHT_LOG) {
        g_create_error = "a device group needs weight_mode LOG";
        return MCL_ERR_UNSUPPORTED;
    }
    if ((int64_t)cfg->max_particles * n_devices >= MCL_MAX_TOTAL_PARTICLES) { g_create_error = "particle total must stay below 2^27"; return MCL_ERR_INVALID_ARG; }
    mcl_group *g = new mcl_group();
    for (int d = 0; d < n_devices; ++d) {
        mcl_config_t c = *cfg;
        c.device = devices[d];
        mcl_engine_t *e = nullptr;
        const int rc = mcl_create(&c, &e);
        if (rc != MCL_OK) { mcl_group_destroy(g); return rc; }
        g->eng.push_back(e);
    }
    g->d_qall.assign(n_devices, nullptr); g->d_cdfall.assign(n_devices, nullptr); g->d_remote.assign(n_devices, nullptr);
    g->d_chunks.assign(n_devices, nullptr); g->chunks_capacity.assign(n_devices, 0);
    g->ev_ready.assign(n_devices, nullptr); g->ev_children.assign(n_devices, nullptr); g->ev_rays.assign(n_devices, nullptr);
    const size_t cap_total = (size_t)cfg->max_particles * n_devices;
    for (int d = 0; d < n_devices; ++d) {
        if (hipSetDevice(devices[d]) != hipSuccess || hipMalloc(&g->d_qall[d], cap_total * 8) != hipSuccess ||
            hipMalloc(&g->d_cdfall[d], cap_total * 8) != hipSuccess || hipMalloc(&g->d_remote[d], 32) != hipSuccess ||
            hipEventCreateWithFlags(&g->ev_ready[d], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&g->ev_rays[d], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&g->ev_children[d], hipEventDisableTiming) != hipSuccess) {
            g_create_error = "group buffers: hipMalloc failed";
            mcl_group_destroy(g);
            return MCL_ERR_HIP;
        }
        // parents are read where they live: peer access to every other device of the group
        for (int o = 0; o < n_devices; ++o) {
            if (devices[o] == devices[d]) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, devices[d], devices[o]) != hipSuccess || !can) {
                g_create_error = "devices of a group must have peer access to each other";
                mcl_group_destroy(g);
                return MCL_ERR_UNSUPPORTED;
            }
            const hipError_t pe = hipDeviceEnablePeerAccess(devices[o], 0);
            if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) { g_create_error = "hipDeviceEnablePeerAccess failed"; mcl_group_destroy(g); return MCL_ERR_HIP; }
            (void)hipGetLastError();
        }
    }
    *out = g;
    return MCL_OK;
}

int32_t mcl_group_size(const mcl_group_t *g) { return g ? (int32_t)g->eng.size() : 0; }

int mcl_group_engine(mcl_group_t *g, int32_t i, mcl_engine_t **out)
{
    if (!g || !out || i < 0 || i >= (int32_t)g->eng.size()) return MCL_ERR_INVALID_ARG;
    *out = g->eng[i];
    return MCL_OK;
}

int mcl_group_set_map(mcl_group_t *g, const int8_t *data, uint32_t width, uint32_t height, float resolution, double origin_x, double origin_y)
{
    if (!g) return MCL_ERR_INVALID_ARG;
    for (auto *e : g->eng) {
        const int rc = mcl_set_map(e, data, width, height, resolution, origin_x, origin_y);
        if (rc) return gfail(g, rc, e->err);
    }
    return MCL_OK;
}

int mcl_group_set_beam_angles(mcl_group_t *g, const float *angles, int32_t n_beams)
{
    if (!g) return MCL_ERR_INVALID_ARG;
    for (auto *e : g->eng) {
        const int rc = mcl_set_beam_angles(e, angles, n_beams);
        if (rc) return gfail(g, rc, e->err);
    }
    return MCL_OK;
}

static int group_check_total(mcl_group *g, int64_t n_total)
{
    const int64_t G = (int64_t)g->eng.size();
    if (n_total <= 0 || n_total % G != 0 || n_total / G > g->eng[0]->cap)
        return gfail(g, MCL_ERR_INVALID_ARG, "the particle total must be a multiple of the device count and fit max_particles per device");
    g->n_total = n_total; g->n_per = n_total / G;
    return MCL_OK;
}

int mcl_group_set_particles(mcl_group_t *g, const double *xyz, const double *weights, int64_t n_total)
{
    if (!g || !xyz || !weights) return MCL_ERR_INVALID_ARG;
    int rc = group_check_total(g, n_total);
    if (rc) return rc;
    // all shards quantise their weights against the same scale: the maximum over the whole set, which is what a single
    // engine holding all particles would use
    double wmax = 0.0;
    for (int64_t i = 0; i < n_total; ++i) wmax = std::max(wmax, weights[i]);
    if (!(wmax > 0.0)) return gfail(g, MCL_ERR_INVALID_ARG, "weights must have a positive maximum");
    std::vector<double> shard((size_t)g->n_per * 3);
    for (size_t d = 0; d < g->eng.size(); ++d) {
        for (int c = 0; c < 3; ++c)
            std::memcpy(shard.data() + (size_t)c * g->n_per, xyz + (size_t)c * n_total + d * (size_t)g->n_per, (size_t)g->n_per * 8);
        rc = set_particles_impl(g->eng[d], shard.data(), weights + d * (size_t)g->n_per, g->n_per, &wmax);
        if (rc) return gfail(g, rc, g->eng[d]->err);
    }
    return group_sync_q_total(g);
}

int mcl_group_init_particles_pose(mcl_group_t *g, const double pose[3], int64_t n_total)
{
    if (!g || !pose) return MCL_ERR_INVALID_ARG;
    int rc = group_check_total(g, n_total);
    if (rc) return rc;
    for (size_t d = 0; d < g->eng.size(); ++d) {
        rc = mcl_init_particles_pose(g->eng[d], pose, g->n_per, (int64_t)d * g->n_per, n_total);
        if (rc) return gfail(g, rc, g->eng[d]->err);
    }
    return group_sync_q_total(g);
}

int mcl_group_init_global(mcl_group_t *g, int64_t n_total)
{
    if (!g) return MCL_ERR_INVALID_ARG;
    int rc = group_check_total(g, n_total);
    if (rc) return rc;
    for (size_t d = 0; d < g->eng.size(); ++d) {
        rc = mcl_init_global(g->eng[d], g->n_per, (int64_t)d * g->n_per, n_total);
        if (rc) return gfail(g, rc, g->eng[d]->err);
    }
    return group_sync_q_total(g);
}

#define GHIP(g, call)                                                                            \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess) return gfail(g, MCL_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

int mcl_group_update(mcl_group_t *g, const double action[3], const float *obs, int32_t n_beams)
{
    if (!g || !action || !obs) return MCL_ERR_INVALID_ARG;
    if (g->n_per <= 0 || !g->have_q_total) return gfail(g, MCL_ERR_NOT_READY, "particles not set");
    const auto t0 = std::chrono::steady_clock::now();
    const int G = (int)g->eng.size();
    const int64_t n = g->n_per, nt = g->n_total;
    // Phases are ordered by EVENTS between the devices' streams, not by host-side synchronisation: ev_ready[s] = shard s's
    // parent data (records or compact list, weights) may be read by its peers; ev_children[d] = device d has drawn its children
    // and no longer reads anybody's parent data.  The host waits only where it needs a value (the maxima, the sums).
    // Exchange: when every shard has a compact parent list (the usual case after an update with many beams) the devices copy
    // each other's LISTS (44 B per particle that carries weight); otherwise every weight (8 B per particle) and the
    // selected parents are read where they live.
    // Adaptive resampling (E9, cfg.resample_neff_permille > 0): the set is kept -- no exchange, no resampling -- when the effective
    // sample size of the WHOLE set after the previous update is at least r / 1000 of it (mcl_update's rule on the group's sums)
    bool keep = false;
    {
        const int r = g->eng[0]->cfg.resample_neff_permille;
        bool carry = r > 0;
        for (int s = 0; s < G; ++s) carry = carry && g->eng[s]->carry_valid;
        if (carry) keep = g->sum_ww > 0.0 && g->sums[0] * g->sums[0] >= ((double)r / 1000.0) * (double)nt * g->sum_ww;
    }
    bool compact = g->q_total != 0 && !keep;
    int64_t longest = 0;
    for (int s = 0; s < G; ++s) { compact = compact && g->eng[s]->compact_n >= 0; longest = std::max(longest, g->eng[s]->compact_n); }
    const int64_t centries = std::max<int64_t>(64, (longest + 63) & ~(int64_t)63);
    for (int s = 0; s < G; ++s) {
        mcl_engine *e = g->eng[s];
        GHIP(g, hipSetDevice(e->cfg.device));
        if (!compact && !keep && !e->pack_valid[e->cur]) {      // first update after set_particles / init: the records do not exist yet
            hipLaunchKernelGGL(mcl::k_pack_records, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, e->stream, e->d_x[e->cur], e->d_y[e->cur],
                               e->d_th[e->cur], n, e->d_pack[e->cur]);
            e->pack_valid[e->cur] = true;
        }
        GHIP(g, hipEventRecord(g->ev_ready[s], e->stream));
    }
    const double4 *parents[mcl::kMaxShards] = {};      // the launch below flips an engine's current buffer: take the pointers first
    for (int s = 0; s < G; ++s) parents[s] = g->eng[s]->d_pack[g->eng[s]->cur];
    int64_t counts[mcl::kMaxShards] = {};
    uint64_t totals[mcl::kMaxShards] = {};
    for (int s = 0; s < G; ++s) { counts[s] = g->eng[s]->compact_n; totals[s] = g->eng[s]->q_total; }
    for (int d = 0; d < G; ++d) {
        mcl_engine *e = g->eng[d];
        GHIP(g, hipSetDevice(e->cfg.device));
        for (int s = 0; s < G; ++s)
            if (s != d) GHIP(g, hipStreamWaitEvent(e->stream, g->ev_ready[s], 0));
        GHIP(g, hipMemsetAsync(g->d_remote[d], 0, 8, e->stream));
        int rc;
        if (keep) {
            rc = mcl_stage_keep(e, (int64_t)d * n, nt, action);
        } else if (compact) {
            const size_t need = (size_t)G * (size_t)centries * 44;
            if (need > g->chunks_capacity[d]) {
                if (g->d_chunks[d]) { GHIP(g, hipStreamSynchronize(e->stream)); (void)hipFree(g->d_chunks[d]); g->d_chunks[d] = nullptr; }
                g->chunks_capacity[d] = 0;
                GHIP(g, hipMalloc(&g->d_chunks[d], need));
                g->chunks_capacity[d] = need;
            }
            for (int s = 0; s < G; ++s) {
                rc = export_compact_launch(g->eng[s], g->d_chunks[d] + (size_t)s * (size_t)centries * 44, centries, e->cfg.device, e->stream);
                if (rc) return gfail(g, rc, g->eng[s]->err);
            }
            rc = stage_resample_compact_launch(e, g->d_chunks[d], G, centries, counts, totals, n, d, (int64_t)d * n, nt, action, g->d_remote[d]);
        } else {
            for (int s = 0; s < G; ++s)
                GHIP(g, hipMemcpyPeerAsync(g->d_qall[d] + (size_t)s * n, e->cfg.device, g->eng[s]->d_q, g->eng[s]->cfg.device, (size_t)n * 8, e->stream));
            if ((size_t)nt / mcl::kScanTile + 2 > e->blocktot_capacity) {
                graph_reset(e);                // a captured update graph of this engine holds the old pointer
                GHIP(g, hipStreamSynchronize(e->stream));
                dfree(e->d_blocktot);
                GHIP(g, hipMalloc(&e->d_blocktot, ((size_t)nt / mcl::kScanTile + 2) * 8));
                e->blocktot_capacity = (size_t)nt / mcl::kScanTile + 2;
            }
            rc = scan_weights(e, g->d_qall[d], g->d_cdfall[d], nt, 0, nullptr);
            if (rc) return gfail(g, rc, e->err);
            ParentSource src;
            for (int s = 0; s < G; ++s) src.rank_records[s] = parents[s];
            src.n_per_rank = n; src.self_rank = d; src.remote_count = g->d_remote[d];
            rc = stage_resample_launch(e, src, g->d_cdfall[d], nt, g->q_total, (int64_t)d * n, nt, action);
        }
        if (rc) return gfail(g, rc, e->err);
        GHIP(g, hipEventRecord(g->ev_children[d], e->stream));
    }
    // phase 2: rays + likelihood on every device; the global maximum is taken ON the devices (every device reads the peers'
    // local maxima once their ray stages have finished: events, no host wait)
    // phase 3: weights against the global maximum, sums.  The weights (and the compact list) of a shard are rewritten here:
    // every device must have drawn its children first.  The host waits once, for the sums.
    double gs[5] = {0, 0, 0, 0, 0};
    double sww = 0.0;
    uint64_t qt = 0;
    unsigned long long remote = 0;
    uint64_t listed = 0;
    for (int pass = 0; pass < 2; ++pass) {
        const bool redo = pass == 1;          // only after a fix-up list overflow: the synchronous ray stage falls back by itself
        for (int d = 0; d < G; ++d) {
            mcl_engine *e = g->eng[d];
            double *lmax = reinterpret_cast<double *>(g->d_remote[d] + 1);
            int rc = stage_rays_launch(e, obs, n_beams, false, redo ? nullptr : lmax);
            if (!rc && redo) {
                rc = stage_rays_finish(e, obs, n_beams);             // (waits; relaunches with k_rays_skip after an overflow)
                if (!rc) hipLaunchKernelGGL(mcl::k_copy_double, dim3(1), dim3(1), 0, e->stream, e->d_scalars, lmax);
            }
            if (rc) return gfail(g, rc, e->err);
            GHIP(g, hipEventRecord(g->ev_rays[d], e->stream));
        }
        for (int d = 0; d < G; ++d) {
            mcl_engine *e = g->eng[d];
            GHIP(g, hipSetDevice(e->cfg.device));
            mcl::GroupMaxArgs ma{};
            for (int o = 0; o < G; ++o) {
                if (o != d) {
                    GHIP(g, hipStreamWaitEvent(e->stream, g->ev_rays[o], 0));
                    GHIP(g, hipStreamWaitEvent(e->stream, g->ev_children[o], 0));
                }
                ma.src[o] = reinterpret_cast<const double *>(g->d_remote[o] + 1);
            }
            ma.n = G; ma.out = reinterpret_cast<double *>(g->d_remote[d] + 2);
            hipLaunchKernelGGL(mcl::k_group_max, dim3(1), dim3(1), 0, e->stream, ma);
            const int rc = stage_weights_launch(e, 0.0, ma.out);
            if (rc) return gfail(g, rc, e->err);
        }
        for (int k = 0; k < 5; ++k) gs[k] = 0.0;
        sww = 0.0;
        qt = 0; remote = 0; listed = 0;
        bool overflow = false;
        for (int d = 0; d < G; ++d) {
            mcl_engine *e = g->eng[d];
            int rc = stage_weights_finish(e);                         // THE host wait of this device's update
            if (rc) return gfail(g, rc, e->err);
            stage_rays_note(e);
            overflow = overflow || (e->last_quad && e->h_fix_count != 0);
            for (int k = 0; k < 5; ++k) gs[k] += e->global_sums[k];      // unpack_result left the LOCAL sums there
            sww += e->h_scalars[7];
            qt += e->q_total;
            unsigned long long r = 0;
            GHIP(g, hipMemcpy(&r, g->d_remote[d], 8, hipMemcpyDeviceToHost));
            remote += r;
            if (compact) listed += (uint64_t)counts[d];
        }
        if (!overflow) break;
        if (redo) return gfail(g, MCL_ERR_HIP, "the ray stage's work lists overflowed twice (internal)");
    }
    for (int k = 0; k < 5; ++k) g->sums[k] = gs[k];
    g->sum_ww = sww;
    g->kept_last = keep;
    g->q_total = qt;
    for (int d = 0; d < G; ++d) mcl_stage_finish(g->eng[d], gs);
    // received per device: the other shards' lists / weights; parents read from peers (dense exchange only)
    // (lists are copied entry-exact, not as padded chunks: the device that holds the shortest list receives the most)
    uint64_t shortest = ~0ull;
    for (int d = 0; d < G; ++d) shortest = std::min<uint64_t>(shortest, compact ? (uint64_t)counts[d] : 0u);
    g->bytes_weights = compact ? (listed - shortest) * 44u : (uint64_t)(G - 1) * (uint64_t)n * 8u;
    g->bytes_parents = compact ? 0u : (uint64_t)remote * 32u;         // upper bound: children of remote parents x record size
    if (keep) { g->bytes_weights = 0; g->bytes_parents = 0; }         // nothing was exchanged
    g->compact_last = compact;
    for (int k = 0; k < 5; ++k) {
        double m = 0.0;
        for (int d = 0; d < G; ++d) m = std::max(m, g->eng[d]->timings[k]);
        g->timings[k] = m;
    }
    g->timings[5] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return MCL_OK;
}

int mcl_group_expected_pose(mcl_group_t *g, double out[3])
{
    if (!g || !out) return MCL_ERR_INVALID_ARG;
    if (g->n_per <= 0) return MCL_ERR_NOT_READY;
    const double s = g->sums[0];
    const double k = (s > 0.0) ? 1.0 / s : 1.0;
    out[0] = g->sums[1] * k;
    out[1] = g->sums[2] * k;
    out[2] = std::atan2(g->sums[3] * k, g->sums[4] * k);
    return MCL_OK;
}

int mcl_group_get_particles(mcl_group_t *g, double *xyz, int64_t n_total)
{
    if (!g || !xyz || n_total != g->n_total || g->n_per <= 0) return MCL_ERR_INVALID_ARG;
    std::vector<double> shard((size_t)g->n_per * 3);
    for (size_t d = 0; d < g->eng.size(); ++d) {
        const int rc = mcl_get_particles(g->eng[d], shard.data(), g->n_per);
        if (rc) return gfail(g, rc, g->eng[d]->err);
        for (int c = 0; c < 3; ++c)
            std::memcpy(xyz + (size_t)c * n_total + d * (size_t)g->n_per, shard.data() + (size_t)c * g->n_per, (size_t)g->n_per * 8);
    }
    return MCL_OK;
}

int mcl_group_get_weights(mcl_group_t *g, double *weights, int64_t n_total)
{
    if (!g || !weights || n_total != g->n_total || g->n_per <= 0) return MCL_ERR_INVALID_ARG;
    for (size_t d = 0; d < g->eng.size(); ++d) {
        const int rc = mcl_get_weights(g->eng[d], weights + d * (size_t)g->n_per, g->n_per);
        if (rc) return gfail(g, rc, g->eng[d]->err);
    }
    return MCL_OK;
}

int mcl_group_get_resample_indices(mcl_group_t *g, int32_t *idx, int64_t n_total)
{
    if (!g || !idx || n_total != g->n_total || g->n_per <= 0) return MCL_ERR_INVALID_ARG;
    for (size_t d = 0; d < g->eng.size(); ++d) {
        const int rc = mcl_get_resample_indices(g->eng[d], idx + d * (size_t)g->n_per, g->n_per);
        if (rc) return gfail(g, rc, g->eng[d]->err);
    }
    return MCL_OK;
}

int mcl_group_get_stage_timings(const mcl_group_t *g, double ms[6])
{
    if (!g || !ms) return MCL_ERR_INVALID_ARG;
    std::memcpy(ms, g->timings, sizeof(g->timings));
    return MCL_OK;
}

int mcl_group_exchange_bytes(const mcl_group_t *g, uint64_t out[2])
{
    if (!g || !out) return MCL_ERR_INVALID_ARG;
    out[0] = g->bytes_weights; out[1] = g->bytes_parents;
    return MCL_OK;
}

int32_t mcl_group_exchanged_lists(const mcl_group_t *g) { return g && g->compact_last ? 1 : 0; }

}  // extern "C"
