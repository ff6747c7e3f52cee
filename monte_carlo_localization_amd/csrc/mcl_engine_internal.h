// mcl_engine_internal.h -- what the translation units of libmcl_hip_engine.so share: the engine object, the status helpers, and
// the handful of engine-side functions the sharded hosts are built on.  Not part of the ABI (include/mcl_hip_engine.h is).
//   mcl_engine.hip   the engine: map / beams / particles, the update and its stages, every kernel
//   mcl_comm.hip     mcl_comm_*: one process per GPU, the collectives of an update over RCCL on the engine's stream
//   mcl_group.hip    mcl_group_*: several GPUs behind one handle, driven by one process (peer copies)
// Only mcl_engine.hip includes the kernels (mcl_kernels.h); the other two reach the few kernels they launch through the
// launch_* functions below.
#pragma once
#include "../../include/mcl_hip_engine.h"

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdint>
#include <string>
#include <vector>

#include "mcl_types.h"

enum { EV_START = 0, EV_RESAMPLE, EV_QUERY, EV_RAYS, EV_SENSOR, EV_K0, EV_K1, EV_COUNT };   // K0..K1 bracket the dominant kernel


constexpr unsigned long long kExactCap = 1ull << 16;   // level-3 rays per launch handled by k_rays_exact (more: inline)
// d_result / h_result: [0..7] scalars, [8..11] counters, [12] work-list overflow flag, [13] work counter, [14] level-3 list
// length, [15] far-list length (and the staging word of a global maximum), [16] length of the compact parent list
constexpr int kResultWords = 17;
constexpr int kResultStage = 40;             // h_result word that stages a host value on its way to the device
constexpr int kResultStamp = 32;             // h_result word a small update's last kernel stamps (the host polls it)

struct mcl_comm;
void comm_free(struct mcl_comm *c);        // mcl_comm.hip
void comm_forget(struct mcl_comm *c);      // mcl_comm.hip: the particle set changed, what the exchange knew is void
struct mcl_engine {
    mcl_config_t cfg{};
    int num_cu = 256;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;      // the per-update observation tables are built here, beside the resampling and ordering kernels
    hipEvent_t ev_obs = nullptr;        // ... and the ray stage waits for this
    bool obs_wait_pending = false;      // ... where its first kernel that reads them is launched (launch_rays)
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
    bool sweep_hyb_layout_ok = false;   // ... and of its hybrid form
    int env_sweep_hybrid = 1;           // MCL_SWEEP_HYBRID=0: ranges beyond the window take the global-field form alone; 2: the hybrid also for a fresh / spread set
    int env_sw_guide = 0;               // MCL_SW_GUIDE=<n>: the guide of k_sweep_plan's run lengths (default 2; 3 for the forms of ranges beyond the LDS window)
    int hyb_backoff = 0, hyb_skip = 0;  // updates in the global-field form after a hybrid update that flagged many particles (sweep_hybrid_decide)
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

#define HIPCHK(h, call)                                                                          \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess) {                                                                  \
            (h)->err = std::string(#call) + ": " + hipGetErrorString(e_);                        \
            return MCL_ERR_HIP;                                                                  \
        }                                                                                        \
    } while (0)

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

// ---- engine-side functions the sharded hosts use (defined in mcl_engine.hip)
namespace mcl_host {
int fail(mcl_engine *h, int code, const char *msg);
int fail(mcl_engine *h, int code, const std::string &msg);
std::string &create_error();                 // what mcl_*_last_error(NULL) reports (thread-local)
template <class T>
inline void dfree(T *&p)
{
    if (p) (void)hipFree(p);
    p = nullptr;
}
bool ready(mcl_engine *h, bool need_particles);
float elapsed(hipEvent_t a, hipEvent_t b);
void graph_reset(mcl_engine *h);
int scan_weights(mcl_engine *h, const uint64_t *d_q, uint64_t *d_cdf, int64_t n, uint64_t offset, uint64_t *d_total);
void unpack_result(mcl_engine *h);
int layout_adopt(mcl_engine *h, int64_t n);
int set_particles_impl(mcl_engine_t *h, const double *xyz, const double *weights, int64_t n, const double *weight_scale);
int stage_resample_launch(mcl_engine_t *h, const ParentSource &src, const uint64_t *d_cdf, int64_t n_parents, uint64_t q_total,
                          int64_t child_first, int64_t n_children_total, const double action[3]);
int export_compact_launch(mcl_engine_t *h, void *d_chunk, int64_t chunk_entries, int dst_device, hipStream_t stream);
int stage_resample_compact_launch(mcl_engine_t *h, const void *d_chunks, int32_t n_shards, int64_t chunk_entries, const int64_t *counts,
                                  const uint64_t *totals, int64_t n_per_shard, int32_t self_shard, int64_t child_first,
                                  int64_t n_children_total, const double action[3], unsigned long long *remote_count);
int stage_rays_launch(mcl_engine_t *h, const float *obs, int32_t n_beams, bool force_skip, double *d_max_out = nullptr);
void stage_rays_note(mcl_engine_t *h);
int stage_rays_finish(mcl_engine_t *h, const float *obs, int32_t n_beams);
int stage_weights_launch(mcl_engine_t *h, double global_max_logw, const double *d_global_max = nullptr);
int stage_weights_finish(mcl_engine_t *h);
void stage_commit_carry(mcl_engine_t *h);
// the few kernels mcl_comm.hip / mcl_group.hip launch themselves (on `stream`)
void launch_copy_double(hipStream_t stream, const double *src, double *dst);
void launch_set_double(hipStream_t stream, double *dst, double v);
void launch_spin_ms(hipStream_t stream, double ms);
void launch_stage_pack(hipStream_t stream, const unsigned long long *d_result, double *d_vec, int n_shards, int self, int listed, unsigned long long list_cap);
void launch_pack_records(hipStream_t stream, const double *x, const double *y, const double *th, int64_t n, double4 *out);
void launch_group_max(hipStream_t stream, const mcl::GroupMaxArgs &a);
}  // namespace mcl_host
