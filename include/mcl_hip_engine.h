/*
 * mcl_hip_engine.h — C ABI of the MI355X particle-filter update engine (libmcl_hip_engine.so).
 *
 * Drop-in boundary for the hot path of AE-HYU/monte_carlo_localization: the bodies of
 *     void ParticleFilter::MCL(const Eigen::Vector3d&, const std::vector<float>&)   src/particle_filter.cpp:652-694
 *     Eigen::Vector3d ParticleFilter::expected_pose()                               src/particle_filter.cpp:696-716
 * plus the state hand-over points that surround them (get_omap cpp:190-224, lidarCB cpp:297-313,
 * initialize_particles_pose cpp:388-398, initialize_global cpp:433-443, visualize cpp:946-958).
 * The reference has no FFI of its own (the functions are private members, hpp:39-43,74-75); the
 * entry points below are what a binding for this path has to offer, one per hand-over point, and
 * INTEGRATION.md shows the ~40-line patch that wires them into the unchanged class.
 *
 * Conventions
 *   - plain C, no C++/Eigen/ROS/torch types; the caller owns every host buffer, the engine owns
 *     all device memory behind the opaque handle and copies in/out synchronously;
 *   - particle matrices use the memory layout of Eigen::MatrixXd(N,3): column-major, i.e. N x's,
 *     then N y's, then N thetas (hpp:102); weights are std::vector<double> (hpp:103);
 *   - every function returns MCL_OK (0) or a negative mcl_status; nothing throws across the ABI.
 *     MCL_ERR_INVALID_ARG / NOT_READY / UNSUPPORTED are detected before anything is touched: the engine state is left as
 *     it was (the host patch then skips the tick exactly like a failed state_lock_.try_lock(), cpp:756).
 *     MCL_ERR_HIP means the HIP runtime failed part-way (out of memory, a lost device): what the call was replacing is
 *     then undefined -- after mcl_set_map / mcl_set_beam_angles the engine reports "not ready" until the call
 *     succeeds, after mcl_update / mcl_stage_* the particle set must be set or initialised again;
 *   - calls on one handle must be serialised by the caller (they are: single-threaded executor
 *     cpp:1022 + state_lock_ cpp:756/387/408).  mcl_update() is synchronous: when it returns the
 *     new particle set, weights and pose are final (its wall time feeds delay compensation,
 *     cpp:792-796).
 */
#ifndef MCL_HIP_ENGINE_H
#define MCL_HIP_ENGINE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MCL_ABI_VERSION 1

typedef struct mcl_engine mcl_engine_t;

typedef enum {
    MCL_OK = 0,
    MCL_ERR_INVALID_ARG = -1,   /* null pointer, size mismatch, bad enum            */
    MCL_ERR_NOT_READY = -2,     /* map / beam angles / particles not set yet        */
    MCL_ERR_HIP = -3,           /* a HIP runtime call failed; see mcl_last_error()  */
    MCL_ERR_NO_DEVICE = -4,     /* no gfx950 device visible                         */
    MCL_ERR_UNSUPPORTED = -5,   /* e.g. ray steps requested but not kept            */
    MCL_ERR_PEER = -6,          /* sharded update: another rank reported a failure; the update is void on every rank */
    MCL_ERR_TIMEOUT = -7        /* sharded update: a collective did not finish in time; the communicator was aborted  */
} mcl_status;

typedef enum {
    MCL_RESAMPLE_MULTINOMIAL = 0, /* what the reference does: std::discrete_distribution, cpp:658-665 */
    MCL_RESAMPLE_SYSTEMATIC = 1   /* one offset per update, u_m = (m + u0)/N                           */
} mcl_resample_mode;

typedef enum {
    MCL_WEIGHT_LOG = 0,           /* sum of log table entries, max-subtracted (default; no underflow)  */
    MCL_WEIGHT_PRODUCT = 1        /* sequential double product + pow, bit-for-bit cpp:566-578 incl.
                                     its underflow at >~200 beams; needs keep_ray_steps                */
} mcl_weight_mode;

typedef enum {
    MCL_RAYS_AUTO = 0,            /* MCL_RAYS_SWEEP from 65536 particles and 2^23 rays when the map and beam set
                                     allow it, else MCL_RAYS_SKIP                                      */
    MCL_RAYS_MARCH = 1,           /* literal fixed-step fp64 march on the int8 grid (cpp:611-650)      */
    MCL_RAYS_SKIP = 2,            /* same sample lattice, empty-space skipping on an LDS-resident
                                     distance-to-obstacle window; exactness guard falls back to MARCH  */
    MCL_RAYS_QUAD = 3,            /* SKIP with the work split by ray direction: one-byte-per-cell quadrant
                                     windows, two workgroups per CU.  LEGACY: only in a library built with
                                     -DMCL_LEGACY_RAY_KERNELS (the tests' libmcl_hip_engine_legacy.so); the product
                                     library answers MCL_ERR_UNSUPPORTED                                */
    MCL_RAYS_CELL = 4,            /* QUAD on particles ordered by grid cell and heading, one particle per
                                     lane: the lanes of a wave trace near-identical rays.  LEGACY, as QUAD */
    MCL_RAYS_SWEEP = 5            /* cell-sorted particles, one particle per lane, work items planned on the device (runs of
                                     units x direction wedges), 256-cell mirrored LDS windows, 64-bit fixed-point positions,
                                     beam directions turned by the scan's increment (evenly spaced scans); ranges up to 243 px
                                     in LDS windows, beyond that the same walk on mirrored copies of the wedge fields in global
                                     memory (any range whose sixteen fields fit 2^32 bytes).  What AUTO runs from 65 536 particles
                                     and 2^23 rays; below that MCL_RAYS_SKIP. */
} mcl_ray_kernel;

/* Upper bound (exclusive) on max_particles and on the particle total of a sharded set: weights are quantised to 2^-36 and
 * their exact sum is kept in 64 bits, so N * 2^36 must stay below 2^64 with one bit to spare. */
#define MCL_MAX_TOTAL_PARTICLES ((int64_t)1 << 27)

/* Numeric subset of the node's parameters (cpp:23-78) + engine knobs. */
typedef struct {
    int64_t max_particles;          /* MAX_PARTICLES, cpp:51                                   */
    int32_t device;                 /* HIP device ordinal                                      */
    uint64_t seed;                  /* Philox key; the reference seeds from random_device cpp:20 */
    double max_range_m;             /* cpp:54                                                  */
    double z_hit, z_short, z_max, z_rand, sigma_hit;   /* cpp:64-68                            */
    double squash_factor;           /* cpp:53 (INV_SQUASH_FACTOR = 1/squash_factor)            */
    double motion_dispersion_x, motion_dispersion_y, motion_dispersion_theta; /* cpp:71-73      */
    int32_t resample_mode;          /* mcl_resample_mode                                       */
    int32_t weight_mode;            /* mcl_weight_mode                                         */
    int32_t ray_kernel;             /* mcl_ray_kernel                                          */
    int32_t keep_ray_steps;         /* !=0: keep the N*B uint8 step indices of the last update */
    int32_t debug_force_exact;      /* 1: every ray takes the literal-march fallback (level 3);
                                       2: every ray takes the fp64 skipping loop (level 2)     */
    int32_t debug_count_probes;     /* !=0: tally examined grid probes (counters[2]); slower   */
    int32_t rays_per_lane;          /* SKIP kernel: independent rays in flight per lane (1..4); 0 = default */
    int32_t resample_neff_permille; /* 0 (default): resample on every update like the reference (cpp:656-665);
                                       r in 1..1000: resample only when the effective sample size
                                       (sum w)^2 / sum w^2 of the previous update is below r/1000 * N, otherwise the
                                       particles keep their identity and their weights multiply */
    int32_t graph_mode;             /* small updates are a chain of dependent launches: 0 (default) and 2 = shorten it once
                                       a first update has run with the same sizes (k_rays_skip path only): up to 8192
                                       particles mcl_update is three launches whose last one writes the result block to
                                       pinned host memory, where the host polls a stamp (MCL_TINY_POLL=0 in the
                                       environment at mcl_create: wait for the stream instead); above that the part after
                                       the resampling kernel is replayed as one hipGraph.  1 = launch by launch.  The
                                       results do not depend on it. */
    int32_t reserved[3];
} mcl_config_t;

/* Fills *cfg with the reference's defaults (config/mcl_config.yaml:6-40, cpp:23-47). */
void mcl_default_config(mcl_config_t *cfg);

int mcl_abi_version(void);

/* ---- lifetime ---------------------------------------------------------------------------- */
int mcl_create(const mcl_config_t *cfg, mcl_engine_t **out);
void mcl_destroy(mcl_engine_t *h);
/* Message of the last failing call on this handle ("" if none).  h may be NULL for mcl_create. */
const char *mcl_last_error(const mcl_engine_t *h);

/* ---- map + sensor table: replaces what get_omap() keeps (cpp:190-195) and
 *      precompute_sensor_model() builds (cpp:233-292) ------------------------------------- */
/* data: nav_msgs/OccupancyGrid.data, row-major H x W int8; resolution: MapMetaData.resolution
 * (float32); origin: info.origin.position.{x,y} (map yaw is ignored, cpp:628-629). */
int mcl_set_map(mcl_engine_t *h, const int8_t *data, uint32_t width, uint32_t height,
                float resolution, double origin_x, double origin_y);
/* MAX_RANGE_PX (cpp:195) for the current map. */
int mcl_get_max_range_px(const mcl_engine_t *h, int32_t *out);
/* The engine's sensor_model_table_: (P+1)^2 doubles, Eigen column-major (index d*(P+1)+r). */
int mcl_get_sensor_table(const mcl_engine_t *h, double *out, size_t n);

/* ---- beam geometry: downsampled_angles_ (cpp:306-310, hpp:117) ---------------------------- */
int mcl_set_beam_angles(mcl_engine_t *h, const float *angles, int32_t n_beams);

/* ---- particle state: particles_ / weights_ (hpp:102-103) ---------------------------------- */
/* After initialize_particles_pose / initialize_global rewrote the host copies (cpp:388-398,
 * 433-443).  n must be <= max_particles and becomes the active particle count. */
int mcl_set_particles(mcl_engine_t *h, const double *xyz_colmajor, const double *weights, int64_t n);
/* The same for one shard of a larger set: the fixed-point weights are scaled by the maximum weight of the WHOLE set (every
 * shard must be given the same value), so that the shards' weights are comparable. */
int mcl_set_particles_shard(mcl_engine_t *h, const double *xyz_colmajor, const double *weights, int64_t n, double max_weight_of_the_whole_set);
/* Device-side versions of the two initialisers (Philox draws instead of the host's rng_; same formulas):
 * Gaussian cloud (0.5 m, 0.5 m, 0.4 rad) around `pose` (cpp:388-398) / uniform over free cells with
 * theta ~ U[0,2pi) (cpp:408-443).  n particles become active with weights 1/n_total; first_global_index is
 * this shard's offset in a sharded set (0 and n_total = n on a single GPU).  At 4M-32M particles this
 * replaces a 100-800 MB host->device upload. */
int mcl_init_particles_pose(mcl_engine_t *h, const double pose[3], int64_t n, int64_t first_global_index, int64_t n_total);
int mcl_init_global(mcl_engine_t *h, int64_t n, int64_t first_global_index, int64_t n_total);
int mcl_get_particles(mcl_engine_t *h, double *xyz_colmajor, int64_t n);
int mcl_get_weights(mcl_engine_t *h, double *weights, int64_t n);
/* visualize()'s weighted sample of k rows (cpp:949-956): k draws from the current weights.
 * uniforms: k doubles in [0,1) (e.g. from the host's rng_) or NULL for Philox draws.
 * out: k x 3 column-major. */
int mcl_sample_particles(mcl_engine_t *h, int32_t k, const double *uniforms, double *out_colmajor);
/* get_current_pose()'s particles_.colwise().mean() (cpp:903-908). */
int mcl_particle_mean(mcl_engine_t *h, double out[3]);

/* ---- the update: body of MCL(action, observation) (cpp:652-694) --------------------------- */
/* action = (forward displacement m, unused, yaw displacement rad) (cpp:761-772);
 * obs = n_beams downsampled ranges in metres (cpp:316-320).
 * Injection hooks for reference-exact parity: normals = N x 3 ROW-major doubles in
 * the reference's draw order (n_x, n_y, n_theta per particle, cpp:496-498), uniforms = N doubles
 * in [0,1) as discrete_distribution would draw them (cpp:663).  Either may be NULL -> Philox. */
int mcl_update(mcl_engine_t *h, const double action[3], const float *obs, int32_t n_beams,
               const double *normals_nx3, const double *uniforms_n);
/* Same update fed with the RAW LaserScan.ranges: the engine keeps every angle_step-th range like
 * lidarCB does (cpp:316-320) — ceil(n_ranges/angle_step) must equal the number of beam angles set. */
int mcl_update_scan(mcl_engine_t *h, const double action[3], const float *ranges, int32_t n_ranges, int32_t angle_step);
/* sensor_model() + normalisation only, on the current particles (cpp:676-686): no resample, no
 * motion.  Used by parity tests of the ray-cast / likelihood stage. */
int mcl_sensor_update(mcl_engine_t *h, const float *obs, int32_t n_beams);

/* ---- expected_pose() (cpp:696-716) --------------------------------------------------------- */
int mcl_expected_pose(mcl_engine_t *h, double out[3]);

/* ---- TimingStats mirror (utils.hpp:51-57): resampling, motion_model, query_prep, ray_casting,
 *      sensor_model (table eval + normalise), total — milliseconds of the LAST update -------- */
int mcl_get_stage_timings(const mcl_engine_t *h, double ms[6]);

/* ---- parity / diagnostics ------------------------------------------------------------------ */
int mcl_get_resample_indices(mcl_engine_t *h, int32_t *idx, int64_t n);      /* parents of last update */
int mcl_get_ray_steps(mcl_engine_t *h, uint8_t *steps, size_t n);            /* N*B, needs keep_ray_steps; MAX_RANGE_PX <= 255 */
int mcl_get_ray_steps16(mcl_engine_t *h, uint16_t *steps, size_t n);         /* the same for any range (cpp:195 has no bound) */
int mcl_get_log_weights(mcl_engine_t *h, double *logw, int64_t n);           /* un-normalised log w   */
/* counters of the last update: [0] rays resolved by the literal-march fallback (level 3),
 * [1] particles that did not fit the LDS window (global-memory path), [2] grid probes examined
 * (only with debug_count_probes), [3] rays re-run by the fp64 loop (level 2) */
int mcl_get_counters(mcl_engine_t *h, uint64_t out[4]);
/* The compact parent list: the scan of an update's fixed-point weights also lists, in index order, the
 * particles whose weight is not zero -- after an update with many beams a few per cent of the set -- and the next
 * resampling searches and gathers from that list instead of the full CDF and record arrays (same draw: a particle with a
 * zero fixed-point weight is never selected).  n_entries: length of the list that describes the current weights, -1 when
 * there is none (more than max_particles / 4 particles carry weight, or a small update); used_by_last_update: whether the
 * last mcl_update / staged resampling drew from a list.  MCL_NO_COMPACT=1 in the environment at mcl_create disables it. */
int mcl_get_compact_list(const mcl_engine_t *h, int64_t *n_entries, int32_t *used_by_last_update);
/* switches cfg.debug_count_probes on an existing engine (the next update tallies counters[2]; bench.py's untimed probe count) */
int mcl_set_debug_count_probes(mcl_engine_t *h, int32_t on);
/* duration (ms) of the dominant kernel (ray cast + likelihood) in the last update, measured
 * with HIP events on the engine's own stream */
int mcl_get_ray_kernel_ms(const mcl_engine_t *h, double *ms);
/* which ray kernel the last update ran: 1 k_rays_march, 2 k_rays_skip, 3 k_rays_quad, 4 k_rays_cell, 5 k_rays_sweep */
int mcl_get_ray_kernel_id(const mcl_engine_t *h, int32_t *kernel);
/* The form of the windowed ray kernel the last ray stage ran (k_rays_sweep, csrc/mcl_rays_sweep.h): out = [1 if it probed the wedge
 * fields in global memory (ranges beyond 243 px), 2 if it walked in LDS windows and went on in those fields where a ray left its
 * window (the hybrid form of such ranges: evenly spaced scans), 1 if it turned the beam direction by the scan's increment instead of fetching it
 * (evenly spaced scans), 1 if it walked two rays per lane].  A performance diagnostic (bench.py prices the instruction stream of the
 * form that ran); results do not depend on it. */
int mcl_get_ray_kernel_variant(const mcl_engine_t *h, int32_t out[3]);
/* Which ray kernel an update over n_particles (<= 0: the active count, or max_particles before any particles are set) WILL run
 * with the current configuration, map and beam set -- the same pure function mcl_update consults, callable before the first
 * update (needs the map and the beam angles).  *kernel as mcl_get_ray_kernel_id, 0 = the configured kernel cannot run with
 * this map / beam set (mcl_update would return MCL_ERR_UNSUPPORTED).  *reason (may be NULL) receives a static string naming
 * what decided, e.g. why AUTO stays off k_rays_sweep: the fast windowed kernels need beam angles that increase over less
 * than a turn and MAX_RANGE_PX <= 243; anything else takes k_rays_skip (same results, several times slower at size). */
int mcl_get_planned_ray_kernel(const mcl_engine_t *h, int64_t n_particles, int32_t *kernel, const char **reason);
/* Effective sample size (sum w)^2 / sum w^2 of the current weights, and whether the last mcl_update resampled
 * (always 1 with resample_neff_permille == 0). */
int mcl_get_effective_sample_size(const mcl_engine_t *h, double *n_eff, int32_t *resampled_last_update);

/* ---- host-side precomputation, callable without a device (what mcl_set_map uploads) --------- */
/* (P+1)^2 doubles, Eigen column-major (index d*(P+1)+r): the restatement of precompute_sensor_model
 * (cpp:233-292) the engine uses. */
int mcl_host_sensor_table(const mcl_config_t *cfg, int32_t max_range_px, double *out, size_t n);
/* Skip-distance field of DESIGN.md §4.2 on the padded grid: (height+1) x (width+1) bytes, row-major,
 * 0 = stop cell, otherwise how many samples the fixed-step march may advance from a sample in that cell. */
int mcl_host_skip_field(const int8_t *data, uint32_t width, uint32_t height, uint8_t *out, size_t n);
/* Same for rays whose direction lies in `quadrant` (0:+x+y 1:-x+y 2:-x-y 3:+x-y): only stop cells a ray of that
 * quadrant can still reach bound the jump. */
int mcl_host_skip_field_dir(const int8_t *data, uint32_t width, uint32_t height, int32_t quadrant, uint8_t *out, size_t n);
/* Same for rays whose direction angle lies in wedge `wedge` of MCL_WEDGES equal sectors of the turn (sector k spans
 * [2*pi*k/MCL_WEDGES, 2*pi*(k+1)/MCL_WEDGES]); the fields MCL_RAYS_CELL stages in LDS. */
#define MCL_WEDGES 16
int mcl_host_skip_field_wedge(const int8_t *data, uint32_t width, uint32_t height, int32_t wedge, uint8_t *out, size_t n);
/* Layout of the copies of the wedge fields that the global-field form of the ray kernel (ranges beyond 243 px) probes in place:
 * out = [usable (0 / 1), row pitch, rows per field (ringed grid + tail of stop rows), bytes per field, bytes of the allocation,
 * the largest byte offset a walk can form].  The kernel addresses the allocation from its start with offsets >= 0; a walk
 * starts inside the ringed grid of its field and advances by at most max_range_px samples of at most one cell per axis, so
 * out[5] < out[4] < 2^32 is the whole memory-safety argument (tests/test_sweep_addressing.py enumerates it).  Host only. */
int mcl_host_sweep_global_layout(uint32_t width, uint32_t height, int32_t max_range_px, int64_t out[6]);

/* ---- multi-GPU staging (one engine per rank; collectives are the host's) -- */
/* Device pointers of engine-owned buffers so the host can hand them to RCCL without copies. */
typedef enum {
    MCL_BUF_X = 0, MCL_BUF_Y = 1, MCL_BUF_THETA = 2,  /* current particle columns, double[N]      */
    MCL_BUF_QWEIGHT = 3,                               /* uint64[N] fixed-point weights (2^-36)    */
    MCL_BUF_LOGW = 4,                                  /* double[N]                                */
    MCL_BUF_SCALARS = 5                                /* double[8]: max logw, sum w, sum q (as
                                                          u64 bits), sum wx, wy, wsin, wcos, -    */
} mcl_buffer_id;
int mcl_device_ptr(mcl_engine_t *h, int32_t which, void **dev_ptr);
/* Copies the current particle columns and fixed-point weights into caller-owned DEVICE buffers
 * (n entries each; any of them may be NULL), synchronously. */
int mcl_export_state(mcl_engine_t *h, double *d_x, double *d_y, double *d_theta, uint64_t *d_q);
/* Host copy of SCALARS (see mcl_buffer_id), valid after stage_propagate / stage_weights / update. */
int mcl_get_scalars(mcl_engine_t *h, double out[8]);
/* The host copy of SCALARS the last stage call already read back (valid right after mcl_stage_rays: [0] = local
 * max log-weight; after mcl_stage_weights: the local sums): no device access, no synchronisation. */
int mcl_get_host_scalars(const mcl_engine_t *h, double out[8]);
/* Stage 1: resample this rank's n children [child_first, child_first+n) out of the GLOBAL parent
 * set (device pointers, n_parents entries each: columns + inclusive global CDF of q with total
 * q_total), apply motion, cast rays, leave log-weights and the local max in SCALARS[0]. */
int mcl_stage_propagate(mcl_engine_t *h, const double *d_px, const double *d_py, const double *d_pth,
                        const uint64_t *d_cdf, int64_t n_parents, uint64_t q_total,
                        int64_t child_first, int64_t n_children_total,
                        const double action[3], const float *obs, int32_t n_beams);
/* Stage 1 in two halves, so that the host can start gathering the children for the NEXT update while the
 * ray kernel of this one runs: mcl_stage_resample returns when the children (resampled + moved) are final,
 * mcl_stage_rays casts their rays and leaves the local max log-weight in SCALARS[0]. */
int mcl_stage_resample(mcl_engine_t *h, const double *d_px, const double *d_py, const double *d_pth,
                       const uint64_t *d_cdf, int64_t n_parents, uint64_t q_total,
                       int64_t child_first, int64_t n_children_total, const double action[3]);
/* The same with the parents as packed records {x, y, theta, unused} (4 doubles each): one all-gather instead of
 * three and one fetch per gathered parent.  mcl_export_records copies this engine's current particles in that
 * form to d_records (N x 32 bytes, device memory). */
int mcl_export_records(mcl_engine_t *h, void *d_records);
int mcl_stage_resample_records(mcl_engine_t *h, const void *d_records, const uint64_t *d_cdf, int64_t n_parents,
                               uint64_t q_total, int64_t child_first, int64_t n_children_total, const double action[3]);
/* The exchange without wholesale gathers: only the fixed-point weights (8 B per particle) are gathered; each rank then
 * asks which parents its children selected (mcl_stage_resample_indices: global parent index per local child, written to
 * a caller-owned DEVICE buffer of N int32, engine state untouched), fetches the distinct ones from their owners with
 * whatever transport the host has (torch.distributed all-to-all in dist.py), and hands the compact record table plus the
 * per-child position in it to mcl_stage_motion_records, which gathers, applies the motion model and makes the children
 * current.  Same random streams as the fused call: bit-identical children. */
/* Helpers of that exchange, so that the host needs no pass over n_total elements of its own:
 *   mcl_stage_distinct_parents: d_parent[n_children] (global indices < n_total, DEVICE) -> the distinct ones in ascending
 *     order (= grouped by owning shard) in d_distinct (DEVICE, room for n_children int64), the position of every child's
 *     parent among them in d_slot (DEVICE, n_children int32) and their number in *count (host).  A bitmap over the
 *     global indices and its popcount prefix: every pass but the marking runs over n_total / 32 words.
 *   mcl_export_records_at: the packed records {x, y, theta, unused} of the listed local particles (indices into this
 *     engine's current set, DEVICE int64) -> d_out[count] (DEVICE): what a shard answers a request with. */
int mcl_stage_distinct_parents(mcl_engine_t *h, const int32_t *d_parent, int64_t n_children, int64_t n_total,
                               int64_t *d_distinct, int32_t *d_slot, int64_t *count);
int mcl_export_records_at(mcl_engine_t *h, const int64_t *d_index, int64_t count, void *d_out);
int mcl_stage_resample_indices(mcl_engine_t *h, const uint64_t *d_cdf, int64_t n_parents, uint64_t q_total, int64_t child_first,
                               int64_t n_children_total, int32_t *d_parent_idx);
int mcl_stage_motion_records(mcl_engine_t *h, const void *d_records, int64_t n_records, const int32_t *d_record_of_child,
                             int64_t child_first, int64_t n_children_total, const double action[3]);
/* The exchange of a sharded set when every shard has a compact parent list (mcl_get_compact_list; the usual case after an
 * update with many beams): the shards gather their LISTS -- 44 B per particle that carries weight, a few per cent of the set --
 * instead of every weight, and nothing else has to travel: the merged lists are the whole parent population.
 *   mcl_compact_chunk_bytes: size of a chunk of chunk_entries list entries (a multiple of 64, the same on every shard and at
 *     least the longest list); mcl_export_compact copies this engine's list into d_chunk (DEVICE memory of that size); the
 *     host gathers the chunks in shard order (all-gather) into d_chunks;
 *   mcl_stage_resample_compact: counts[r] = length of shard r's list, totals[r] = its fixed-point weight total (SCALARS[2] of
 *     shard r as uint64); merges the chunks into one CDF, draws this shard's children from it (same thresholds as every other
 *     path: bit-identical children), applies the motion model and makes the children current.  Global parent index =
 *     shard * n_per_shard + local index. */
int mcl_compact_chunk_bytes(int64_t chunk_entries, int64_t *bytes);
int mcl_export_compact(mcl_engine_t *h, void *d_chunk, int64_t chunk_entries);
int mcl_stage_resample_compact(mcl_engine_t *h, const void *d_chunks, int32_t n_shards, int64_t chunk_entries, const int64_t *counts,
                               const uint64_t *totals, int64_t n_per_shard, int32_t self_shard, int64_t child_first, int64_t n_children_total,
                               const double action[3]);
int mcl_stage_rays(mcl_engine_t *h, const float *obs, int32_t n_beams);
/* Leave n_cus compute units out of k_rays_quad's persistent grid (it otherwise occupies every CU for the
 * whole kernel, which would serialise a collective launched beside it). */
int mcl_set_reserved_cus(mcl_engine_t *h, int32_t n_cus);
/* Stage 2: given the GLOBAL max log-weight, compute w, q and the local partial sums (SCALARS). */
int mcl_stage_weights(mcl_engine_t *h, double global_max_logw);
/* Stage 3: install the GLOBAL sums (sum w, wx, wy, wsin, wcos) so that get_weights /
 * expected_pose report globally normalised values. */
int mcl_stage_finish(mcl_engine_t *h, const double global_sums[5]);
/* ---- the same stages ORDERED ON THE DEVICE: one host wait per update ----------------------------------------------------
 * The calls above return when their stage has finished (the host reads a value between them).  The *_async forms only enqueue
 * on the engine's stream; the values the stages exchange stay in device memory the caller owns (its collective library's
 * buffers), and the caller's stream and the engine's are ordered by events:
 *   mcl_external_wait_stream(h, s): work enqueued on s after the call waits for everything enqueued on the engine so far;
 *   mcl_stream_wait_external(h, s): the engine's later work waits for everything enqueued on s so far (s: a hipStream_t).
 * One update of a sharded set, lists known from the previous update's sums:
 *   mcl_export_compact_async -> [s waits] all-gather of the chunks on s -> [engine waits] mcl_stage_resample_compact_async ->
 *   mcl_stage_rays_async (local max log-weight -> *d_local_max) -> [s waits] all-reduce MAX on s -> [engine waits]
 *   mcl_stage_weights_async (reads *d_global_max; writes this shard's part of the SUM vector, 5 + 3 * n_shards + 2 doubles:
 *   [sum w, sum w x, sum w y, sum w sin, sum w cos | per shard: list length + 1 (0: no list), low / high 32 bits of its
 *   fixed-point weight total | 1.0 if this shard's ray stage overflowed its fix-up lists | sum w^2], the other shards' slots zeroed)
 *   -> [s waits] all-reduce SUM on s, copy to the host, THE host wait -> mcl_stage_complete(global sums, &redo): waits for the
 *   engine's stream (already drained), takes over the read-backs the synchronous calls do one by one.  *redo = 1 (the last
 *   but one element of the summed vector is non-zero on every rank then): this shard's log-weights are incomplete -- run
 *   mcl_stage_rays, the MAX exchange, mcl_stage_weights, the SUM exchange and mcl_stage_finish once more (all ranks).
 * Results are those of the synchronous calls bit for bit (same kernels, same order). */
/* Adaptive resampling (cfg.resample_neff_permille = r > 0) in a sharded set.  The decision is the host's, from the sums of the
 * PREVIOUS update over the whole set (the same numbers on every shard): keep when (sum w)^2 >= r / 1000 * N_total * sum w^2.
 * mcl_stage_keep then replaces the exchange and the resampling call of this update: every particle is its own parent (reported
 * as child_first + its index), the motion model runs with the same random streams, and the ray stage adds the previous
 * update's log-weights minus their global maximum, as mcl_update does.  Launch only.  MCL_ERR_NOT_READY before the first staged
 * update of a particle set.  (mcl_comm_update and mcl_group_update take the decision themselves.) */
int mcl_stage_keep(mcl_engine_t *h, int64_t child_first, int64_t n_children_total, const double action[3]);
int mcl_stream_wait_external(mcl_engine_t *h, void *stream);
int mcl_external_wait_stream(mcl_engine_t *h, void *stream);
int mcl_export_compact_async(mcl_engine_t *h, void *d_chunk, int64_t chunk_entries);
int mcl_stage_resample_compact_async(mcl_engine_t *h, const void *d_chunks, int32_t n_shards, int64_t chunk_entries, const int64_t *counts,
                                     const uint64_t *totals, int64_t n_per_shard, int32_t self_shard, int64_t child_first,
                                     int64_t n_children_total, const double action[3]);
int mcl_stage_rays_async(mcl_engine_t *h, const float *obs, int32_t n_beams, double *d_local_max);
int mcl_stage_weights_async(mcl_engine_t *h, const double *d_global_max, double *d_vec, int32_t n_shards, int32_t self_shard);
int mcl_stage_complete(mcl_engine_t *h, const double global_sums[5], int32_t *redo);
/* ---- one process per GPU, the exchange in native code (RCCL on the engine's own stream) -----------------------------------
 * The engine holds an RCCL communicator; mcl_comm_update is ONE sharded update: all-gather of the compact parent lists,
 * resampling + motion from the merged lists, ray stage, all-reduce MAX of the max log-weight, weights + scan + list,
 * all-reduce SUM of the sums -- the three collectives enqueued on the engine's stream between its kernels (no second stream,
 * no event between streams, no host code between the stages), ONE host wait at the end.  Bit-identical to mcl_update of a
 * single engine holding all shards (same kernels as the staged calls above).
 *   mcl_comm_available: MCL_OK when an RCCL library can be used (the one already loaded in the process, e.g. a torch
 *     process's, else librccl.so.1 of the ROCm installation; taken with dlopen -- the engine does not link RCCL);
 *   mcl_comm_unique_id: on ONE rank; the host passes the 128 bytes to every rank (any channel: MPI, torch.distributed, a file);
 *   mcl_comm_create: COLLECTIVE (every rank calls it, ncclCommInitRank inside); one rank per device;
 *   mcl_comm_update: == ParticleFilter::MCL(action, observation) + expected_pose() (cpp:652-716) for the whole sharded set.  The
 *     shards' list lengths and weight totals come from the PREVIOUS update's summed vector (mcl_stage_weights_async documents
 *     it), which the communicator keeps.  When they are not known (the first update after the particles were set or
 *     initialised on every rank) or some shard has no list, the update takes the DENSE exchange instead, also on the engine's
 *     stream: all-gather of every shard's fixed-point weights and packed records (8 + 32 B per particle), one global CDF, the
 *     same draw (one more host wait, for the weight total).  Every rank decides from the same numbers, so every rank issues the
 *     same collectives.  The global sums are installed as mcl_stage_finish does; mcl_comm_get_vector returns the summed vector
 *     (5 + 3 * n_ranks + 2 doubles) of the last update.  mcl_comm_set_lists hands over list lengths (-1: none) and weight totals
 *     found by other means (a host that ran an update through the stage calls);
 *   mcl_comm_stats: bytes the last LIST exchange delivered to this rank (padded chunks) / carried (entries), host waits of the
 *     last update; mcl_comm_last_exchange: *dense = 0 the last update exchanged lists, 1 it took the dense exchange (with the bytes
 *     received), 2 it exchanged nothing (adaptive resampling kept the set). */
int mcl_comm_available(const char **why);
int mcl_comm_unique_id(unsigned char id[128]);
int mcl_comm_create(mcl_engine_t *h, const unsigned char id[128], int32_t n_ranks, int32_t rank);
int mcl_comm_selftest(mcl_engine_t *h);   /* COLLECTIVE: the three collectives of an update on known data; MCL_OK or what failed */
int mcl_comm_destroy(mcl_engine_t *h);
int mcl_comm_set_lists(mcl_engine_t *h, const int64_t *counts, const uint64_t *totals);
int mcl_comm_update(mcl_engine_t *h, const double action[3], const float *obs, int32_t n_beams, double pose_out[3]);
int mcl_comm_get_vector(const mcl_engine_t *h, double *vec_out, int32_t n);
int mcl_comm_stats(const mcl_engine_t *h, uint64_t *list_bytes_received, uint64_t *list_payload_bytes, int32_t *host_waits);
int mcl_comm_last_exchange(const mcl_engine_t *h, int32_t *dense, uint64_t *weights_bytes, uint64_t *records_bytes);

/* Inclusive scan of q (uint64) on the engine's stream: cdf[i] = offset + q[0] + ... + q[i]. */
int mcl_scan_weights(mcl_engine_t *h, const uint64_t *d_q, uint64_t *d_cdf, int64_t n, uint64_t offset);

/* ---- several GPUs behind one handle (SURVEY.md §8(b).1 "device list", §8(e)) -------------------------------------
 * The reference is ONE process (main, cpp:1019-1025; MCL called from timer_update, cpp:777): a group owns one engine per
 * listed device, shards the particle set contiguously (n_total / n_devices each; cfg->max_particles is PER DEVICE,
 * cfg->device is ignored) and mirrors the single-engine entry points the host patch uses.  Per update a device receives
 * the other shards' compact parent lists (44 B per particle that carries weight; mcl_get_compact_list) and draws its own
 * children from the merged lists -- or, when some shard has no list (first update, flat weights), the other shards'
 * fixed-point weights (8 B per particle), scans the same exact global CDF and reads each selected parent where it lives (peer
 * pointer over xGMI); the maximum is taken on the devices (each reads its peers' maxima), the sums are combined on the host (one
 * wait per update); the phases are ordered by events between the devices' streams.  Results are bit-identical to a single engine
 * holding all particles.  The devices must have peer access to each other; weight_mode LOG only; resample_neff_permille works
 * on the sums of the whole set. */
typedef struct mcl_group mcl_group_t;
int mcl_group_create(const mcl_config_t *cfg, const int32_t *devices, int32_t n_devices, mcl_group_t **out);
void mcl_group_destroy(mcl_group_t *g);
const char *mcl_group_last_error(const mcl_group_t *g);
int32_t mcl_group_size(const mcl_group_t *g);
int mcl_group_engine(mcl_group_t *g, int32_t i, mcl_engine_t **out);           /* the i-th shard's engine (diagnostics) */
int mcl_group_set_map(mcl_group_t *g, const int8_t *data, uint32_t width, uint32_t height, float resolution, double origin_x,
                      double origin_y);
int mcl_group_set_beam_angles(mcl_group_t *g, const float *angles, int32_t n_beams);
/* xyz: n_total x 3 column-major, weights: n_total (any non-negative weights with a positive maximum: every shard is scaled
 * by the maximum of the whole set); n_total a multiple of the device count */
int mcl_group_set_particles(mcl_group_t *g, const double *xyz, const double *weights, int64_t n_total);
int mcl_group_init_particles_pose(mcl_group_t *g, const double pose[3], int64_t n_total);
int mcl_group_init_global(mcl_group_t *g, int64_t n_total);
int mcl_group_update(mcl_group_t *g, const double action[3], const float *obs, int32_t n_beams);
int mcl_group_expected_pose(mcl_group_t *g, double out[3]);
int mcl_group_get_particles(mcl_group_t *g, double *xyz, int64_t n_total);
int mcl_group_get_weights(mcl_group_t *g, double *weights, int64_t n_total);
int mcl_group_get_resample_indices(mcl_group_t *g, int32_t *idx, int64_t n_total);   /* global parent indices */
int mcl_group_get_stage_timings(const mcl_group_t *g, double ms[6]);           /* per stage: the slowest device */
/* bytes the last update moved between devices: [0] received by the device that received MOST for the parent population -- the
 * other shards' compact lists (44 B per particle that carries weight, copied entry-exact) or, when some shard had no list, their
 * fixed-point weights (8 B per particle);
 * [1] parent records read from peers by ALL devices (weights exchange only; children with a remote parent x 32 B: an upper
 * bound, a shared parent is cached after its first fetch).  mcl_group_exchanged_lists: 1 when the last update exchanged lists. */
int mcl_group_exchange_bytes(const mcl_group_t *g, uint64_t out[2]);
int32_t mcl_group_exchanged_lists(const mcl_group_t *g);

#ifdef __cplusplus
}
#endif
#endif /* MCL_HIP_ENGINE_H */
