// mcl_group.hip -- mcl_group_*: several GPUs behind one handle, driven by one process (peer copies, peer-pointer reads).
#include "mcl_engine_internal.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>

using namespace mcl_host;

extern "C" {

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

const char *mcl_group_last_error(const mcl_group_t *g) { return g ? g->err.c_str() : create_error().c_str(); }

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
    create_error().clear();
    if (!cfg || !devices || !out || n_devices <= 0 || n_devices > mcl::kMaxShards) { create_error() = "bad group arguments (1..16 devices)"; return MCL_ERR_INVALID_ARG; }
    *out = nullptr;
    if (cfg->weight_mode != MCL_WEIGHT_LOG) {
        create_error() = "a device group needs weight_mode LOG";
        return MCL_ERR_UNSUPPORTED;
    }
    if ((int64_t)cfg->max_particles * n_devices >= MCL_MAX_TOTAL_PARTICLES) { create_error() = "particle total must stay below 2^27"; return MCL_ERR_INVALID_ARG; }
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
            create_error() = "group buffers: hipMalloc failed";
            mcl_group_destroy(g);
            return MCL_ERR_HIP;
        }
        // parents are read where they live: peer access to every other device of the group
        for (int o = 0; o < n_devices; ++o) {
            if (devices[o] == devices[d]) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, devices[d], devices[o]) != hipSuccess || !can) {
                create_error() = "devices of a group must have peer access to each other";
                mcl_group_destroy(g);
                return MCL_ERR_UNSUPPORTED;
            }
            const hipError_t pe = hipDeviceEnablePeerAccess(devices[o], 0);
            if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) { create_error() = "hipDeviceEnablePeerAccess failed"; mcl_group_destroy(g); return MCL_ERR_HIP; }
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
            launch_pack_records(e->stream, e->d_x[e->cur], e->d_y[e->cur], e->d_th[e->cur], n, e->d_pack[e->cur]);
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
                if (!rc) launch_copy_double(e->stream, e->d_scalars, lmax);
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
            launch_group_max(e->stream, ma);
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
