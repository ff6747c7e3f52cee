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

#include "mcl_engine_internal.h"
#include "mcl_kernels.h"

namespace {

thread_local std::string g_create_error;

}  // namespace

static_assert(MCL_WEDGES == mcl::kWedges || MCL_KWEDGES != 16, "include/mcl_hip_engine.h and csrc/mcl_wedge.h disagree");


namespace {


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
// The hybrid form of k_rays_sweep (ranges beyond the LDS window: the walk leaves the window into the global wedge fields where a
// ray is that long): for evenly spaced scans (the turned-direction walk), when both sets of fields exist.  MCL_SWEEP_HYBRID=0:
// the global-field form alone, as in round 4.
bool sweep_hybrid(const mcl_engine *h)
{
    return h->sweep_global && h->env_sweep_hybrid != 0 && h->rec_ok && h->sweep_hyb_layout_ok && h->d_beam_err != nullptr &&
           h->d_distw != nullptr && h->d_distg != nullptr && (size_t)mcl::kSwWinBytes + (size_t)h->ltd_cols * 8 <= 80 * 1024 - 64;
}

// (... of THIS update: a set that was just initialised, or whose last update left many particles to the far pass -- the spread
//  cloud of a re-localisation --, takes the global-field form: every lane has its own origin there, where the hybrid's windows
//  would hand a good part of such a cloud to the far pass.  fine025 stand-in, 4M uniform: first update 39 ms against 64.)
bool sweep_far_expected(const mcl_engine *h) { return h->far_fresh || h->h_result[15] >= mcl::kFarWindowedMin / 2; }
// Decided ONCE per update (launch_rays).  A hybrid update that left many particles to the far pass -- a cloud that has not converged
// yet -- is followed by 1, 2, 4 .. 32 updates in the global-field form before the hybrid is tried again (a global-field update
// flags nothing, so its count says nothing about the cloud: without the back-off the two forms would alternate).
// MCL_SWEEP_HYBRID=2: always the hybrid (the tests' setting).
bool sweep_hybrid_decide(mcl_engine *h)
{
    if (!sweep_hybrid(h)) return false;
    if (h->env_sweep_hybrid == 2) return true;
    if (h->far_fresh) { h->hyb_backoff = 0; h->hyb_skip = 0; return false; }
    if (h->last_sweep_global == 2) {
        if (h->h_result[15] >= mcl::kFarWindowedMin / 2) { h->hyb_backoff = h->hyb_backoff ? std::min(32, 2 * h->hyb_backoff) : 1; h->hyb_skip = h->hyb_backoff; }
        else h->hyb_backoff = 0;
    }
    if (h->hyb_skip > 0) { --h->hyb_skip; return false; }
    return true;
}

int sweep_play(const mcl_engine *h, bool hybrid)
{
    if (hybrid) return mcl::kSwSide - (mcl::kSwHybReach + 2) - 3;
    if (h->sweep_global) return 1 << 20;             // every lane has its own origin there: no window a run could outgrow
    return mcl::kSwSide - (h->P + 2) - 3;
}

int64_t max_sweep_units(int64_t n) { return (n + mcl::kSwUnit - 1) / mcl::kSwUnit + mcl::kSwMaxCuts + 2; }

int launch_sweep_plan(mcl_engine *h, int64_t n, int nwg, int g, bool hybrid)
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
    const int play = sweep_play(h, hybrid);                             // cells a window leaves for the particles of an item
    // (EV_K0 = the stop event of the kernel before the ray kernel, EV_K1 = the ray kernel's own: its duration, dispatch included, at no cost)
    hipExtLaunchKernelGGL(mcl::k_sweep_plan, dim3(1), dim3(1024), mcl::kPlanLds, h->stream, nullptr, h->ev[EV_K0], 0, h->d_unit_sums, h->d_nunits, ngroups, nwg,
                          (double)(play / 2 - 1), h->env_sw_guide > 0 ? h->env_sw_guide : (h->sweep_global ? 3 : 2), h->d_items, h->d_centres, h->d_nitems);
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
    [[maybe_unused]] const bool windows_ok = no_windows == nullptr;           // monotone beams over less than a turn, room for a byte window
#ifdef MCL_LEGACY_RAY_KERNELS
    if (rk == MCL_RAYS_QUAD) { w = windows_ok && h->quad_layout_ok ? "configured: MCL_RAYS_QUAD" : (no_windows ? no_windows : "k_rays_quad's LDS layout check failed"); return windows_ok && h->quad_layout_ok ? 3 : 0; }
    if (rk == MCL_RAYS_CELL) { w = windows_ok && h->cell_layout_ok ? "configured: MCL_RAYS_CELL" : (no_windows ? no_windows : "k_rays_cell's LDS layout check failed"); return windows_ok && h->cell_layout_ok ? 4 : 0; }
#else
    // k_rays_quad / k_rays_cell, the predecessors of k_rays_sweep, are not part of this library (csrc/mcl_rays_legacy.h, a build
    // flag; the tests load libmcl_hip_engine_legacy.so for them)
    if (rk == MCL_RAYS_QUAD || rk == MCL_RAYS_CELL) { w = "MCL_RAYS_QUAD / MCL_RAYS_CELL are not built into this library (build flag MCL_LEGACY_RAY_KERNELS)"; return 0; }
#endif
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
        w = sweep_hybrid(h) ? "AUTO: at least 65536 particles and 2^23 rays, evenly spaced beams; MAX_RANGE_PX > 243 (or MCL_SWEEP_GLOBAL): k_rays_sweep walks in LDS windows and goes on in the wedge fields in global memory where a ray leaves its window"
          : h->sweep_global ? "AUTO: at least 65536 particles and 2^23 rays, monotone beams; MAX_RANGE_PX > 243 (or MCL_SWEEP_GLOBAL): k_rays_sweep probes the wedge fields in global memory"
                            : "AUTO: at least 65536 particles and 2^23 rays, monotone beams, MAX_RANGE_PX <= 243";
        return 5;
    }
#ifdef MCL_LEGACY_RAY_KERNELS
    if (big && windows_ok && h->cell_layout_ok) { w = no_sweep; return 4; }
#endif
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
    a.beam_alast = h->B > 0 ? (double)h->angles[h->B - 1] : 0.0;
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
        const bool sweep_hyb_now = sweep && sweep_hybrid_decide(h);      // k_rays_sweep's hybrid form for THIS update
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
            clr.bbox_play = h->env_no_bucket_cuts ? -1 : (sweep ? sweep_play(h, sweep_hyb_now) : 0);
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
            if (!radix && !sweep) hipLaunchKernelGGL(mcl::k_hist_clear, dim3(nparts), dim3(256), 0, h->stream, h->d_hist, h->d_tile_used);
            if (sweep) {
                // (... and, after a counting sort, the histogram's used tiles back to zero: k_hist_clear's work)
                hipLaunchKernelGGL(mcl::k_unit_sums, dim3((unsigned)std::min<int64_t>(max_sweep_units(n), (n + mcl::kSwUnit - 1) / mcl::kSwUnit + 256)), dim3(256), 0, h->stream, h->d_pcs, h->d_unit_begin, h->d_nunits,
                                   h->d_unit_sums, radix ? (uint32_t *)nullptr : h->d_hist, h->d_tile_used, nparts);
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
        // the tables of this update's scan were built on the second stream beside the resampling and the ordering: from here on they are read
        if (h->obs_wait_pending) { h->obs_wait_pending = false; HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_obs, 0)); }
        if (sweep) {
            if (!h->d_Ltd) return fail(h, MCL_ERR_HIP, "k_rays_sweep: table not allocated (internal)");
            if (!h->ltd_ready) build_ltd(h);          // a caller whose table decision was made for another particle count
            const int rc_plan = launch_sweep_plan(h, n, nseg, sweep_g, sweep_hyb_now);
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
            a.far_windowed = sweep_far_expected(h) ? 1 : 0;
            h->far_fresh = false;
            a.far_list = h->d_far_list; a.far_count = h->d_result + 15;      // word 15 of the result block, zeroed below
        }
        const bool sweep_hyb = sweep && sweep_hyb_now;          // LDS windows as far as they reach, the global fields beyond
        const bool sweep_glob = sweep && h->sweep_global && !sweep_hyb;       // probes in global memory: no window in LDS
        // REC: the walk turns the beam direction by the scan's increment instead of fetching it (evenly spaced scans; the beams'
        // offsets from the grid sit in LDS behind the window: 8 bytes per table column)
        // (... as long as two workgroups still fit a CU's 160 KB: up to ~1800 table columns; more beams than that fetch their directions)
        const bool sweep_rec = sweep && h->rec_ok && h->sweep_rec_layout_ok && h->d_beam_err != nullptr &&
                               (size_t)mcl::kSwWinBytes + (size_t)h->ltd_cols * 8 <= 80 * 1024 - 64;
        // PAIRS (two rays per lane) where the walk waits for memory: always in the global-field form; in LDS windows for a set
        // that was set / initialised since the last update (the spread cloud of a re-localisation: -11 % on its first update, where
        // the tracking cloud gains nothing and the levine stand-in loses 2 %); MCL_SWEEP_PAIRS=0 / 1 overrides
        const bool sweep_pairs = sweep_rec && !sweep_hyb && (h->env_sweep_pairs >= 0 ? h->env_sweep_pairs != 0 : (sweep_glob || a.far_windowed != 0));
        size_t qlds = sweep ? (sweep_glob ? 0 : (size_t)mcl::kSwWinBytes) + (sweep_rec ? (size_t)h->ltd_cols * 8 : 0) : (size_t)h->qside * h->qside;
        h->last_sweep_global = sweep_glob ? 1 : (sweep_hyb ? 2 : 0); h->last_sweep_rec = sweep_rec ? 1 : 0; h->last_sweep_pairs = (sweep_rec && (sweep_glob || sweep_pairs)) ? 1 : 0;
        dim3 qg((unsigned)nseg);   // persistent: 2 workgroups per CU
        unsigned long long *d_dbg = nullptr;
        const char *dbgpath = h->env_debug_wg.empty() ? nullptr : h->env_debug_wg.c_str();
        if (dbgpath) { HIPCHK(h, hipMalloc(&d_dbg, (size_t)qg.x * 32)); HIPCHK(h, hipMemset(d_dbg, 0, (size_t)qg.x * 32)); a.dbg = d_dbg; }
        // k_rays_far is bound by global-memory latency: 4 workgroups per CU worth of blocks (2 resident at a time)
        dim3 gfar((unsigned)std::max<int64_t>(1, std::min<int64_t>(4 * (int64_t)h->num_cu, (n + 15) / 16)));
        const int fix_split = std::max(1, std::min(16, (8 * h->num_cu) / std::max(nseg, 1)));   // ~8 workgroups of k_rays_fix per CU (2 .. 16 per segment: no difference, round 4)
        if (!sweep) HIPCHK(h, hipEventRecord(h->ev[EV_K0], h->stream));
        if (count) {
            if (sweep_hyb) hipExtLaunchKernelGGL((mcl::k_rays_sweep<true, false, true, false, true>), qg, b, qlds, h->stream, nullptr, h->ev[EV_K1], 0, a);
            else if (sweep_glob && sweep_rec) hipExtLaunchKernelGGL((mcl::k_rays_sweep<true, true, true, true>), qg, b, qlds, h->stream, nullptr, h->ev[EV_K1], 0, a);
            else if (sweep_glob) hipExtLaunchKernelGGL((mcl::k_rays_sweep<true, true>), qg, b, qlds, h->stream, nullptr, h->ev[EV_K1], 0, a);
            else if (sweep && sweep_rec && sweep_pairs) hipExtLaunchKernelGGL((mcl::k_rays_sweep<true, false, true, true>), qg, b, qlds, h->stream, nullptr, h->ev[EV_K1], 0, a);
            else if (sweep && sweep_rec) hipExtLaunchKernelGGL((mcl::k_rays_sweep<true, false, true>), qg, b, qlds, h->stream, nullptr, h->ev[EV_K1], 0, a);
            else if (sweep) hipExtLaunchKernelGGL((mcl::k_rays_sweep<true>), qg, b, qlds, h->stream, nullptr, h->ev[EV_K1], 0, a);
#ifdef MCL_LEGACY_RAY_KERNELS
            else if (cell) hipLaunchKernelGGL((mcl::k_rays_cell<true>), qg, b, qlds, h->stream, a);
            else hipLaunchKernelGGL((mcl::k_rays_quad<true>), qg, b, qlds, h->stream, a);
#endif
            if (!sweep) HIPCHK(h, hipEventRecord(h->ev[EV_K1], h->stream));
            if (sweep && a.far_windowed) { const int rcw = launch_far_windowed(h, a, n, true); if (rcw) return rcw; }
            hipLaunchKernelGGL((mcl::k_rays_far<true>), gfar, b, 0, h->stream, a);
            hipLaunchKernelGGL((mcl::k_rays_fix<true>), dim3(nseg * fix_split), dim3(256), 0, h->stream, a);
            hipLaunchKernelGGL((mcl::k_rays_exact<true>), dim3(2 * h->num_cu), dim3(256), 0, h->stream, a);
        } else {
            if (sweep_hyb) hipExtLaunchKernelGGL((mcl::k_rays_sweep<false, false, true, false, true>), qg, b, qlds, h->stream, nullptr, h->ev[EV_K1], 0, a);
            else if (sweep_glob && sweep_rec) hipExtLaunchKernelGGL((mcl::k_rays_sweep<false, true, true, true>), qg, b, qlds, h->stream, nullptr, h->ev[EV_K1], 0, a);
            else if (sweep_glob) hipExtLaunchKernelGGL((mcl::k_rays_sweep<false, true>), qg, b, qlds, h->stream, nullptr, h->ev[EV_K1], 0, a);
            else if (sweep && sweep_rec && sweep_pairs) hipExtLaunchKernelGGL((mcl::k_rays_sweep<false, false, true, true>), qg, b, qlds, h->stream, nullptr, h->ev[EV_K1], 0, a);
            else if (sweep && sweep_rec) hipExtLaunchKernelGGL((mcl::k_rays_sweep<false, false, true>), qg, b, qlds, h->stream, nullptr, h->ev[EV_K1], 0, a);
            else if (sweep) hipExtLaunchKernelGGL((mcl::k_rays_sweep<false>), qg, b, qlds, h->stream, nullptr, h->ev[EV_K1], 0, a);
#ifdef MCL_LEGACY_RAY_KERNELS
            else if (cell) hipLaunchKernelGGL((mcl::k_rays_cell<false>), qg, b, qlds, h->stream, a);
            else hipLaunchKernelGGL((mcl::k_rays_quad<false>), qg, b, qlds, h->stream, a);
#endif
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
    if (const char *e = getenv("MCL_SWEEP_HYBRID")) h->env_sweep_hybrid = atoi(e);
    if (const char *e = getenv("MCL_SW_GUIDE")) h->env_sw_guide = atoi(e);
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
#ifdef MCL_LEGACY_RAY_KERNELS
    CRT(hipFuncSetAttribute(reinterpret_cast<const void *>(&mcl::k_rays_quad<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    CRT(hipFuncSetAttribute(reinterpret_cast<const void *>(&mcl::k_rays_quad<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    CRT(hipFuncSetAttribute(reinterpret_cast<const void *>(&mcl::k_rays_cell<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    CRT(hipFuncSetAttribute(reinterpret_cast<const void *>(&mcl::k_rays_cell<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
#endif
    CRT(hipFuncSetAttribute(reinterpret_cast<const void *>(&mcl::k_tiny_tail), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    CRT(hipFuncSetAttribute(reinterpret_cast<const void *>(&mcl::k_sweep_plan), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    CRT(hipFuncSetAttribute(reinterpret_cast<const void *>(&mcl::k_rays_sweep<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    CRT(hipFuncSetAttribute(reinterpret_cast<const void *>(&mcl::k_rays_sweep<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    CRT(hipFuncSetAttribute(reinterpret_cast<const void *>(&mcl::k_rays_sweep<false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024 - 64));
    CRT(hipFuncSetAttribute(reinterpret_cast<const void *>(&mcl::k_rays_sweep<true, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024 - 64));
    CRT(hipFuncSetAttribute(reinterpret_cast<const void *>(&mcl::k_rays_sweep<false, false, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024 - 64));
    CRT(hipFuncSetAttribute(reinterpret_cast<const void *>(&mcl::k_rays_sweep<false, false, true, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024 - 64));
    CRT(hipFuncSetAttribute(reinterpret_cast<const void *>(&mcl::k_rays_sweep<true, false, true, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024 - 64));
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
            h->sweep_hyb_layout_ok = static_lds_is(reinterpret_cast<const void *>(&mcl::k_rays_sweep<false, false, true, false, true>), qb) &&
                                     static_lds_is(reinterpret_cast<const void *>(&mcl::k_rays_sweep<true, false, true, false, true>), qb);
            h->sweep_rec_layout_ok = static_lds_is(reinterpret_cast<const void *>(&mcl::k_rays_sweep<false, false, true>), qb) &&
                                     static_lds_is(reinterpret_cast<const void *>(&mcl::k_rays_sweep<true, false, true>), qb) &&
                                     static_lds_is(reinterpret_cast<const void *>(&mcl::k_rays_sweep<false, false, true, true>), qb) &&
                                     static_lds_is(reinterpret_cast<const void *>(&mcl::k_rays_sweep<true, false, true, true>), qb) &&
                                     static_lds_fits(reinterpret_cast<const void *>(&mcl::k_rays_sweep<false, true, true, true>), 0, qb) &&
                                     static_lds_fits(reinterpret_cast<const void *>(&mcl::k_rays_sweep<true, true, true, true>), 0, qb);
        }
#ifdef MCL_LEGACY_RAY_KERNELS
        h->cell_layout_ok = static_lds_is(reinterpret_cast<const void *>(&mcl::k_rays_cell<false>), qb) && static_lds_is(reinterpret_cast<const void *>(&mcl::k_rays_cell<true>), qb);
        h->quad_layout_ok = static_lds_is(reinterpret_cast<const void *>(&mcl::k_rays_quad<false>), qb) && static_lds_is(reinterpret_cast<const void *>(&mcl::k_rays_quad<true>), qb);
#endif
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
    // (the layout of the NEXT update, made beside this one's ray stage: whether that update takes the hybrid form is not known
    //  yet -- its units are cut for the hybrid's windows, which costs the global-field form nothing but shorter runs)
    const int play = h->env_no_bucket_cuts ? -1 : (choose_ray_mode(h, n, false) == 5 ? sweep_play(h, sweep_hybrid(h)) : 0);
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
        // (the ordering kernels of the ray stage do not read the tables either: the stream waits for them where the ray kernel is
        //  launched -- launch_rays --, not here: 13 us of a 262 144-particle update)
        h->obs_wait_pending = true;
    } else {
        rc = prepare_observation(h, obs, obs_stride);
        if (rc) return rc;
    }
    // (the tables were made beside the ordering: no query-preparation stage on this stream, nothing to time)
    h->ev_query_skipped = obs_early;
    if (!obs_early) HIPCHK(h, hipEventRecord(h->ev[EV_QUERY], h->stream));
    h->ev_rays_bound = false;
    rc = launch_rays(h, h->d_x[h->cur], h->d_y[h->cur], h->d_th[h->cur], n);
    if (h->obs_wait_pending) { h->obs_wait_pending = false; HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_obs, 0)); }      // (a path that launched no windowed kernel)
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


}  // extern "C"

// ---- what mcl_comm.hip / mcl_group.hip reach of this translation unit (mcl_engine_internal.h)
namespace mcl_host {
int fail(mcl_engine *h, int code, const char *msg) { return ::fail(h, code, msg); }
int fail(mcl_engine *h, int code, const std::string &msg) { return ::fail(h, code, msg); }
std::string &create_error() { return g_create_error; }
bool ready(mcl_engine *h, bool need_particles) { return ::ready(h, need_particles); }
float elapsed(hipEvent_t a, hipEvent_t b) { return ::elapsed(a, b); }
void graph_reset(mcl_engine *h) { ::graph_reset(h); }
int scan_weights(mcl_engine *h, const uint64_t *d_q, uint64_t *d_cdf, int64_t n, uint64_t offset, uint64_t *d_total) { return ::scan_weights(h, d_q, d_cdf, n, offset, d_total); }
void unpack_result(mcl_engine *h) { ::unpack_result(h); }
int layout_adopt(mcl_engine *h, int64_t n) { return ::layout_adopt(h, n); }
int set_particles_impl(mcl_engine_t *h, const double *xyz, const double *weights, int64_t n, const double *weight_scale) { return ::set_particles_impl(h, xyz, weights, n, weight_scale); }
int stage_resample_launch(mcl_engine_t *h, const ParentSource &src, const uint64_t *d_cdf, int64_t n_parents, uint64_t q_total,
                          int64_t child_first, int64_t n_children_total, const double action[3])
{
    return ::stage_resample_launch(h, src, d_cdf, n_parents, q_total, child_first, n_children_total, action);
}
int export_compact_launch(mcl_engine_t *h, void *d_chunk, int64_t chunk_entries, int dst_device, hipStream_t stream) { return ::export_compact_launch(h, d_chunk, chunk_entries, dst_device, stream); }
int stage_resample_compact_launch(mcl_engine_t *h, const void *d_chunks, int32_t n_shards, int64_t chunk_entries, const int64_t *counts,
                                  const uint64_t *totals, int64_t n_per_shard, int32_t self_shard, int64_t child_first,
                                  int64_t n_children_total, const double action[3], unsigned long long *remote_count)
{
    return ::stage_resample_compact_launch(h, d_chunks, n_shards, chunk_entries, counts, totals, n_per_shard, self_shard, child_first, n_children_total, action, remote_count);
}
int stage_rays_launch(mcl_engine_t *h, const float *obs, int32_t n_beams, bool force_skip, double *d_max_out) { return ::stage_rays_launch(h, obs, n_beams, force_skip, d_max_out); }
void stage_rays_note(mcl_engine_t *h) { ::stage_rays_note(h); }
int stage_rays_finish(mcl_engine_t *h, const float *obs, int32_t n_beams) { return ::stage_rays_finish(h, obs, n_beams); }
int stage_weights_launch(mcl_engine_t *h, double global_max_logw, const double *d_global_max) { return ::stage_weights_launch(h, global_max_logw, d_global_max); }
int stage_weights_finish(mcl_engine_t *h) { return ::stage_weights_finish(h); }
void stage_commit_carry(mcl_engine_t *h) { ::stage_commit_carry(h); }
void launch_copy_double(hipStream_t stream, const double *src, double *dst) { hipLaunchKernelGGL(mcl::k_copy_double, dim3(1), dim3(1), 0, stream, src, dst); }
void launch_set_double(hipStream_t stream, double *dst, double v) { hipLaunchKernelGGL(mcl::k_set_double, dim3(1), dim3(1), 0, stream, dst, v); }
void launch_spin_ms(hipStream_t stream, double ms) { hipLaunchKernelGGL(mcl::k_spin_ms, dim3(1), dim3(64), 0, stream, ms); }
void launch_stage_pack(hipStream_t stream, const unsigned long long *d_result, double *d_vec, int n_shards, int self, int listed, unsigned long long list_cap)
{
    hipLaunchKernelGGL(mcl::k_stage_pack, dim3(1), dim3(64), 0, stream, d_result, d_vec, n_shards, self, listed, list_cap);
}
void launch_pack_records(hipStream_t stream, const double *x, const double *y, const double *th, int64_t n, double4 *out)
{
    hipLaunchKernelGGL(mcl::k_pack_records, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, x, y, th, n, out);
}
void launch_group_max(hipStream_t stream, const mcl::GroupMaxArgs &a) { hipLaunchKernelGGL(mcl::k_group_max, dim3(1), dim3(1), 0, stream, a); }
}  // namespace mcl_host

