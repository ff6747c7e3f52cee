/*
 * mcl_oracle.c — CPU ORACLE for the particle-filter update path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under monte_carlo_localization_amd/ (the product)
 * may include, link, import or execute this file.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg use it, and only as the checker / the timed CPU
 * baseline, never as the thing shipped.
 *
 * Two families of functions live here:
 *
 *   orc_ref_*  — a plain-C restatement of the reference's algorithm, statement by
 *                statement, in the reference's own arithmetic types (double state, float
 *                ranges, int indices), with every random draw INJECTED by the caller so
 *                that the libstdc++ <random> stream the reference consumes can be replayed
 *                (oracle/refdraws.cpp produces that stream).  Citations "cpp:N" are
 *                /root/reference/src/particle_filter.cpp line numbers, "utils:N" are
 *                /root/reference/src/utils.cpp.
 *
 *   orc_eng_*  — a scalar restatement of the ENGINE SPEC: log-domain
 *                weights from an fp32 log table (exact fp64 sums), deterministic exp,
 *                fixed-point CDF, Philox4x32-10 draws.  These exist so that the HIP path can
 *                be checked bit-for-bit at sizes/beam counts where the reference's own
 *                arithmetic degenerates or is not parallelisable (D6).
 *
 * PARITY UNPINNED.  The reference holds no tests, fixtures or golden vectors for this path
 * (CMakeLists.txt:126-133 enables lint only) and cannot be built here: src/particle_filter.cpp
 * needs rclcpp / tf2 / nav_msgs / Eigen headers, none of which exist in this image, and a build
 * against stand-in headers is not an admissible pin -- so oracle/_ref is NOT built and nothing
 * the reference itself executed here backs this file.  What it does reproduce, to all 17 digits,
 * are the known answers SURVEY.md Appendix B recorded from the survey's stub-header run of the
 * reference (sensor table entries, synthetic scans, a full seeded MCL chain incl. the libstdc++
 * draws): tests/test_oracle_known_answers.py keeps those as a regression net, not as a pin.
 * Everything else rests on reading: orc_ref_* follows cpp:233-292, 449-503, 506-650, 652-716
 * statement by statement, each function citing its lines.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ------------------------------------------------------------------------------------------
 * Map description shared by all functions (what cpp:190-195 derives from the OccupancyGrid).
 * resolution is the float32 of MapMetaData widened to double.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    const int8_t *data; /* row-major H x W, row 0 = bottom; values {-1,0,100} */
    int32_t width, height;
    double resolution;  /* (double)(float)yaml_resolution                      */
    double origin_x, origin_y;
    double max_range_m; /* MAX_RANGE_METERS                                    */
    int32_t max_range_px; /* MAX_RANGE_PX = (int)(max_range_m / resolution), cpp:195 */
} orc_map_t;

int32_t orc_max_range_px(double max_range_m, float resolution_f32)
{
    double res = (double)resolution_f32;            /* cpp:191 */
    return (int32_t)(max_range_m / res);            /* cpp:195 */
}

/* utils:43-48 */
double orc_normalize_angle(double angle)
{
    while (angle > M_PI) angle -= 2.0 * M_PI;
    while (angle < -M_PI) angle += 2.0 * M_PI;
    return angle;
}

/* ------------------------------------------------------------------------------------------
 * Sensor table, cpp:233-292.  out is (P+1)^2 doubles, Eigen column-major: out[d*(P+1)+r]
 * with r = observed px, d = expected px.
 * ------------------------------------------------------------------------------------------ */
void orc_ref_sensor_table(int32_t P, double z_hit, double z_short, double z_max, double z_rand,
                          double sigma_hit, double *out)
{
    int tw = P + 1;
    for (int d = 0; d < tw; ++d) {
        double norm = 0.0;
        for (int r = 0; r < tw; ++r) {
            double prob = 0.0;
            double z = (double)(r - d);
            prob += z_hit * exp(-(z * z) / (2.0 * sigma_hit * sigma_hit)) / (sigma_hit * sqrt(2.0 * M_PI)); /* cpp:258 */
            if (r < d) prob += 2.0 * z_short * (d - r) / (double)d;   /* cpp:261-264 */
            if (r == P) prob += z_max;                                /* cpp:267-270 */
            if (r < P) prob += z_rand * 1.0 / (double)P;              /* cpp:273-276 */
            norm += prob;
            out[(size_t)d * tw + r] = prob;
        }
        if (norm > 0) {
            for (int r = 0; r < tw; ++r) out[(size_t)d * tw + r] /= norm;  /* cpp:283-286 */
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * cast_ray, cpp:611-650.  Returns the float range; *step_out receives the integer the
 * reference multiplies by the resolution (0..P-1), or P when the loop runs out (cpp:649).
 * ------------------------------------------------------------------------------------------ */
float orc_ref_cast_ray(const orc_map_t *m, double x, double y, double angle, int32_t *step_out)
{
    double dx = cos(angle) * m->resolution;   /* cpp:616 */
    double dy = sin(angle) * m->resolution;   /* cpp:617 */
    double cx = x, cy = y;
    for (int step = 0; step < m->max_range_px; ++step) {
        cx += dx;
        cy += dy;
        int gx = (int)((cx - m->origin_x) / m->resolution);  /* cpp:628 truncation */
        int gy = (int)((cy - m->origin_y) / m->resolution);
        if (gx < 0 || gx >= m->width || gy < 0 || gy >= m->height) {  /* cpp:632-636 */
            if (step_out) *step_out = step;
            return (float)(step * m->resolution);
        }
        int idx = gy * m->width + gx;
        if (m->data[idx] > 50) {                                       /* cpp:642 */
            if (step_out) *step_out = step;
            return (float)(step * m->resolution);
        }
    }
    if (step_out) *step_out = m->max_range_px;
    return (float)m->max_range_m;                                      /* cpp:649 */
}

/* Convenience: many rays (x,y,angle) -> ranges + steps; same omp pragma as cpp:593. */
void orc_ref_cast_many(const orc_map_t *m, int64_t n, const double *x, const double *y,
                       const double *angle, float *ranges, int32_t *steps, int use_omp)
{
    if (use_omp) {
#pragma omp parallel for schedule(dynamic)
        for (int64_t i = 0; i < n; ++i) {
            int32_t s;
            float r = orc_ref_cast_ray(m, x[i], y[i], angle[i], &s);
            if (ranges) ranges[i] = r;
            if (steps) steps[i] = s;
        }
    } else {
        for (int64_t i = 0; i < n; ++i) {
            int32_t s;
            float r = orc_ref_cast_ray(m, x[i], y[i], angle[i], &s);
            if (ranges) ranges[i] = r;
            if (steps) steps[i] = s;
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * motion_model, cpp:449-503.  p is N x 3 column-major (x col, y col, theta col) like
 * Eigen::MatrixXd; normals is N x 3 ROW-major in draw order (n_x, n_y, n_theta per particle,
 * cpp:496-498).
 * ------------------------------------------------------------------------------------------ */
void orc_ref_motion_scalars(const double action[3], double *dt_out, double *v_out, double *w_out)
{
    double dt = 0.01, velocity = 0.0, angular_velocity = 0.0;
    double fd = action[0], ad = action[2];
    if (fabs(fd) > 0.001) {                 /* cpp:459 */
        if (fabs(fd) < 0.1) dt = fabs(fd) / 1.0;
        else dt = fabs(fd) / 5.0;
        dt = fmax(0.001, fmin(dt, 0.1));    /* cpp:465 */
        velocity = fd / dt;
    }
    if (fabs(ad) > 0.001) angular_velocity = ad / dt;  /* cpp:469-471 */
    *dt_out = dt; *v_out = velocity; *w_out = angular_velocity;
}

void orc_ref_motion_model(int64_t N, double *p, const double action[3], const double *normals,
                          double disp_x, double disp_y, double disp_theta)
{
    double dt, velocity, angular_velocity;
    orc_ref_motion_scalars(action, &dt, &velocity, &angular_velocity);
    double *X = p, *Y = p + N, *T = p + 2 * N;
    for (int64_t i = 0; i < N; ++i) {
        double x = X[i], y = Y[i], theta = T[i];
        if (fabs(angular_velocity) < 1e-6) {                         /* cpp:480-484 */
            X[i] = x + velocity * dt * cos(theta);
            Y[i] = y + velocity * dt * sin(theta);
            T[i] = theta;
        } else {                                                     /* cpp:485-493 */
            double radius = velocity / angular_velocity;
            double delta_theta = angular_velocity * dt;
            X[i] = x + radius * (sin(theta + delta_theta) - sin(theta));
            Y[i] = y - radius * (cos(theta + delta_theta) - cos(theta));
            T[i] = theta + delta_theta;
        }
        X[i] += normals[3 * i + 0] * disp_x;                         /* cpp:496-498 */
        Y[i] += normals[3 * i + 1] * disp_y;
        T[i] += normals[3 * i + 2] * disp_theta;
        T[i] = orc_normalize_angle(T[i]);                            /* cpp:501 */
    }
}

/* ------------------------------------------------------------------------------------------
 * obs -> obs_idx, cpp:549-554, 570, 573.  NaN maps to 0 (x86 cvttss2si result clamped,
 * SURVEY row E).
 * ------------------------------------------------------------------------------------------ */
void orc_ref_obs_index(const float *obs, int32_t B, double resolution, int32_t P, int32_t *obs_idx)
{
    for (int j = 0; j < B; ++j) {
        float px = (float)((double)obs[j] / resolution);   /* cpp:551 */
        if (px > (float)P) px = (float)P;                   /* cpp:552-553 */
        float r = roundf(px);                               /* cpp:570 */
        int idx;
        if (r != r) idx = 0;
        else if (r <= -2147483648.0f) idx = 0;
        else idx = (int)r;
        if (idx > P) idx = P;
        if (idx < 0) idx = 0;                               /* cpp:573 */
        obs_idx[j] = idx;
    }
}

/* ------------------------------------------------------------------------------------------
 * sensor_model, cpp:506-583, with the same materialised arrays (queries N*B x 3 doubles
 * column-major, ranges / ranges_px float) so that its cost profile matches the reference.
 * weights_out[i] = pow(prod_j T(obs_idx_j, range_idx_ij), inv_squash)   (cpp:566-578)
 * steps_out (optional, N*B) receives range_idx.
 * timing_ms (optional, 3 doubles): query prep, ray casting, table evaluation.
 * ------------------------------------------------------------------------------------------ */
static double now_ms(void)
{
#ifdef _OPENMP
    return omp_get_wtime() * 1e3;
#else
    return 0.0;
#endif
}

int orc_ref_sensor_model(const orc_map_t *m, int64_t N, const double *p /* N x 3 col-major */,
                         int32_t B, const float *angles, const float *obs,
                         const double *table /* (P+1)^2 col-major */, double inv_squash,
                         double *weights_out, int32_t *steps_out, int use_omp, double *timing_ms)
{
    const int P = m->max_range_px;
    const int tw = P + 1;
    int64_t R = N * (int64_t)B;
    if (R >= (int64_t)1 << 31) return -1; /* reference indexes rays with int */
    double *queries = (double *)malloc((size_t)R * 3 * sizeof(double));  /* cpp:514 */
    float *ranges = (float *)malloc((size_t)R * sizeof(float));          /* cpp:515/590 */
    float *ranges_px = (float *)malloc((size_t)R * sizeof(float));       /* cpp:547 */
    float *obs_px = (float *)malloc((size_t)B * sizeof(float));          /* cpp:546 */
    if (!queries || !ranges || !ranges_px || !obs_px) { free(queries); free(ranges); free(ranges_px); free(obs_px); return -2; }
    const double *X = p, *Y = p + N, *T = p + 2 * N;

    double t0 = now_ms();
    for (int64_t i = 0; i < N; ++i) {                                    /* cpp:526-535 */
        for (int j = 0; j < B; ++j) {
            int64_t idx = i * B + j;
            queries[idx] = X[i];
            queries[R + idx] = Y[i];
            queries[2 * R + idx] = T[i] + (double)angles[j];
        }
    }
    double t1 = now_ms();
    if (use_omp) {                                                       /* cpp:592-603 */
#pragma omp parallel for schedule(dynamic)
        for (int64_t i = 0; i < R; ++i)
            ranges[i] = orc_ref_cast_ray(m, queries[i], queries[R + i], queries[2 * R + i], NULL);
    } else {
        for (int64_t i = 0; i < R; ++i)
            ranges[i] = orc_ref_cast_ray(m, queries[i], queries[R + i], queries[2 * R + i], NULL);
    }
    double t2 = now_ms();
    for (int j = 0; j < B; ++j) {                                        /* cpp:549-554 */
        obs_px[j] = (float)((double)obs[j] / m->resolution);
        if (obs_px[j] > (float)P) obs_px[j] = (float)P;
    }
    for (int64_t i = 0; i < R; ++i) {                                    /* cpp:556-561 */
        ranges_px[i] = (float)((double)ranges[i] / m->resolution);
        if (ranges_px[i] > (float)P) ranges_px[i] = (float)P;
    }
    for (int64_t i = 0; i < N; ++i) {                                    /* cpp:564-579 */
        double weight = 1.0;
        for (int j = 0; j < B; ++j) {
            float ro = roundf(obs_px[j]);
            int obs_idx = (ro != ro) ? 0 : (int)ro;
            int range_idx = (int)roundf(ranges_px[i * B + j]);
            if (obs_idx > P) obs_idx = P;
            if (obs_idx < 0) obs_idx = 0;
            if (range_idx > P) range_idx = P;
            if (range_idx < 0) range_idx = 0;
            weight *= table[(size_t)range_idx * tw + obs_idx];           /* (r=obs_idx, d=range_idx) */
            if (steps_out) steps_out[i * B + j] = range_idx;
        }
        weights_out[i] = pow(weight, inv_squash);                        /* cpp:578 */
    }
    double t3 = now_ms();
    if (timing_ms) { timing_ms[0] = t1 - t0; timing_ms[1] = t2 - t1; timing_ms[2] = t3 - t2; }
    free(queries); free(ranges); free(ranges_px); free(obs_px);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Multinomial resampling exactly as std::discrete_distribution<int> does it
 * (cpp:658-665; libstdc++ bits/random.tcc _M_initialize + operator()):
 *   p_i = w_i / sum(w) (sequential accumulate), cp = partial_sum(p), cp[N-1] = 1.0,
 *   idx = lower_bound(cp, u) - cp.begin(), u = generate_canonical<double,53>(rng) injected.
 * ------------------------------------------------------------------------------------------ */
void orc_ref_resample_indices(int64_t N, const double *weights, const double *uniforms, int32_t *idx_out)
{
    double *cp = (double *)malloc((size_t)N * sizeof(double));
    double sum = 0.0;
    for (int64_t i = 0; i < N; ++i) sum += weights[i];
    double acc = 0.0;
    for (int64_t i = 0; i < N; ++i) { acc += weights[i] / sum; cp[i] = acc; }
    cp[N - 1] = 1.0;
    for (int64_t k = 0; k < N; ++k) {
        double u = uniforms[k];
        int64_t lo = 0, len = N;           /* std::lower_bound: first cp[i] with !(cp[i] < u) */
        while (len > 0) {
            int64_t half = len >> 1, mid = lo + half;
            if (cp[mid] < u) { lo = mid + 1; len = len - half - 1; }
            else len = half;
        }
        idx_out[k] = (int32_t)lo;
    }
    free(cp);
}

/* normalise, cpp:678-686 */
double orc_ref_normalize(int64_t N, double *w)
{
    double s = 0.0;
    for (int64_t i = 0; i < N; ++i) s += w[i];
    if (s > 0) for (int64_t i = 0; i < N; ++i) w[i] /= s;
    return s;
}

/* expected_pose, cpp:696-716 */
void orc_ref_expected_pose(int64_t N, const double *p, const double *w, double out[3])
{
    const double *X = p, *Y = p + N, *T = p + 2 * N;
    double px = 0.0, py = 0.0, ss = 0.0, sc = 0.0;
    for (int64_t i = 0; i < N; ++i) {
        px += w[i] * X[i];
        py += w[i] * Y[i];
        ss += w[i] * sin(T[i]);
        sc += w[i] * cos(T[i]);
    }
    out[0] = px; out[1] = py; out[2] = atan2(ss, sc);
}

/* ------------------------------------------------------------------------------------------
 * One full MCL step, cpp:652-694, with injected draws (uniforms N for the resample, then
 * normals N x 3 for the motion model — the order the shared rng_ is consumed in).
 * particles (N x 3 col-major) and weights are updated in place; optional outputs:
 * idx_out N, steps_out N*B, raw_weights_out N (before normalisation).
 * timing_ms (optional, 6): resample, motion, query, raycast, table, total (TimingStats order,
 * utils.hpp:51-57).
 * ------------------------------------------------------------------------------------------ */
int orc_ref_mcl_step(const orc_map_t *m, int64_t N, double *particles, double *weights,
                     const double action[3], int32_t B, const float *angles, const float *obs,
                     const double *table, double inv_squash,
                     const double *uniforms, const double *normals,
                     double disp_x, double disp_y, double disp_theta,
                     int32_t *idx_out, int32_t *steps_out, double *raw_weights_out,
                     int use_omp, double *timing_ms)
{
    double T0 = now_ms();
    int32_t *idx = (int32_t *)malloc((size_t)N * sizeof(int32_t));
    double *prop = (double *)malloc((size_t)N * 3 * sizeof(double));     /* cpp:659 */
    if (!idx || !prop) { free(idx); free(prop); return -2; }
    orc_ref_resample_indices(N, weights, uniforms, idx);                 /* cpp:658,663 */
    for (int64_t i = 0; i < N; ++i) {                                    /* cpp:664 */
        prop[i] = particles[idx[i]];
        prop[N + i] = particles[N + idx[i]];
        prop[2 * N + i] = particles[2 * N + idx[i]];
    }
    double T1 = now_ms();
    orc_ref_motion_model(N, prop, action, normals, disp_x, disp_y, disp_theta);   /* cpp:671 */
    double T2 = now_ms();
    double sm[3] = {0, 0, 0};
    int rc = orc_ref_sensor_model(m, N, prop, B, angles, obs, table, inv_squash, weights, steps_out, use_omp, sm); /* cpp:676 */
    if (rc) { free(idx); free(prop); return rc; }
    if (raw_weights_out) memcpy(raw_weights_out, weights, (size_t)N * sizeof(double));
    orc_ref_normalize(N, weights);                                       /* cpp:679-686 */
    memcpy(particles, prop, (size_t)N * 3 * sizeof(double));             /* cpp:689 */
    double T3 = now_ms();
    if (idx_out) memcpy(idx_out, idx, (size_t)N * sizeof(int32_t));
    if (timing_ms) {
        timing_ms[0] = T1 - T0; timing_ms[1] = T2 - T1; timing_ms[2] = sm[0];
        timing_ms[3] = sm[1]; timing_ms[4] = sm[2]; timing_ms[5] = T3 - T0;
    }
    free(idx); free(prop);
    return 0;
}

int orc_omp_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
    return omp_get_max_threads();
#else
    (void)n; return 1;
#endif
}

/* ==========================================================================================
 * ENGINE-SPEC restatement.  Scalar, order-independent where the spec is.
 * ========================================================================================== */

/* E1: fp32 log table, row-major by OBSERVED index: L[r*(P+1)+d] = (float)(log(T(r,d))*inv_squash) */
void orc_eng_log_table(int32_t P, const double *table, double inv_squash, float *L)
{
    int tw = P + 1;
    for (int r = 0; r < tw; ++r)
        for (int d = 0; d < tw; ++d)
            L[(size_t)r * tw + d] = (float)(log(table[(size_t)d * tw + r]) * inv_squash);
}

/* E3/E4: log-weights for a particle set (no materialised arrays; chunked, any N).
 * logw_out[i] = sum_j (double)L[obs_idx_j][step_ij]; optionally steps (uint8) and the total
 * number of grid probes (for S-bar). */
void orc_eng_log_weights(const orc_map_t *m, int64_t N, const double *p, int32_t B, const float *angles,
                         const int32_t *obs_idx, const float *L, double *logw_out, uint8_t *steps_out,
                         int64_t *probes_out, int use_omp)
{
    const int tw = m->max_range_px + 1;
    const double *X = p, *Y = p + N, *T = p + 2 * N;
    int64_t probes = 0;
#pragma omp parallel for schedule(static) reduction(+ : probes) if (use_omp)
    for (int64_t i = 0; i < N; ++i) {
        double acc = 0.0;
        for (int j = 0; j < B; ++j) {
            int32_t s;
            orc_ref_cast_ray(m, X[i], Y[i], T[i] + (double)angles[j], &s);
            acc += (double)L[(size_t)obs_idx[j] * tw + s];
            if (steps_out) steps_out[i * B + j] = (uint8_t)s;
            probes += (s < m->max_range_px) ? (s + 1) : m->max_range_px;
        }
        logw_out[i] = acc;
    }
    if (probes_out) *probes_out = probes;
}

/* E5: deterministic exp for x <= 0 (same operation sequence as the device function
 * mcl_det_exp in csrc/mcl_device_math.h; every multiply-add is an explicit fma so host and
 * device round identically).  Returns 0 for x < -745. */
double orc_eng_det_exp(double x)
{
    if (!(x > -745.0)) return (x != x) ? x : 0.0;
    if (x > 0.0) x = 0.0;
    const double LOG2E = 1.4426950408889634074;
    const double LN2_HI = 6.93147180369123816490e-01;
    const double LN2_LO = 1.90821492927058770002e-10;
    double kf = nearbyint(x * LOG2E);
    double r = fma(-kf, LN2_HI, x);
    r = fma(-kf, LN2_LO, r);
    /* Taylor/Horner degree 13 on |r| <= 0.3466 : truncation < 2^-58 relative */
    double p = 1.0 / 6227020800.0;
    p = fma(p, r, 1.0 / 479001600.0);
    p = fma(p, r, 1.0 / 39916800.0);
    p = fma(p, r, 1.0 / 3628800.0);
    p = fma(p, r, 1.0 / 362880.0);
    p = fma(p, r, 1.0 / 40320.0);
    p = fma(p, r, 1.0 / 5040.0);
    p = fma(p, r, 1.0 / 720.0);
    p = fma(p, r, 1.0 / 120.0);
    p = fma(p, r, 1.0 / 24.0);
    p = fma(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    int k = (int)kf;
    /* scale by 2^k in two exact steps so that denormal results round once */
    if (k < -1000) { p = p * 0x1p-1000; k += 1000; }
    union { uint64_t u; double d; } s;
    s.u = (uint64_t)(k + 1023) << 52;
    return p * s.d;
}

#define ORC_WEIGHT_FRAC_BITS 36

/* E5: w_i = det_exp(logw_i - max), q_i = floor(w_i * 2^36).  Returns max. */
double orc_eng_weights_from_log(int64_t N, const double *logw, double *w_out, uint64_t *q_out)
{
    double mx = -INFINITY;
    for (int64_t i = 0; i < N; ++i) if (logw[i] > mx) mx = logw[i];
    for (int64_t i = 0; i < N; ++i) {
        double w = orc_eng_det_exp(logw[i] - mx);
        if (w_out) w_out[i] = w;
        if (q_out) q_out[i] = (uint64_t)(w * 68719476736.0);
    }
    return mx;
}

/* quantise externally supplied (already >= 0) weights: q_i = floor(w_i / max(w) * 2^36) */
void orc_eng_quantize_weights(int64_t N, const double *w, uint64_t *q_out)
{
    double mx = 0.0;
    for (int64_t i = 0; i < N; ++i) if (w[i] > mx) mx = w[i];
    for (int64_t i = 0; i < N; ++i) {
        double v = (mx > 0.0 && w[i] > 0.0) ? (w[i] / mx) : 0.0;
        q_out[i] = (uint64_t)(v * 68719476736.0);
    }
}

typedef unsigned __int128 u128;

/* E6: index search on the exact integer CDF.  mode 0 = multinomial with 53-bit k_m,
 * mode 1 = systematic with a single 32-bit k0.  n_children outputs. */
void orc_eng_resample_indices(int64_t N, const uint64_t *q, int mode, int64_t n_children,
                              const uint64_t *k53 /* n_children (mode 0) */, uint32_t k0 /* mode 1 */,
                              int32_t *idx_out)
{
    uint64_t *C = (uint64_t *)malloc((size_t)N * sizeof(uint64_t));
    uint64_t acc = 0;
    for (int64_t i = 0; i < N; ++i) { acc += q[i]; C[i] = acc; }
    uint64_t Q = acc;
    /* every child's search is independent of the others: the loop may run on several threads (the checker is asked for all
     * 33 554 432 children of BASELINE config #5) without changing a single result */
#pragma omp parallel for schedule(static)
    for (int64_t mth = 0; mth < n_children; ++mth) {
        if (Q == 0) { idx_out[mth] = 0; continue; }
        u128 rhs, lmul;
        if (mode == 0) { rhs = (u128)k53[mth] * Q; lmul = (u128)1 << 53; }
        else { rhs = ((u128)(uint64_t)mth * 4294967296ull + k0) * Q; lmul = (u128)(uint64_t)n_children * 4294967296ull; }
        int64_t lo = 0, len = N;         /* first i with C_i * lmul > rhs */
        while (len > 0) {
            int64_t half = len >> 1, mid = lo + half;
            if (!((u128)C[mid] * lmul > rhs)) { lo = mid + 1; len = len - half - 1; }
            else len = half;
        }
        if (lo >= N) lo = N - 1;
        idx_out[mth] = (int32_t)lo;
    }
    free(C);
}

/* E7: Philox4x32-10 (Salmon et al. 2011), key = (seed_lo, seed_hi). */
void orc_eng_philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                        uint32_t out[4])
{
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
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static uint64_t bits53(uint32_t a, uint32_t b) { return (((uint64_t)a << 32) | b) >> 11; }

/* multinomial uniforms (stream 2) for children [first, first+n) of update `upd` */
void orc_eng_philox_k53(uint64_t seed, uint32_t upd, int64_t first, int64_t n, uint64_t *k53)
{
    for (int64_t i = 0; i < n; ++i) {
        uint32_t o[4];
        uint64_t g = (uint64_t)(first + i);
        orc_eng_philox4x32((uint32_t)g, upd, 2u, (uint32_t)(g >> 32), (uint32_t)seed, (uint32_t)(seed >> 32), o);
        k53[i] = bits53(o[0], o[1]);
    }
}

uint32_t orc_eng_philox_k0(uint64_t seed, uint32_t upd)
{
    uint32_t o[4];
    orc_eng_philox4x32(0u, upd, 3u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), o);
    return o[0];
}

/* motion normals (streams 0 and 1), Box-Muller in fp64; normals N x 3 row-major */
void orc_eng_philox_normals(uint64_t seed, uint32_t upd, int64_t first, int64_t n, double *normals)
{
    const double TWO_M53 = 1.0 / 9007199254740992.0;
    for (int64_t i = 0; i < n; ++i) {
        uint32_t o[4];
        uint64_t g = (uint64_t)(first + i);
        orc_eng_philox4x32((uint32_t)g, upd, 0u, (uint32_t)(g >> 32), (uint32_t)seed, (uint32_t)(seed >> 32), o);
        double u1 = (double)(bits53(o[0], o[1]) + 1) * TWO_M53;
        double u2 = (double)bits53(o[2], o[3]) * TWO_M53;
        double rad = sqrt(-2.0 * log(u1));
        normals[3 * i + 0] = rad * cos(2.0 * M_PI * u2);
        normals[3 * i + 1] = rad * sin(2.0 * M_PI * u2);
        orc_eng_philox4x32((uint32_t)g, upd, 1u, (uint32_t)(g >> 32), (uint32_t)seed, (uint32_t)(seed >> 32), o);
        u1 = (double)(bits53(o[0], o[1]) + 1) * TWO_M53;
        u2 = (double)bits53(o[2], o[3]) * TWO_M53;
        rad = sqrt(-2.0 * log(u1));
        normals[3 * i + 2] = rad * cos(2.0 * M_PI * u2);
    }
}

/* device-side initialisers (engine spec): Philox streams 5/6 and 7; out is N x 3 column-major */
void orc_eng_init_pose(uint64_t seed, uint32_t init_idx, const double pose[3], int64_t first, int64_t n, double *out)
{
    const double TWO_M53 = 1.0 / 9007199254740992.0;
    for (int64_t i = 0; i < n; ++i) {
        uint32_t o[4];
        uint64_t g = (uint64_t)(first + i);
        orc_eng_philox4x32((uint32_t)g, init_idx, 5u, (uint32_t)(g >> 32), (uint32_t)seed, (uint32_t)(seed >> 32), o);
        double u1 = (double)(bits53(o[0], o[1]) + 1) * TWO_M53, u2 = (double)bits53(o[2], o[3]) * TWO_M53;
        double rad = sqrt(-2.0 * log(u1));
        double n0 = rad * cos(2.0 * M_PI * u2), n1 = rad * sin(2.0 * M_PI * u2);
        orc_eng_philox4x32((uint32_t)g, init_idx, 6u, (uint32_t)(g >> 32), (uint32_t)seed, (uint32_t)(seed >> 32), o);
        u1 = (double)(bits53(o[0], o[1]) + 1) * TWO_M53; u2 = (double)bits53(o[2], o[3]) * TWO_M53;
        double n2 = sqrt(-2.0 * log(u1)) * cos(2.0 * M_PI * u2);
        out[i] = pose[0] + n0 * 0.5;                      /* cpp:392 */
        out[n + i] = pose[1] + n1 * 0.5;                  /* cpp:393 */
        out[2 * n + i] = orc_normalize_angle(pose[2] + n2 * 0.4);   /* cpp:394-397 */
    }
}

int orc_eng_init_global(uint64_t seed, uint32_t init_idx, const orc_map_t *m, int64_t first, int64_t n, double *out)
{
    /* permissible_positions in the reference's order (cpp:412-421): rows outer, columns inner, data == 0 */
    size_t cells = (size_t)m->width * m->height, nf = 0;
    uint32_t *fr = (uint32_t *)malloc(cells * sizeof(uint32_t));
    for (size_t i = 0; i < cells; ++i) if (m->data[i] == 0) fr[nf++] = (uint32_t)i;
    if (!nf) { free(fr); return -1; }
    for (int64_t i = 0; i < n; ++i) {
        uint32_t o[4];
        uint64_t g = (uint64_t)(first + i);
        orc_eng_philox4x32((uint32_t)g, init_idx, 7u, (uint32_t)(g >> 32), (uint32_t)seed, (uint32_t)(seed >> 32), o);
        uint64_t k = bits53(o[0], o[1]);
        uint64_t pick = (uint64_t)(((u128)(k << 11) * (u128)nf) >> 64);
        uint32_t cell = fr[pick];
        int row = (int)(cell / (uint32_t)m->width), col = (int)(cell % (uint32_t)m->width);
        out[i] = col * m->resolution + m->origin_x;       /* cpp:438 */
        out[n + i] = row * m->resolution + m->origin_y;   /* cpp:439 */
        out[2 * n + i] = (double)bits53(o[2], o[3]) * (1.0 / 9007199254740992.0) * (2.0 * M_PI);   /* cpp:431,440 */
    }
    free(fr);
    return 0;
}

/* Chebyshev distance-to-stop field on the padded grid, restated naively
 * for cross-checking the engine's host-side builder on small maps.
 * Padded grid: (W+1+pad_hi) x (H+1+pad_hi) ... see python wrapper; here we only give the
 * brute-force distance for one cell list. stop[] is row-major Hp x Wp of 0/1. */
void orc_eng_chebyshev_bruteforce(int32_t Wp, int32_t Hp, const uint8_t *stop, int32_t cap, uint8_t *dist)
{
    for (int y = 0; y < Hp; ++y)
        for (int x = 0; x < Wp; ++x) {
            int best = cap;
            for (int r = 0; r < cap && best == cap; ++r) {
                int y0 = y - r, y1 = y + r, x0 = x - r, x1 = x + r;
                for (int yy = y0; yy <= y1 && best == cap; ++yy)
                    for (int xx = x0; xx <= x1; ++xx) {
                        if (yy != y0 && yy != y1 && xx != x0 && xx != x1) continue;
                        int s = (yy < 0 || yy >= Hp || xx < 0 || xx >= Wp) ? 1 : stop[(size_t)yy * Wp + xx];
                        if (s) { best = r; break; }
                    }
            }
            dist[(size_t)y * Wp + x] = (uint8_t)best;
        }
}
