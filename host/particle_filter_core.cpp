// particle_filter_core.cpp — see the header.  The three functions whose bodies the patch replaces are
// MCL(), expected_pose() and the state hand-over at the end of the two initialisers; the initialisers'
// own arithmetic (host rng_, cpp:390-398 / 433-441) is kept as the reference has it.
#include "particle_filter_core.hpp"

#include <cmath>
#include <cstdio>
#include <limits>

namespace particle_filter_cpp {

namespace {
double normalize_angle(double angle)   // utils.cpp:43-48
{
    while (angle > M_PI) angle -= 2.0 * M_PI;
    while (angle < -M_PI) angle += 2.0 * M_PI;
    return angle;
}
}  // namespace

ParticleFilterCore::ParticleFilterCore(const Params &p, std::function<void(const std::string &)> log_error)
    : MAX_PARTICLES(p.max_particles), rng_(p.seed ? (uint32_t)p.seed : std::random_device{}()), params_(p),
      log_error_(std::move(log_error))
{
    mcl_config_t cfg;
    mcl_default_config(&cfg);
    cfg.max_particles = p.max_particles;
    cfg.device = p.device;
    cfg.seed = p.seed;
    cfg.max_range_m = p.max_range;
    cfg.z_hit = p.z_hit; cfg.z_short = p.z_short; cfg.z_max = p.z_max; cfg.z_rand = p.z_rand; cfg.sigma_hit = p.sigma_hit;
    cfg.squash_factor = p.squash_factor;
    cfg.motion_dispersion_x = p.motion_dispersion_x;
    cfg.motion_dispersion_y = p.motion_dispersion_y;
    cfg.motion_dispersion_theta = p.motion_dispersion_theta;
    if (mcl_create(&cfg, &engine_) != MCL_OK) {
        engine_ = nullptr;
        if (log_error_) log_error_(std::string("mcl_create: ") + mcl_last_error(nullptr));
    }
    particles_.resize(MAX_PARTICLES);                         // cpp:106
    weights_.assign(MAX_PARTICLES, 1.0 / MAX_PARTICLES);      // cpp:107
}

ParticleFilterCore::~ParticleFilterCore() { mcl_destroy(engine_); }

void ParticleFilterCore::fail(const char *what)
{
    if (log_error_) log_error_(std::string(what) + ": " + (engine_ ? mcl_last_error(engine_) : "no engine"));
}

void ParticleFilterCore::set_map(const OccupancyGrid &map)
{
    map_ = map;
    if (!engine_) return;
    if (mcl_set_map(engine_, map_.data.data(), map_.width, map_.height, map_.resolution, map_.origin_x, map_.origin_y) != MCL_OK) {
        fail("mcl_set_map");                                  // cpp:236-240 logs and returns
        return;
    }
    int32_t P = 0;
    mcl_get_max_range_px(engine_, &P);
    MAX_RANGE_PX = P;                                         // cpp:195
    permissible_positions_.clear();                           // cpp:199-213, 411-421
    for (uint32_t i = 0; i < map_.height; ++i)
        for (uint32_t j = 0; j < map_.width; ++j)
            if (map_.data[(size_t)i * map_.width + j] == 0) permissible_positions_.emplace_back((int)i, (int)j);
    map_initialized_ = true;
}

void ParticleFilterCore::set_downsampled_angles(const std::vector<float> &angles)
{
    downsampled_angles_ = angles;
    if (engine_ && mcl_set_beam_angles(engine_, angles.data(), (int32_t)angles.size()) != MCL_OK) fail("mcl_set_beam_angles");
}

void ParticleFilterCore::push_state()
{
    if (!engine_) return;
    if (mcl_set_particles(engine_, particles_.data.data(), weights_.data(), MAX_PARTICLES) != MCL_OK) fail("mcl_set_particles");
    host_particles_stale_ = host_weights_stale_ = false;
}

void ParticleFilterCore::initialize_particles_pose(const Vector3d &pose)
{
    std::fill(weights_.begin(), weights_.end(), 1.0 / MAX_PARTICLES);   // cpp:388
    for (int i = 0; i < MAX_PARTICLES; ++i) {                             // cpp:390-398
        particles_(i, 0) = pose[0] + normal_dist_(rng_) * 0.5;
        particles_(i, 1) = pose[1] + normal_dist_(rng_) * 0.5;
        particles_(i, 2) = pose[2] + normal_dist_(rng_) * 0.4;
        particles_(i, 2) = normalize_angle(particles_(i, 2));
    }
    push_state();
}

void ParticleFilterCore::initialize_global()
{
    if (!map_initialized_) return;                                        // cpp:403
    if (permissible_positions_.empty()) { if (log_error_) log_error_("No free space found in map!"); return; }   // cpp:423-427
    std::uniform_int_distribution<int> pos_dist(0, (int)permissible_positions_.size() - 1);   // cpp:430
    std::uniform_real_distribution<double> angle_dist(0.0, 2.0 * M_PI);
    const double res = (double)map_.resolution;
    for (int i = 0; i < MAX_PARTICLES; ++i) {                             // cpp:433-441
        auto pos = permissible_positions_[pos_dist(rng_)];
        particles_(i, 0) = pos.second * res + map_.origin_x;
        particles_(i, 1) = pos.first * res + map_.origin_y;
        particles_(i, 2) = angle_dist(rng_);
    }
    std::fill(weights_.begin(), weights_.end(), 1.0 / MAX_PARTICLES);     // cpp:443
    push_state();
}

void ParticleFilterCore::MCL(const Vector3d &action, const std::vector<float> &observation)
{
    if (!engine_) return;
    const double a[3] = {action[0], action[1], action[2]};
    int rc;
    if (params_.use_reference_draws) {
        // consume rng_ in the reference's order: N draws of discrete_distribution (cpp:663), then 3N normals (cpp:496-498)
        std::vector<double> u(MAX_PARTICLES), nrm((size_t)MAX_PARTICLES * 3);
        for (auto &x : u) x = std::generate_canonical<double, std::numeric_limits<double>::digits>(rng_);
        for (auto &x : nrm) x = normal_dist_(rng_);
        rc = mcl_update(engine_, a, observation.data(), (int32_t)observation.size(), nrm.data(), u.data());
    } else {
        rc = mcl_update(engine_, a, observation.data(), (int32_t)observation.size(), nullptr, nullptr);
    }
    if (rc != MCL_OK) { fail("mcl_update"); return; }       // particles_/weights_ untouched, like a skipped tick (cpp:756)
    host_particles_stale_ = host_weights_stale_ = true;
    double ms[6];
    if (mcl_get_stage_timings(engine_, ms) == MCL_OK) {      // cpp:667,673,537,606,582,692-693
        timing_stats_.resampling_time += ms[0]; timing_stats_.motion_model_time += ms[1];
        timing_stats_.query_prep_time += ms[2]; timing_stats_.ray_casting_time += ms[3];
        timing_stats_.sensor_model_time += ms[4]; timing_stats_.total_mcl_time += ms[5];
        timing_stats_.measurement_count++;
    }
}

Vector3d ParticleFilterCore::expected_pose()
{
    Vector3d pose;
    if (engine_ && mcl_expected_pose(engine_, pose.v) != MCL_OK) fail("mcl_expected_pose");
    return pose;
}

const MatrixX3d &ParticleFilterCore::particles()
{
    if (engine_ && host_particles_stale_) {
        if (mcl_get_particles(engine_, particles_.data.data(), MAX_PARTICLES) != MCL_OK) fail("mcl_get_particles");
        host_particles_stale_ = false;
    }
    return particles_;
}

const std::vector<double> &ParticleFilterCore::weights()
{
    if (engine_ && host_weights_stale_) {
        if (mcl_get_weights(engine_, weights_.data(), MAX_PARTICLES) != MCL_OK) fail("mcl_get_weights");
        host_weights_stale_ = false;
    }
    return weights_;
}

MatrixX3d ParticleFilterCore::sample_for_visualization(int k)
{
    MatrixX3d out;
    out.resize(k);
    std::vector<double> u(k);
    for (auto &x : u) x = std::generate_canonical<double, std::numeric_limits<double>::digits>(rng_);   // particle_dist(rng_), cpp:954
    if (engine_ && mcl_sample_particles(engine_, k, u.data(), out.data.data()) != MCL_OK) fail("mcl_sample_particles");
    return out;
}

Vector3d ParticleFilterCore::particle_center()
{
    Vector3d c;
    if (engine_ && mcl_particle_mean(engine_, c.v) != MCL_OK) fail("mcl_particle_mean");
    return c;
}

}  // namespace particle_filter_cpp
