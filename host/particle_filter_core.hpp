// particle_filter_core.hpp — host-side mirror of the hot-path surface of the reference's
// particle_filter_cpp::ParticleFilter (include/particle_filter_cpp/particle_filter.hpp:39-48,74-75,
// 102-129), with every numeric step delegated to the MI355X engine through the C ABI
// (include/mcl_hip_engine.h).  Same member/method names, same argument meaning, same "void + log and
// continue" error behaviour as the reference; no ROS, no Eigen — the two Eigen types the path uses are
// replaced by the minimal containers below with the same memory layout (column-major N x 3 doubles).
//
// This file is what the body of the reference class looks like after the patch in INTEGRATION.md; it
// exists so the drop-in can be compiled and tested here, where rclcpp/Eigen are not installed.
#pragma once
#include <cstdint>
#include <functional>
#include <random>
#include <string>
#include <vector>

#include "mcl_hip_engine.h"

namespace particle_filter_cpp {

struct Vector3d {
    double v[3] = {0, 0, 0};
    double &operator[](int i) { return v[i]; }
    double operator[](int i) const { return v[i]; }
};

// Eigen::MatrixXd(N,3) stand-in: column-major, operator()(row, col)
struct MatrixX3d {
    int n = 0;
    std::vector<double> data;
    void resize(int rows) { n = rows; data.assign((size_t)rows * 3, 0.0); }
    double &operator()(int r, int c) { return data[(size_t)c * n + r]; }
    double operator()(int r, int c) const { return data[(size_t)c * n + r]; }
    int rows() const { return n; }
};

// the fields of nav_msgs/OccupancyGrid the path reads (cpp:190-195, 199-213, 628-642)
struct OccupancyGrid {
    uint32_t width = 0, height = 0;
    float resolution = 0.f;               // MapMetaData.resolution is float32
    double origin_x = 0, origin_y = 0;
    std::vector<int8_t> data;
};

// utils::performance::TimingStats (utils.hpp:49-61)
struct TimingStats {
    double total_mcl_time = 0, ray_casting_time = 0, sensor_model_time = 0, motion_model_time = 0, resampling_time = 0,
           query_prep_time = 0;
    int measurement_count = 0;
    void reset() { *this = TimingStats(); }
};

struct Params {   // declare_parameter defaults, cpp:23-47
    int max_particles = 2000;
    double squash_factor = 2.2, max_range = 12.0;
    double z_short = 0.01, z_max = 0.07, z_rand = 0.12, z_hit = 0.80, sigma_hit = 8.0;
    double motion_dispersion_x = 0.05, motion_dispersion_y = 0.025, motion_dispersion_theta = 0.25;
    int device = 0;
    uint64_t seed = 0;                    // the reference seeds rng_ from std::random_device (cpp:20)
    bool use_reference_draws = false;     // draw uniforms/normals from rng_ exactly like cpp:663, 496-498 and inject them
};

class ParticleFilterCore {
  public:
    explicit ParticleFilterCore(const Params &p, std::function<void(const std::string &)> log_error = nullptr);
    ~ParticleFilterCore();
    ParticleFilterCore(const ParticleFilterCore &) = delete;
    ParticleFilterCore &operator=(const ParticleFilterCore &) = delete;

    bool ok() const { return engine_ != nullptr; }

    // get_omap() after the GetMap future resolves (cpp:190-224): keeps the grid, MAX_RANGE_PX, the free-cell
    // mask for initialize_global, and lets the engine build the sensor table.
    void set_map(const OccupancyGrid &map);
    // lidarCB's first message (cpp:297-313)
    void set_downsampled_angles(const std::vector<float> &angles);

    void initialize_particles_pose(const Vector3d &pose);   // cpp:382-399
    void initialize_global();                               // cpp:401-446

    void MCL(const Vector3d &action, const std::vector<float> &observation);   // cpp:652-694
    Vector3d expected_pose();                                                   // cpp:696-716

    // lazily synced copies of the reference's members (hpp:102-103); readers: visualize cpp:946-958,
    // get_current_pose cpp:903-908
    const MatrixX3d &particles();
    const std::vector<double> &weights();
    MatrixX3d sample_for_visualization(int max_viz_particles);   // the weighted draw of cpp:949-956
    Vector3d particle_center();                                   // particles_.colwise().mean(), cpp:904

    int MAX_PARTICLES, MAX_RANGE_PX = 0;
    bool map_initialized_ = false;
    TimingStats timing_stats_;
    std::mt19937 rng_;
    std::normal_distribution<double> normal_dist_{0.0, 1.0};
    mcl_engine_t *engine() { return engine_; }

  private:
    void fail(const char *what);
    void push_state();
    Params params_;
    std::function<void(const std::string &)> log_error_;
    mcl_engine_t *engine_ = nullptr;
    OccupancyGrid map_;
    std::vector<std::pair<int, int>> permissible_positions_;
    std::vector<float> downsampled_angles_;
    MatrixX3d particles_;
    std::vector<double> weights_;
    bool host_particles_stale_ = true, host_weights_stale_ = true;
};

}  // namespace particle_filter_cpp
