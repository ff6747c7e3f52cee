// demo_main.cpp — drives the host mirror the way timer_update() drives the reference class
// (cpp:761-778): init cloud at a pose, then K x { MCL(action, observation); expected_pose() }.
// usage: mcl_demo <map.bin> <scan.bin> <n_particles> <angle_step> <k_updates> <seed> <reference_draws 0|1>
//   map.bin : "W H resolution_f32 ox oy\n" + W*H int8;  scan.bin: 1081 float32 ranges
// prints one JSON object (poses per update, first particle, weight stats, timings).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "particle_filter_core.hpp"

using namespace particle_filter_cpp;

int main(int argc, char **argv)
{
    if (argc < 8) { std::fprintf(stderr, "usage: %s map.bin scan.bin n angle_step k seed refdraws\n", argv[0]); return 2; }
    OccupancyGrid g;
    {
        FILE *f = std::fopen(argv[1], "rb");
        if (!f) return 3;
        if (std::fscanf(f, "%u %u %f %lf %lf\n", &g.width, &g.height, &g.resolution, &g.origin_x, &g.origin_y) != 5) return 3;
        g.data.resize((size_t)g.width * g.height);
        if (std::fread(g.data.data(), 1, g.data.size(), f) != g.data.size()) return 3;
        std::fclose(f);
    }
    std::vector<float> scan(1081);
    {
        FILE *f = std::fopen(argv[2], "rb");
        if (!f || std::fread(scan.data(), 4, scan.size(), f) != scan.size()) return 4;
        std::fclose(f);
    }
    Params p;
    p.max_particles = std::atoi(argv[3]);
    const int angle_step = std::atoi(argv[4]);
    const int k = std::atoi(argv[5]);
    p.seed = std::strtoull(argv[6], nullptr, 10);
    p.use_reference_draws = std::atoi(argv[7]) != 0;
    ParticleFilterCore pf(p, [](const std::string &m) { std::fprintf(stderr, "[ERROR] %s\n", m.c_str()); });
    if (!pf.ok()) return 5;
    pf.set_map(g);
    // lidarCB (cpp:300-320): float angle arithmetic, then downsample
    const float angle_min = (float)(-3.0 * M_PI / 4.0), angle_inc = (float)((3.0 * M_PI / 2.0) / 1080.0);
    std::vector<float> angles, obs;
    for (size_t i = 0; i < scan.size(); i += angle_step) { angles.push_back(angle_min + i * angle_inc); obs.push_back(scan[i]); }
    pf.set_downsampled_angles(angles);
    pf.rng_.seed((uint32_t)p.seed);
    pf.normal_dist_.reset();
    Vector3d pose0;
    pf.initialize_particles_pose(pose0);
    std::printf("{\"max_range_px\": %d, \"beams\": %zu, \"init_p0\": [%.17g, %.17g, %.17g], \"poses\": [", pf.MAX_RANGE_PX, angles.size(),
                pf.particles()(0, 0), pf.particles()(0, 1), pf.particles()(0, 2));
    Vector3d action;
    action[0] = 0.05; action[2] = 0.01;
    for (int it = 0; it < k; ++it) {
        pf.MCL(action, obs);
        Vector3d e = pf.expected_pose();
        std::printf("%s[%.17g, %.17g, %.17g]", it ? ", " : "", e[0], e[1], e[2]);
    }
    const auto &w = pf.weights();
    double sw = 0, wmax = 0;
    for (double x : w) { sw += x; if (x > wmax) wmax = x; }
    const auto &pp = pf.particles();
    auto viz = pf.sample_for_visualization(60);
    Vector3d c = pf.particle_center();
    std::printf("], \"p0\": [%.17g, %.17g, %.17g], \"w0\": %.17g, \"sum_w\": %.17g, \"wmax\": %.17g, \"viz_rows\": %d, "
                "\"center\": [%.17g, %.17g, %.17g], \"mean_total_mcl_ms\": %.6f, \"updates\": %d}\n",
                pp(0, 0), pp(0, 1), pp(0, 2), w[0], sw, wmax, viz.rows(), c[0], c[1], c[2],
                pf.timing_stats_.total_mcl_time / (pf.timing_stats_.measurement_count ? pf.timing_stats_.measurement_count : 1),
                pf.timing_stats_.measurement_count);
    return 0;
}
