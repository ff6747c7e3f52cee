// comm_demo.cpp — one RANK of a particle set sharded one process per GPU, in plain C++ against the C ABI: no Python, no torch.
// The engine holds the RCCL communicator (mcl_comm_*); the rendezvous here is a FILE: rank 0 writes the 128-byte id, the other
// ranks wait for it (any channel the processes share would do).  Every rank then runs K x mcl_comm_update, which is
// ParticleFilter::MCL(action, observation) + expected_pose() (cpp:652-716) for the whole set.
// usage: comm_demo <map.bin> <scan.bin> <n_per_rank> <angle_step> <k_updates> <seed> [n_ranks rank id_file]
//   (inputs as mcl_demo's; device = rank; without the last three arguments: one rank)
// prints one JSON object: the poses of the K updates, the exchange of the last update, a digest of this rank's particles.
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "mcl_hip_engine.h"

#define CHECK(call)                                                                                          \
    do {                                                                                                     \
        const int rc_ = (call);                                                                              \
        if (rc_ != MCL_OK) { std::fprintf(stderr, "%s: rc=%d %s\n", #call, rc_, mcl_last_error(h)); return 10; } \
    } while (0)

int main(int argc, char **argv)
{
    if (argc != 7 && argc != 10) { std::fprintf(stderr, "usage: %s map.bin scan.bin n_per_rank angle_step k seed [n_ranks rank id_file]\n", argv[0]); return 2; }
    unsigned W = 0, H = 0;
    float res = 0;
    double ox = 0, oy = 0;
    std::vector<int8_t> grid;
    {
        FILE *f = std::fopen(argv[1], "rb");
        if (!f || std::fscanf(f, "%u %u %f %lf %lf\n", &W, &H, &res, &ox, &oy) != 5) return 3;
        grid.resize((size_t)W * H);
        if (std::fread(grid.data(), 1, grid.size(), f) != grid.size()) return 3;
        std::fclose(f);
    }
    std::vector<float> scan(1081);
    {
        FILE *f = std::fopen(argv[2], "rb");
        if (!f || std::fread(scan.data(), 4, scan.size(), f) != scan.size()) return 4;
        std::fclose(f);
    }
    const int64_t n = std::atoll(argv[3]);
    const int angle_step = std::atoi(argv[4]), k = std::atoi(argv[5]);
    const int n_ranks = argc == 10 ? std::atoi(argv[7]) : 1, rank = argc == 10 ? std::atoi(argv[8]) : 0;
    const char *id_file = argc == 10 ? argv[9] : nullptr;

    mcl_config_t cfg;
    mcl_default_config(&cfg);
    cfg.max_particles = n;
    cfg.device = rank;
    cfg.seed = std::strtoull(argv[6], nullptr, 10);
    mcl_engine_t *h = nullptr;
    if (mcl_create(&cfg, &h) != MCL_OK) { std::fprintf(stderr, "mcl_create: %s\n", mcl_last_error(nullptr)); return 5; }
    CHECK(mcl_set_map(h, grid.data(), W, H, res, ox, oy));
    const float angle_min = (float)(-3.0 * M_PI / 4.0), angle_inc = (float)((3.0 * M_PI / 2.0) / 1080.0);   // lidarCB, cpp:300-320
    std::vector<float> angles, obs;
    for (size_t i = 0; i < scan.size(); i += angle_step) { angles.push_back(angle_min + i * angle_inc); obs.push_back(scan[i]); }
    CHECK(mcl_set_beam_angles(h, angles.data(), (int32_t)angles.size()));
    const double pose0[3] = {0.0, 0.0, 0.0};
    CHECK(mcl_init_particles_pose(h, pose0, n, (int64_t)rank * n, n * n_ranks));      // Philox keyed by the global particle index

    // rendezvous: 128 bytes from rank 0 to every rank
    const char *why = nullptr;
    if (mcl_comm_available(&why) != MCL_OK) { std::fprintf(stderr, "no RCCL: %s\n", why); return 6; }
    unsigned char id[128];
    if (rank == 0) {
        if (mcl_comm_unique_id(id) != MCL_OK) return 6;
        if (id_file) {
            const std::string tmp = std::string(id_file) + ".tmp";
            FILE *f = std::fopen(tmp.c_str(), "wb");
            if (!f || std::fwrite(id, 1, 128, f) != 128) return 6;
            std::fclose(f);
            if (std::rename(tmp.c_str(), id_file) != 0) return 6;
        }
    } else {
        bool got = false;
        for (int t = 0; t < 600 && !got; ++t) {
            if (FILE *f = std::fopen(id_file, "rb")) { got = std::fread(id, 1, 128, f) == 128; std::fclose(f); }
            if (!got) std::this_thread::sleep_for(std::chrono::milliseconds(100));
        }
        if (!got) { std::fprintf(stderr, "rank %d: no id in %s\n", rank, id_file); return 6; }
    }
    CHECK(mcl_comm_create(h, id, n_ranks, rank));

    const double action[3] = {0.05, 0.0, 0.01};
    std::printf("{\"ranks\": %d, \"rank\": %d, \"beams\": %zu, \"poses\": [", n_ranks, rank, angles.size());
    for (int it = 0; it < k; ++it) {
        double pose[3];
        CHECK(mcl_comm_update(h, action, obs.data(), (int32_t)obs.size(), pose));
        std::printf("%s[%.17g, %.17g, %.17g]", it ? ", " : "", pose[0], pose[1], pose[2]);
    }
    int32_t dense = 0, waits = 0;
    uint64_t lb = 0, lp = 0;
    CHECK(mcl_comm_last_exchange(h, &dense, nullptr, nullptr));
    CHECK(mcl_comm_stats(h, &lb, &lp, &waits));
    std::vector<double> xyz((size_t)3 * n);
    CHECK(mcl_get_particles(h, xyz.data(), n));
    if (const char *dump = std::getenv("MCL_DEMO_DUMP")) {                        // this rank's particles, column-major doubles
        FILE *f = std::fopen(dump, "wb");
        if (!f || std::fwrite(xyz.data(), 8, xyz.size(), f) != xyz.size()) return 7;
        std::fclose(f);
    }
    uint64_t digest = 1469598103934665603ull;                                    // FNV-1a over the particle bytes
    const unsigned char *b = reinterpret_cast<const unsigned char *>(xyz.data());
    for (size_t i = 0; i < xyz.size() * 8; ++i) { digest ^= b[i]; digest *= 1099511628211ull; }
    std::printf("], \"last_exchange\": \"%s\", \"list_bytes_received\": %llu, \"host_waits\": %d, \"p0\": [%.17g, %.17g, %.17g], \"digest\": \"%016llx\"}\n",
                dense ? "dense" : "lists", (unsigned long long)lb, waits, xyz[0], xyz[(size_t)n], xyz[(size_t)2 * n], (unsigned long long)digest);
    CHECK(mcl_comm_destroy(h));
    mcl_destroy(h);
    return 0;
}
