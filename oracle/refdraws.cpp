// refdraws.cpp — ORACLE helper (test infrastructure only, see mcl_oracle.c header).
//
// Reproduces the random-draw stream the reference consumes.  The reference owns ONE
// std::mt19937 rng_ and ONE std::normal_distribution<double> normal_dist_ (hpp:165-167) and
// draws from them sequentially:
//   * normal_dist_(rng_)                       — cpp:392-394 (init cloud), cpp:496-498 (motion)
//   * std::discrete_distribution<int>(...)(rng_) — cpp:663; libstdc++'s operator() draws
//     one std::generate_canonical<double,53>(rng_) per call (bits/random.tcc).
// These are libstdc++ <random> semantics (GCC's polar-method normal_distribution caches its
// second value inside the distribution object), so the honest way to restate them is to call
// the very same library, which is what this file does.  It contains no reference code.
#include <cstdint>
#include <limits>
#include <random>

namespace {
struct Stream {
    std::mt19937 rng;
    std::normal_distribution<double> normal{0.0, 1.0};
    explicit Stream(uint32_t seed) : rng(seed) { normal.reset(); }
};
}  // namespace

extern "C" {

void *rd_create(uint32_t seed) { return new Stream(seed); }
void rd_destroy(void *h) { delete static_cast<Stream *>(h); }

// n sequential normal_dist_(rng_) draws
void rd_normals(void *h, int64_t n, double *out)
{
    Stream *s = static_cast<Stream *>(h);
    for (int64_t i = 0; i < n; ++i) out[i] = s->normal(s->rng);
}

// n sequential draws as discrete_distribution::operator() makes them
void rd_uniforms(void *h, int64_t n, double *out)
{
    Stream *s = static_cast<Stream *>(h);
    for (int64_t i = 0; i < n; ++i)
        out[i] = std::generate_canonical<double, std::numeric_limits<double>::digits>(s->rng);
}

// raw 32-bit outputs (for the uniform_int/uniform_real draws of initialize_global, cpp:430-441)
void rd_raw(void *h, int64_t n, uint32_t *out)
{
    Stream *s = static_cast<Stream *>(h);
    for (int64_t i = 0; i < n; ++i) out[i] = static_cast<uint32_t>(s->rng());
}

}  // extern "C"
