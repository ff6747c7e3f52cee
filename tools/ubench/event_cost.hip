// What a timing event costs between two kernels of one stream, and whether hipExtLaunchKernelGGL's start / stop events (bound to
// the dispatch itself) avoid it.  Prints microseconds per kernel of a chain of short kernels: bare, with a hipEventRecord
// between every pair, with start+stop events attached to every launch; and checks that elapsed times across DIFFERENT launches
// (stop of one, stop of a later one; start of one, stop of a later one) agree with plain events.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <vector>

__global__ void spin(unsigned long long *out, int iters)
{
    unsigned long long t = 0;
    for (int i = 0; i < iters; ++i) t += wall_clock64() & 1;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main()
{
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    unsigned long long *d;
    CK(hipMalloc(&d, 8));
    const int N = 200, iters = 2000;
    std::vector<hipEvent_t> ev(2 * N + 2);
    for (auto &evt : ev) CK(hipEventCreate(&evt));
    auto wall = [&](auto body) -> double {
        (void)hipStreamSynchronize(s);
        const auto t0 = std::chrono::steady_clock::now();
        body();
        (void)hipStreamSynchronize(s);
        return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
    };
    for (int rep = 0; rep < 2; ++rep) {
        const double bare = wall([&] { for (int i = 0; i < N; ++i) hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s, d, iters); return 0; });
        const double rec = wall([&] { for (int i = 0; i < N; ++i) { hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s, d, iters); (void)hipEventRecord(ev[i], s); } return 0; });
        const double ext = wall([&] { for (int i = 0; i < N; ++i) hipExtLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s, ev[2 * i], ev[2 * i + 1], 0, d, iters); return 0; });
        float one = 0, span_ss = 0, span_es = 0;
        CK(hipEventElapsedTime(&one, ev[20], ev[21]));               // start / stop of launch 10
        hipError_t e1 = hipEventElapsedTime(&span_es, ev[21], ev[41]);   // stop of launch 10 -> stop of launch 20
        hipError_t e2 = hipEventElapsedTime(&span_ss, ev[20], ev[41]);   // start of launch 10 -> stop of launch 20
        printf("us per kernel: bare %.2f | + hipEventRecord %.2f | hipExtLaunchKernelGGL start+stop %.2f\n", bare, rec, ext);
        printf("ext events: one launch %.2f us; stop10->stop20 %.2f us (%s); start10->stop20 %.2f us (%s); expected ~ %.2f / %.2f\n", one * 1e3,
               span_es * 1e3, hipGetErrorString(e1), span_ss * 1e3, hipGetErrorString(e2), 10 * ext, 10 * ext + one * 1e3);
    }
    // stop event only (start = nullptr) and start event only
    const double stop_only = wall([&] { for (int i = 0; i < N; ++i) hipExtLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s, nullptr, ev[i], 0, d, iters); return 0; });
    float a = 0;
    hipError_t e3 = hipEventElapsedTime(&a, ev[10], ev[20]);
    printf("stop-only launches: %.2f us per kernel; stop10->stop20 %.2f us (%s)\n", stop_only, a * 1e3, hipGetErrorString(e3));
    return 0;
}
