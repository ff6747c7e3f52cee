// Which loads see the result of device-scope fp64 atomics issued by an EARLIER kernel on gfx950 (8 XCDs, one L2 each)?
// The engine's pattern: kernel Z zeroes acc[] with plain stores, kernel A adds to random entries with atomicAdd
// (no-return, agent scope), kernel R reads every entry.  R is tried with plain loads, nontemporal loads, agent-scope
// relaxed atomic loads (sc1) and returning atomics; the sequence repeats on the same buffer so that lines cached by an
// earlier R are still around.  Prints the number of wrong entries per variant.
//   hipcc --offload-arch=gfx950 -O3 -o coherence coherence.hip && ./coherence
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_zero(double *acc, int64_t n, int mode)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (mode == 0) acc[i] = 0.0;                                                  // plain store
    else __hip_atomic_store(&acc[i], 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // sc1 store (drops the line)
}
__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__global__ void k_add(double *acc, int64_t n, int64_t m, uint32_t salt)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= m) return;
    const int64_t i = (int64_t)(hash32((uint32_t)t ^ salt) % (uint64_t)n);
    atomicAdd(&acc[i], 1.0);
}
__global__ void k_expect(uint32_t *cnt, int64_t n, int64_t m, uint32_t salt)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= m) return;
    atomicAdd(&cnt[hash32((uint32_t)t ^ salt) % (uint64_t)n], 1u);
}
template <int MODE>
__global__ void k_read(double *acc, const uint32_t *cnt, int64_t n, unsigned long long *bad)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v;
    if (MODE == 0) v = acc[i];
    else if (MODE == 1) v = __builtin_nontemporal_load(&acc[i]);
    else if (MODE == 2) v = __hip_atomic_load(&acc[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else v = atomicAdd(&acc[i], 0.0);
    if (v != (double)cnt[i]) atomicAdd(bad, 1ull);
}
int main()
{
    const int64_t n = 4 << 20, m = 3700000;
    double *acc; uint32_t *cnt; unsigned long long *bad;
    CHK(hipMalloc(&acc, n * 8)); CHK(hipMalloc(&cnt, n * 4)); CHK(hipMalloc(&bad, 8));
    const char *names[4] = {"plain load", "nontemporal load", "agent-scope relaxed atomic load (sc1)", "returning atomicAdd(+0)"};
    for (int zmode = 0; zmode < 2; ++zmode)
        for (int mode = 0; mode < 4; ++mode) {
            unsigned long long total = 0;
            for (int rep = 0; rep < 20; ++rep) {
                const uint32_t salt = 1234u + rep * 77u;
                CHK(hipMemset(cnt, 0, n * 4)); CHK(hipMemset(bad, 0, 8));
                hipLaunchKernelGGL(k_expect, dim3((m + 255) / 256), dim3(256), 0, 0, cnt, n, m, salt);
                hipLaunchKernelGGL(k_zero, dim3((n + 255) / 256), dim3(256), 0, 0, acc, n, zmode);
                hipLaunchKernelGGL(k_add, dim3((m + 255) / 256), dim3(256), 0, 0, acc, n, m, salt);
                switch (mode) {
                case 0: hipLaunchKernelGGL(k_read<0>, dim3((n + 255) / 256), dim3(256), 0, 0, acc, cnt, n, bad); break;
                case 1: hipLaunchKernelGGL(k_read<1>, dim3((n + 255) / 256), dim3(256), 0, 0, acc, cnt, n, bad); break;
                case 2: hipLaunchKernelGGL(k_read<2>, dim3((n + 255) / 256), dim3(256), 0, 0, acc, cnt, n, bad); break;
                default: hipLaunchKernelGGL(k_read<3>, dim3((n + 255) / 256), dim3(256), 0, 0, acc, cnt, n, bad); break;
                }
                unsigned long long b = 0;
                CHK(hipMemcpy(&b, bad, 8, hipMemcpyDeviceToHost));
                total += b;
            }
            printf("zero by %-12s read by %-40s wrong entries over 20 rounds: %llu of %lld\n", zmode ? "sc1 store" : "plain store", names[mode], total, 20ll * n);
        }
    return 0;
}
