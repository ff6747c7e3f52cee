#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../monte_carlo_localization_amd/csrc/mcl_kernels.h"
int main()
{
    int nb = 0;
    for (size_t lds : {0ul, 40000ul, 65536ul, 70000ul, 78400ul, 81920ul}) {
        hipFuncSetAttribute(reinterpret_cast<const void *>(&mcl::k_rays_quad<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 81920);
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void *>(&mcl::k_rays_quad<false>), 1024, lds);
        printf("k_rays_quad dyn LDS %zu -> %d blocks/CU (%s)\n", lds, nb, hipGetErrorString(e));
    }
    hipFuncAttributes at;
    hipFuncGetAttributes(&at, reinterpret_cast<const void *>(&mcl::k_rays_quad<false>));
    printf("numRegs %d sharedSizeBytes %zu localSizeBytes %zu maxThreads %d\n", at.numRegs, at.sharedSizeBytes, at.localSizeBytes, at.maxThreadsPerBlock);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("sharedMemPerBlock %zu maxSharedMemoryPerMultiProcessor %zu regsPerBlock %d maxThreadsPerMP %d\n", p.sharedMemPerBlock, p.maxSharedMemoryPerMultiProcessor, p.regsPerBlock, p.maxThreadsPerMultiProcessor);
    return 0;
}
