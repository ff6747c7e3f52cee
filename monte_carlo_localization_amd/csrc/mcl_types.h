// mcl_types.h -- the few plain types and constants that both the kernels (mcl_kernels.h) and the host-only translation units
// (mcl_comm.hip, mcl_group.hip, through mcl_engine_internal.h) need.  No device code.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace mcl {

constexpr int kScanThreads = 256;
constexpr int kScanItems = 8;
constexpr int kScanTile = kScanThreads * kScanItems;
constexpr int kMaxShards = 16;

// scratch the ray stage needs zeroed, cleared here instead of by one memset each (null = not used by this launch)
struct PrepClear {
    double *logw_acc;                  // n
    uint32_t *far_flags;               // n
    unsigned long long *fix_count;     // fix_words 64-bit words
    int fix_words;
    unsigned long long *fix_over;      // 2 words (overflow flag, work counter)
    unsigned long long *exact_count;   // 1 word
    unsigned long long *far_count;     // 1 word
    int *bbox;                         // 4: +big, +big, -big, -big; [4], [5]: k_tile_compact; [6] = bbox_play
    int bbox_play;                     // cells a ray window leaves for the particles of a work item (0: no windowed kernel): sort_layout
    uint32_t *hist;                    // hist_n bucket counters
    uint32_t hist_n;
};

// mcl_group_update: the maximum over the shards' maxima, read where they live (peer pointers)
struct GroupMaxArgs { const double *src[kMaxShards]; int n; double *out; };

}  // namespace mcl
