// mcl_rays_legacy.h -- k_rays_quad (K3c) and k_rays_cell (K3d): the windowed ray kernels of rounds 1-2, the predecessors of
// k_rays_sweep (mcl_rays_sweep.h).  AUTO never picks them; they are compiled only with -DMCL_LEGACY_RAY_KERNELS
// (libmcl_hip_engine_legacy.so, which the tests load for ray_kernel = MCL_RAYS_QUAD / MCL_RAYS_CELL: two more implementations of
// cpp:586-650 that the sweep kernel is compared with).  Included from mcl_kernels.h, inside its includes' context.
#pragma once

namespace mcl {

// ---- K3c: quadrant windows, one byte per cell (MCL_RAYS_QUAD) -------------------------------------
//
// Same algorithm and the same three precision levels as k_rays_skip, with the work split by ray
// direction: workgroup (slice, q) handles, for the particles of its slice, the beams whose direction lies
// in quadrant q.  Those rays only ever move away from the particle in x and in y, so the window needs
// MAX_RANGE_PX cells on ONE side of the cloud per axis: a (P + extent)^2 window with a whole byte per
// cell fits half the LDS of a CU.  Two effects: the probe block loses the nibble decode (10 VALU
// instead of 13) and two 1024-thread workgroups are resident per CU (8 waves per SIMD instead of 4), which
// hides the LDS round trip of the dependent probe chain.  To run at 64 VGPRs without spills the hot
// kernel contains level 1 only:
//   * a ray with a sample too close to a cell boundary is appended to a device work list and resolved by
//     k_rays_fix (levels 2 and 3 on the global field);
//   * a (particle, quadrant) pair that does not fit the window is flagged and handled by k_rays_far
//     (the global-field path of k_rays_skip restricted to that quadrant's beams).
// All three kernels add their partial log-weights with fp64 atomics, which is exact and order-independent
// here (every term is a multiple of 2^-24 and the sums stay below 2^13).
constexpr int kQFx = 22;                 // same fixed point as k_rays_skip (|U| <= 2^22 fits v_mad_i32_i24)
constexpr uint32_t kQG1 = 132u;          // (1 + s)/2 <= 128 units for s <= 255

template <bool COUNT>
__global__ __launch_bounds__(kRayThreads, 8) void k_rays_quad(RayArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    __shared__ int item_sh;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned long long dbg_t0 = 0, cnt_probe = 0;
    if (a.dbg) dbg_t0 = __builtin_amdgcn_s_memrealtime();
    const int64_t per = (a.n + a.nslices - 1) / a.nslices;
    const int nitems = 4 * a.nslices;
    // the probe block addresses the window with raw LDS offsets (ds_read_u8 ... offset:kQLdsBase)
    // the window is addressed from the raw LDS offset kQLdsBase: checked on the host at mcl_create (choose_ray_mode keeps the
    // engine off this kernel when the layout differs); a mismatch reports a full fix-up list (the host re-runs the stage with
    // k_rays_skip) instead of aborting the device
    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)lds_raw != (uint32_t)kQLdsBase) {
        if (threadIdx.x == 0) a.fix_count[(size_t)blockIdx.x * 8] = a.fix_cap + 1ull;
        return;
    }
    // Persistent workgroups (2 per CU) pull (slice, quadrant) items from a device-side queue: quadrants carry
    // very different numbers of beams (a 270-degree scan puts ~360 beams in two quadrants and ~180 in the
    // other two), and a static grid of one workgroup per item kept only half of the slots busy (measured).
    for (;;) {
    __syncthreads();                                    // every wave is done with the previous window
    if (threadIdx.x == 0) item_sh = (int)atomicAdd(a.work_counter, 1ull);
    __syncthreads();
    const int item = item_sh;
    if (item >= nitems) break;
    // heavy quadrants are not known in advance; rotate so that consecutive items differ in quadrant
    const int slice = item >> 2;
    const int q = ((item & 3) + (item >> 3)) & 3;
    const int64_t p_begin = (int64_t)slice * per;
    const int64_t p_end = (p_begin + per < a.n) ? p_begin + per : a.n;
    if (p_begin >= p_end) continue;
    const int sxp = (q == 0 || q == 3), syp = (q == 0 || q == 1);   // rays move toward +x / +y ?

    const int S = a.qside;
    const int mlo = 3;                                  // cells kept behind the particle (truncation column + guard)
    int wx0, wy0;
    {
        double *red = reinterpret_cast<double *>(lds_raw);
        double sx = 0.0, sy = 0.0;
        for (int64_t i = p_begin + threadIdx.x; i < p_end; i += kRayThreads) {
            double4 c = a.pc[i];
            double gx = c.z, gy = c.w;
            if (gx == gx && gy == gy && fabs(gx) < 1e9 && fabs(gy) < 1e9) { sx += gx; sy += gy; }
        }
        sx = wave_sum(sx); sy = wave_sum(sy);
        if (lane == 0) { red[2 * wave] = sx; red[2 * wave + 1] = sy; }
        __syncthreads();
        double mx = 0.0, my = 0.0;
        for (int k = 0; k < kRayWaves; ++k) { mx += red[2 * k]; my += red[2 * k + 1]; }
        int64_t cntp = p_end - p_begin;
        mx /= (double)cntp; my /= (double)cntp;
        // the cloud may extend E/2 on either side of its mean; the rays add P+2 cells on the forward side
        const int E = S - (a.P + 2) - mlo;
        const int back = E / 2 + mlo;
        int cxm = (int)floor(mx) + 1, cym = (int)floor(my) + 1;          // padded coordinates
        wx0 = sxp ? cxm - back : cxm + back - S;
        wy0 = syp ? cym - back : cym + back - S;
        __syncthreads();
        uint64_t *win = reinterpret_cast<uint64_t *>(lds_raw);
        const uint8_t *fieldq = a.distq[q];              // only stops a quadrant-q ray can reach bound its jumps
        const int wpr = S >> 3;                          // 8 cells per 64-bit word
        const int nwords = wpr * S;
        for (int wi = threadIdx.x; wi < nwords; wi += kRayThreads) {
            int row = wi / wpr, cw = wi - row * wpr;
            int gy = wy0 + row, gx = wx0 + cw * 8;
            uint64_t b8 = 0;
            if (gy >= 0 && gy < a.Hp) {
                const uint8_t *rowp = fieldq + (size_t)gy * a.Wps;
                if (gx >= 0 && gx + 8 <= a.Wps) {
                    b8 = *reinterpret_cast<const uint64_t *>(rowp + gx);       // any byte alignment
                } else {
                    for (int k = 0; k < 8; ++k)
                        if (gx + k >= 0 && gx + k < a.Wps) b8 |= (uint64_t)rowp[gx + k] << (8 * k);
                }
            }
            // LDS encoding: stop (0) -> 255, skips capped at 254.  A stop then fails the single loop test
            // "skip <= samples left" (samples left <= 254), which saves the compare with zero in the probe block.
            uint64_t enc = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                uint32_t v = (uint32_t)(b8 >> (8 * k)) & 0xFFu;
                v = v == 0u ? 255u : (v == 255u ? 254u : v);
                enc |= (uint64_t)v << (8 * k);
            }
            win[wi] = enc;
        }
        __syncthreads();
    }
    uint32_t stride_v = (uint32_t)S, gbias_v = kQG1 << (32 - kQFx);
    asm volatile("" : "+v"(stride_v), "+v"(gbias_v));
    const unsigned char *ldsb = lds_raw;
    const uint32_t gthresh = a.force_exact ? 0xFFFFFFFFu : ((2u * kQG1) << (32 - kQFx));

    for (int64_t i = p_begin + wave; i < p_end; i += kRayWaves) {
        int ja, jb, ja2;
        quad_ranges(a.qr[i], q, a.B, ja, jb, ja2);
        if (ja >= jb && ja2 >= a.B) continue;
        const double4 pci = a.pc[i];
        const double wpx = pci.z - (double)(wx0 - 1);     // window-relative padded coordinate
        const double wpy = pci.w - (double)(wy0 - 1);
        const double fwd = (double)(a.P + 2), bwd = 2.0;
        const bool inx = sxp ? (wpx - bwd >= 0.0 && wpx + fwd < (double)S) : (wpx - fwd >= 0.0 && wpx + bwd < (double)S);
        const bool iny = syp ? (wpy - bwd >= 0.0 && wpy + fwd < (double)S) : (wpy - fwd >= 0.0 && wpy + bwd < (double)S);
        if (!(inx && iny)) {                               // not in this window (or NaN): k_rays_far does this pair
            if (lane == 0) atomicOr(reinterpret_cast<unsigned int *>(a.far_flags) + i, 1u << (8 * q));
            continue;
        }
        // the particle's own cell gives a first skip shared by all its beams; if the particle itself sits
        // within 2^-30 px of a cell boundary all of its rays go to the fix-up list
        const double p0x = wpx + kMagic, p0y = wpy + kMagic;
        const uint32_t lox = (uint32_t)__double2loint(p0x), loy = (uint32_t)__double2loint(p0y);
        const int cx0 = (__double2hiint(p0x) & 0xFFFFF) - kCellBase, cy0 = (__double2hiint(p0y) & 0xFFFFF) - kCellBase;
        const int d0 = ldsb[cy0 * S + cx0];               // 255 = the particle sits in a stop cell
        const int s0 = (d0 == 255 || d0 < 1) ? 1 : d0;
        const uint32_t g0 = ((lox < loy ? lox : loy) < kGuard) ? 0u : 0xFFFFFFFFu;
        const uint32_t P0x = (uint32_t)rint_i32(wpx * 4194304.0 - 2147483648.0) + 0x80000000u;   // rint(wpx*2^22) mod 2^32
        const uint32_t P0y = (uint32_t)rint_i32(wpy * 4194304.0 - 2147483648.0) + 0x80000000u;
        const int rem_start = s0 <= a.P ? a.P - s0 : 0;
        const double ncth = -pci.x * 4194304.0, sths = pci.y * 4194304.0;
        double acc = 0.0;
        for (int seg = 0; seg < 2; ++seg)
        for (int j0 = seg ? ja2 : ja, je = seg ? a.B : jb; j0 < je; j0 += 64) {
            const int j = j0 + lane;                       // beam_cs is padded by 256 entries
            const bool valid = j < je;
            const double2 cs = a.beam_cs[j];
            // a padding lane (beam outside this range) gets a zero direction: it probes the particle's own cell,
            // which is inside the window and never reads as 0, so it leaves the loop at once
            const int NUx = valid ? rint_i32(__builtin_fma(ncth, cs.x, sths * cs.y)) : 0;
            const int NUy = valid ? rint_i32(__builtin_fma(ncth, cs.y, -(sths * cs.x))) : 0;
            const uint32_t Pex = mad_i24(-a.P, NUx, P0x), Pey = mad_i24(-a.P, NUy, P0y);
            int rem = valid ? rem_start : 0;
            uint32_t g = g0, byte = 0;
            if (!COUNT) {
                // The whole probe loop in one block: 9 VALU + 1 LDS read + 4 SALU per trip (the scalar unit is
                // shared by the CU's four SIMDs and the compiler's loop bookkeeping made it the bottleneck).
                // A lane leaves when the skip exceeds the samples left (always on a stop, encoded 255): the
                // borrow of v_sub_co clears its exec bit; its rem/byte keep the values of its last probe.
                // The scalar countdown only exists so that no LDS content whatsoever can make a wave spin.
                uint32_t Tx, Ty, t0, t1, addr;
                unsigned long long saved_exec;
                uint32_t countdown;
                asm volatile(
                    "s_mov_b64 %[sv], exec\n\t"
                    "s_movk_i32 %[cd], 300\n"
                    "1:\n\t"
                    "v_mad_i32_i24 %[tx], %[rem], %[nux], %[pex]\n\t"
                    "v_mad_i32_i24 %[ty], %[rem], %[nuy], %[pey]\n\t"
                    "v_lshrrev_b32 %[t0], 22, %[tx]\n\t"                    // cx
                    "v_lshrrev_b32 %[t1], 22, %[ty]\n\t"                    // cy
                    "v_mad_u32_u24 %[ad], %[t1], %[str], %[t0]\n\t"         // byte offset = cy * S + cx
                    "ds_read_u8 %[by], %[ad] offset:%[lb]\n\t"              // + LDS offset of the window
                    "v_lshl_add_u32 %[t0], %[tx], 10, %[gb]\n\t"            // biased fractions in the top 22 bits
                    "v_lshl_add_u32 %[t1], %[ty], 10, %[gb]\n\t"
                    "v_min3_u32 %[g], %[g], %[t0], %[t1]\n\t"
                    "s_waitcnt lgkmcnt(0)\n\t"
                    "v_sub_co_u32 %[rem], vcc, %[rem], %[by]\n\t"           // samples left -= skip; borrow = done
                    "s_andn2_b64 exec, exec, vcc\n\t"
                    "s_cbranch_execz 2f\n\t"
                    "s_sub_u32 %[cd], %[cd], 1\n\t"
                    "s_cbranch_scc0 1b\n"
                    "2:\n\t"
                    "s_mov_b64 exec, %[sv]"
                    : [tx] "=&v"(Tx), [ty] "=&v"(Ty), [t0] "=&v"(t0), [t1] "=&v"(t1), [ad] "=&v"(addr), [by] "+v"(byte), [g] "+v"(g),
                      [rem] "+v"(rem), [sv] "=&s"(saved_exec), [cd] "=&s"(countdown)
                    : [nux] "v"(NUx), [nuy] "v"(NUy), [pex] "v"(Pex), [pey] "v"(Pey), [str] "v"(stride_v), [gb] "v"(gbias_v),
                      [lb] "n"(kQLdsBase)
                    : "memory", "vcc", "scc");
                if (rem >= 0) g = 0u;                        // countdown expired (impossible with a well-formed window): fix-up list
            } else {
                bool go;
                int trips = 0;
                do {
                    const uint32_t Tx = mad_i24(rem, NUx, Pex), Ty = mad_i24(rem, NUy, Pey);
                    const uint32_t gx = (Tx << (32 - kQFx)) + gbias_v, gy = (Ty << (32 - kQFx)) + gbias_v;
                    const uint32_t gm = gx < gy ? gx : gy;
                    g = g < gm ? g : gm;
                    byte = ldsb[(Ty >> kQFx) * (uint32_t)S + (Tx >> kQFx)];
                    uint32_t nr;
                    const bool over = __builtin_usub_overflow((uint32_t)rem, byte, &nr);
                    go = !over;
                    rem = (int)nr;
                    cnt_probe += go ? 1 : 0;
                } while (go && ++trips <= 300);
                if (go) g = 0u;
            }
            if (COUNT && valid) ++cnt_probe;
            const bool amb = valid && g < gthresh;
            if (valid && !amb) {
                const int r = (byte == 255u) ? a.P - (rem + 255) - 1 : a.P;
                acc += (double)a.Lt[__mul24(r, a.bpad) + j];      // 32-bit index: (P+1) * bpad < 2^24
                if (a.steps) a.steps[(size_t)i * a.B + j] = (uint8_t)r;
            }
            if (amb) {
                // append to THIS workgroup's segment of the fix-up list (segments are private to a workgroup;
                // counters and entries are only ever touched by device-scope atomics, which execute at the
                // memory side and are therefore coherent across the XCDs' L2s)
                const unsigned long long slot = atomicAdd(&a.fix_count[(size_t)blockIdx.x * 8], 1ull);
                if (slot < a.fix_cap)
                    atomicExch(&a.fix_list[(size_t)blockIdx.x * a.fix_cap + slot], ((unsigned long long)i << 16) | (unsigned long long)j);
            }
        }
        acc = wave_sum(acc);
        if (lane == 0) atomicAdd(&a.logw[i], acc);
    }
    }   // work items
    if (COUNT && a.counters) {
        cnt_probe = wave_sum_u64(cnt_probe);
        if (lane == 0 && cnt_probe) atomicAdd(&a.counters[2], cnt_probe);
    }
    if (a.dbg) {
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long *d = a.dbg + (size_t)blockIdx.x * 4;
            d[0] = dbg_t0; d[1] = __builtin_amdgcn_s_memrealtime();
            d[2] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));    // HW_REG_HW_ID
            d[3] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));   // HW_REG_XCC_ID
        }
    }
}


// ---- K3d: one particle per lane on cell-sorted particles (MCL_RAYS_CELL) --------------------------------------
//
// Same windows, fields, fixed point, guard, fix-up list and far path as k_rays_quad; what changes is who does
// what.  Work item (slice, q): the slice is a run of the CELL-SORTED particle order, each lane owns one particle
// and walks that particle's beams of quadrant q one after the other (slot t = 0, 1, ...).  Per-particle work
// (window-relative origin, first skip, quadrant ranges, heading rotation constants) is done once per item
// instead of once per 64 rays, no cross-lane reduction is needed (a lane adds its particle's partial log-weight
// with one atomic), and the lanes of a wave run near-identical rays (same slot of the same quadrant range,
// origins within a cell or two, headings within a degree), so the probe loop runs with nearly all lanes active.
template <bool COUNT>
__global__ __launch_bounds__(kRayThreads, 8) void k_rays_cell(RayArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    __shared__ int item_sh;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned long long cnt_probe = 0;
    const int64_t per = (a.n + a.nslices - 1) / a.nslices;
    const int nitems = kWedges * a.nslices;
    // the window is addressed from the raw LDS offset kQLdsBase: checked on the host at mcl_create (choose_ray_mode keeps the
    // engine off this kernel when the layout differs); a mismatch reports a full fix-up list (the host re-runs the stage with
    // k_rays_skip) instead of aborting the device
    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)lds_raw != (uint32_t)kQLdsBase) {
        if (threadIdx.x == 0) a.fix_count[(size_t)blockIdx.x * 8] = a.fix_cap + 1ull;
        return;
    }
    for (;;) {
    __syncthreads();
    if (threadIdx.x == 0) item_sh = (int)atomicAdd(a.work_counter, 1ull);
    __syncthreads();
    const int item = item_sh;
    if (item >= nitems) break;
    const int slice = item / kWedges;
    const int kbin = ((item % kWedges) + 5 * slice) % kWedges;     // consecutive items differ in direction
    const int q = kbin >> kWedgeShift;
    const int64_t p_begin = (int64_t)slice * per;
    const int64_t p_end = (p_begin + per < a.n) ? p_begin + per : a.n;
    if (p_begin >= p_end) continue;
    const int sxp = (q == 0 || q == 3), syp = (q == 0 || q == 1);

    const int S = a.qside;
    const int mlo = 3;
    int wx0, wy0;
    {
        const double2 mm = a.slice_mean[slice];                          // wave-uniform: scalar loads
        const double mx = mm.x, my = mm.y;
        const int E = S - (a.P + 2) - mlo;
        const int back = E / 2 + mlo;
        int cxm = (int)floor(mx) + 1, cym = (int)floor(my) + 1;
        wx0 = sxp ? cxm - back : cxm + back - S;
        wy0 = syp ? cym - back : cym + back - S;
        uint64_t *win = reinterpret_cast<uint64_t *>(lds_raw);
        const uint8_t *fieldq = a.distw + (size_t)kbin * a.distw_stride;   // only stops a wedge-kbin ray can reach bound its jumps
        const int wpr = S >> 3;
        const int nwords = wpr * S;
        // thread t copies words t, t + 1024, ...: (row, word-in-row) advance by constants, no division in the loop
        const int drow = kRayThreads / wpr, dcw = kRayThreads - drow * wpr;
        int row = (int)threadIdx.x / wpr, cw = (int)threadIdx.x - row * wpr;
        for (int wi = threadIdx.x; wi < nwords; wi += kRayThreads, row += drow, cw += dcw) {
            if (cw >= wpr) { cw -= wpr; ++row; }
            int gy = wy0 + row, gx = wx0 + cw * 8;
            // the wedge fields are stored in the LDS encoding (stop = 0xFF, skips 1..127); outside the grid is stop
            uint64_t b8 = ~0ull;
            if (gy >= 0 && gy < a.Hp) {
                const uint8_t *rowp = fieldq + (size_t)gy * a.Wps;
                if (gx >= 0 && gx + 8 <= a.Wp) {
                    b8 = *reinterpret_cast<const uint64_t *>(rowp + gx);
                } else {
                    for (int k = 0; k < 8; ++k)
                        if (gx + k >= 0 && gx + k < a.Wp) b8 = (b8 & ~(0xFFull << (8 * k))) | ((uint64_t)rowp[gx + k] << (8 * k));
                }
            }
            win[wi] = b8;
        }
        __syncthreads();
    }
    // level-1 error bound: 0.5 unit for the origin + 0.5 unit per sample for the direction, s <= P samples;
    // a sample within that (+4) of a cell boundary sends its ray to the fix-up list
    const uint32_t guard_units = (uint32_t)(a.P + 1) / 2u + 5u;
    uint32_t stride_v = (uint32_t)S, gbias_v = guard_units << (32 - kQFx);
    asm volatile("" : "+v"(stride_v), "+v"(gbias_v));
    const unsigned char *ldsb = lds_raw;
    const uint32_t gthresh = a.force_exact ? 0xFFFFFFFFu : ((2u * guard_units) << (32 - kQFx));
    const int negP = __builtin_amdgcn_readfirstlane(-a.P);

    for (int64_t s0g = p_begin + (int64_t)wave * 64; s0g < p_end; s0g += (int64_t)kRayWaves * 64) {
        const int64_t slot = s0g + lane;
        const bool have = slot < p_end;
        const int64_t sl = have ? slot : p_end - 1;
        const double4 pci = a.pcs[sl];
        const uint32_t i = a.perm[sl];
        // beams of this particle in wedge kbin: [ja, jb) and, for scans wider than a turn minus one wedge, [ja2, B)
        int ja = 0, jb = 0, ja2 = a.B;
        {
            const double th = a.ths[sl];
            if (th == th && fabs(th) < 1e6) {
                const int w0 = beam_wedge(th, a.beam_angle[0]), wl = beam_wedge(th, a.beam_angle[a.B - 1]);
                const int m = w0 + ((kbin - w0) & (kWedges - 1));
                if (m <= wl) {
                    ja = m == w0 ? 0 : first_beam_in_wedge(th, a.beam_angle, a.B, m, a.beam_a0, a.beam_inv_inc);
                    jb = m == wl ? a.B : first_beam_in_wedge(th, a.beam_angle, a.B, m + 1, a.beam_a0, a.beam_inv_inc);
                    if (m + kWedges <= wl) ja2 = first_beam_in_wedge(th, a.beam_angle, a.B, m + kWedges, a.beam_a0, a.beam_inv_inc);
                }
            } else if (kbin == 0) {
                jb = a.B;            // garbage heading: one range in quadrant 0 like k_particle_prep (position is NaN -> far path)
            }
        }
        int n1 = jb > ja ? jb - ja : 0, n2 = a.B > ja2 ? a.B - ja2 : 0;
        if (!have) { n1 = 0; n2 = 0; }
        const double wpx = pci.z - (double)(wx0 - 1);
        const double wpy = pci.w - (double)(wy0 - 1);
        const double fwd = (double)(a.P + 2), bwd = 2.0;
        const bool inx = sxp ? (wpx - bwd >= 0.0 && wpx + fwd < (double)S) : (wpx - fwd >= 0.0 && wpx + bwd < (double)S);
        const bool iny = syp ? (wpy - bwd >= 0.0 && wpy + fwd < (double)S) : (wpy - fwd >= 0.0 && wpy + bwd < (double)S);
        const bool inwin = inx && iny;
        if (!inwin && n1 + n2 > 0) {                           // not in this window (or NaN): k_rays_far does this pair
            atomicOr(reinterpret_cast<unsigned int *>(a.far_flags) + i, 1u << (8 * q));
            n1 = 0; n2 = 0;
        }
        const int total = n1 + n2;
        // a lane without rays gets a zero direction and zero samples from the window's cell (2, 2): every probe it
        // makes reads that cell, whose byte is never 0, and leaves the loop at once
        const bool live = total > 0;
        const double lpx = live ? wpx : 2.5, lpy = live ? wpy : 2.5;
        const double p0x = lpx + kMagic, p0y = lpy + kMagic;
        const uint32_t lox = (uint32_t)__double2loint(p0x), loy = (uint32_t)__double2loint(p0y);
        const int cx0 = (__double2hiint(p0x) & 0xFFFFF) - kCellBase, cy0 = (__double2hiint(p0y) & 0xFFFFF) - kCellBase;
        const int d0 = ldsb[cy0 * S + cx0];
        const int s0 = (d0 > 127 || d0 < 1) ? 1 : d0;               // own cell is a stop: first sample one step away
        const uint32_t g0 = ((lox < loy ? lox : loy) < kGuard) ? 0u : 0xFFFFFFFFu;
        const uint32_t P0x = (uint32_t)rint_i32(lpx * 4194304.0 - 2147483648.0) + 0x80000000u;
        const uint32_t P0y = (uint32_t)rint_i32(lpy * 4194304.0 - 2147483648.0) + 0x80000000u;
        const int rem_start = (live && s0 <= a.P) ? a.P - s0 : 0;
        const double ncth = live ? -pci.x * 4194304.0 : 0.0, sths = live ? pci.y * 4194304.0 : 0.0;
        int tmax = total;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) tmax = max(tmax, __shfl_xor(tmax, o, 64));
        tmax = __builtin_amdgcn_readfirstlane(tmax);
        // slot t of this lane is beam ja + t for t < n1, then ja2 + (t - n1); past its last slot a lane repeats
        // its last beam (result discarded) so that it stays on a valid in-window ray
        const int jlast = total > 0 ? (n2 > 0 ? a.B - 1 : jb - 1) : 0;
        int j = n1 > 0 ? ja : (n2 > 0 ? ja2 : 0);
        // a second range only exists for scans wider than three quadrants; the common case steps j by one
        const bool wraps = __builtin_amdgcn_readfirstlane((int)(__ballot(n2 > 0 && n1 > 0) != 0ull)) != 0;
        const uint32_t bpad4 = (uint32_t)__builtin_amdgcn_readfirstlane(a.bpad * 4);
        double acc = 0.0;
        auto walk = [&](auto wrap_tag) {
        constexpr bool WRAP = decltype(wrap_tag)::value;
        // software pipeline: the direction of the next beam is requested before this beam's probe loop and the table
        // entry of this beam is added after the next one's, so neither load is waited for where it is issued
        double2 cs = *reinterpret_cast<const double2 *>(reinterpret_cast<const char *>(a.beam_cs) + ((uint32_t)j << 4));
        float lt_pending = 0.f;                    // table entry of the previous beam (0 when it had none), not yet in acc
        for (int t = 0; t < tmax; ++t) {
            // consume the previous beam's entry first: its register is then free for this beam's load, and the wait
            // for it sits a whole beam after its issue
            acc += (double)lt_pending;
            asm volatile("" : "+v"(acc));
            const bool valid = t < total;
            const int jcur = j;
            {
                int jn = j + 1;
                if (WRAP && jn == jb && n1 > 0 && t < n1) jn = ja2;    // end of the first range: continue with the second
                j = min(jn, jlast);
            }
            const int NUx = rint_i32(__builtin_fma(ncth, cs.x, sths * cs.y));
            const int NUy = rint_i32(__builtin_fma(ncth, cs.y, -(sths * cs.x)));
            // cs is dead now: the next beam's direction lands in the same registers while this beam is traced
            cs = *reinterpret_cast<const double2 *>(reinterpret_cast<const char *>(a.beam_cs) + ((uint32_t)j << 4));
            const uint32_t Pex = mad_i24_s(negP, NUx, P0x), Pey = mad_i24_s(negP, NUy, P0y);
            int rem;
            uint32_t g;
            bool expired = false;
            if (!COUNT) {
                // Probe loop as in k_rays_quad with two changes.  The cell byte is read SIGNED: a stop (0xFF = -1) makes
                // the unsigned subtraction borrow whatever rem holds and leaves rem + 1 > 0, a skip larger than the
                // samples left leaves rem < 0, so "samples left at the hit" is max(rem, 0) and the reversed table is
                // indexed with it directly.  The first trip is peeled so that the per-particle start values are read
                // in place (no copies per ray).
                uint32_t Tx, Ty, t0, t1, addr, byte;
                unsigned long long saved_exec;
                uint32_t countdown;
                asm volatile(
                    "s_mov_b64 %[sv], exec\n\t"
                    "s_movk_i32 %[cd], 300\n\t"
                    "v_mad_i32_i24 %[tx], %[rem0], %[nux], %[pex]\n\t"
                    "v_mad_i32_i24 %[ty], %[rem0], %[nuy], %[pey]\n\t"
                    "v_lshrrev_b32 %[t0], 22, %[tx]\n\t"
                    "v_lshrrev_b32 %[t1], 22, %[ty]\n\t"
                    "v_mad_u32_u24 %[ad], %[t1], %[str], %[t0]\n\t"
                    "ds_read_i8 %[by], %[ad] offset:%[lb]\n\t"
                    "v_lshl_add_u32 %[t0], %[tx], 10, %[gb]\n\t"
                    "v_lshl_add_u32 %[t1], %[ty], 10, %[gb]\n\t"
                    "v_min3_u32 %[g], %[g0], %[t0], %[t1]\n\t"
                    "s_waitcnt lgkmcnt(0)\n\t"
                    "v_sub_co_u32 %[rem], vcc, %[rem0], %[by]\n\t"
                    "s_andn2_b64 exec, exec, vcc\n\t"
                    "s_cbranch_execz 2f\n"
                    "1:\n\t"
                    "v_mad_i32_i24 %[tx], %[rem], %[nux], %[pex]\n\t"
                    "v_mad_i32_i24 %[ty], %[rem], %[nuy], %[pey]\n\t"
                    "v_lshrrev_b32 %[t0], 22, %[tx]\n\t"
                    "v_lshrrev_b32 %[t1], 22, %[ty]\n\t"
                    "v_mad_u32_u24 %[ad], %[t1], %[str], %[t0]\n\t"
                    "ds_read_i8 %[by], %[ad] offset:%[lb]\n\t"
                    "v_lshl_add_u32 %[t0], %[tx], 10, %[gb]\n\t"
                    "v_lshl_add_u32 %[t1], %[ty], 10, %[gb]\n\t"
                    "v_min3_u32 %[g], %[g], %[t0], %[t1]\n\t"
                    "s_waitcnt lgkmcnt(0)\n\t"
                    "v_sub_co_u32 %[rem], vcc, %[rem], %[by]\n\t"
                    "s_andn2_b64 exec, exec, vcc\n\t"
                    "s_cbranch_execz 2f\n\t"
                    "s_sub_u32 %[cd], %[cd], 1\n\t"
                    "s_cbranch_scc0 1b\n"
                    "2:\n\t"
                    "s_mov_b64 exec, %[sv]"
                    : [tx] "=&v"(Tx), [ty] "=&v"(Ty), [t0] "=&v"(t0), [t1] "=&v"(t1), [ad] "=&v"(addr), [by] "=&v"(byte), [g] "=&v"(g),
                      [rem] "=&v"(rem), [sv] "=&s"(saved_exec), [cd] "=&s"(countdown)
                    : [rem0] "v"(rem_start), [g0] "v"(g0), [nux] "v"(NUx), [nuy] "v"(NUy), [pex] "v"(Pex), [pey] "v"(Pey), [str] "v"(stride_v),
                      [gb] "v"(gbias_v), [lb] "n"(kQLdsBase)
                    : "memory", "vcc", "scc");
                // the countdown only expires if the window is malformed (impossible): every ray of the pass to the fix-up list
                expired = __builtin_amdgcn_readfirstlane((int)countdown) < 0;
            } else {
                bool go;
                int trips = 0;
                rem = rem_start;
                g = g0;
                do {
                    const uint32_t Tx = mad_i24(rem, NUx, Pex), Ty = mad_i24(rem, NUy, Pey);
                    const uint32_t gx = (Tx << (32 - kQFx)) + gbias_v, gy = (Ty << (32 - kQFx)) + gbias_v;
                    const uint32_t gm = gx < gy ? gx : gy;
                    g = g < gm ? g : gm;
                    const uint32_t byte = (uint32_t)(int)(int8_t)ldsb[(Ty >> kQFx) * (uint32_t)S + (Tx >> kQFx)];
                    uint32_t nr;
                    const bool over = __builtin_usub_overflow((uint32_t)rem, byte, &nr);
                    go = !over;
                    rem = (int)nr;
                    cnt_probe += (go && valid) ? 1 : 0;
                } while (go && ++trips <= 300);
                if (go) g = 0u;
            }
            if (COUNT && valid) ++cnt_probe;
            const uint32_t thr = expired ? 0xFFFFFFFFu : gthresh;     // wave-uniform: a scalar select
            const bool amb = valid && g < thr;
            lt_pending = 0.f;
            if (valid && !amb) {
                const int left = rem > 0 ? rem : 0;                 // samples left at the hit; 0 = no hit (step index P)
                lt_pending = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(a.Ltr) + mad_u24_s((uint32_t)left, bpad4, (uint32_t)jcur << 2));
                if (a.steps) a.steps[(size_t)i * a.B + jcur] = (uint8_t)(a.P - left);
            }
            if (amb) {
                const unsigned long long fslot = atomicAdd(&a.fix_count[(size_t)blockIdx.x * 8], 1ull);
                if (fslot < a.fix_cap)
                    atomicExch(&a.fix_list[(size_t)blockIdx.x * a.fix_cap + fslot], ((unsigned long long)i << 16) | (unsigned long long)jcur);
            }
        }
        acc += (double)lt_pending;
        };
        if (wraps) walk(std::true_type{}); else walk(std::false_type{});
        if (live) atomicAdd(&a.logw[i], acc);
    }
    }   // work items
    if (COUNT && a.counters) {
        cnt_probe = wave_sum_u64(cnt_probe);
        if (lane == 0 && cnt_probe) atomicAdd(&a.counters[2], cnt_probe);
    }
}


}  // namespace mcl
