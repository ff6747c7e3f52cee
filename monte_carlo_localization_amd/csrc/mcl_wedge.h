// mcl_wedge.h — direction-binned skip fields ("wedge fields"), shared by the host code and the kernels.
//
// skip_k(c) bounds how far the fixed-step march (cast_ray, cpp:611-650) may jump from a sample inside cell c
// when the ray's direction angle lies in wedge k = [2*pi*k/K, 2*pi*(k+1)/K]:
//     skip_k(c) = floor( min over stop cells t REACHABLE from c in wedge k of gap(c, t) ) + 1,
// gap(c,t)^2 = max(|dx|-1,0)^2 + max(|dy|-1,0)^2 as for the isotropic field (mcl_engine.hip).  A sample p in
// the half-open square of c and a later sample p + s*u (u in the wedge) inside the square of t differ by a
// vector of the wedge that lies in the OPEN square (t - c) + (-1,1)^2, so t is reachable only if that open
// square meets the closed wedge.  The wedge is the intersection of two half-planes n1.v >= 0, n2.v >= 0 inside
// one quadrant; an open square o + (-1,1)^2 meets {n.v >= 0} iff n.o + |nx| + |ny| > 0, and it meets the
// quadrant iff sx*ox >= 0 and sy*oy >= 0.  Testing the three conditions separately keeps a superset of the
// reachable cells (corners near the apex), which is the safe direction.  Walls beside a ray no longer shorten
// its jumps; with K = 16 the benchmark input needs 3.9 probes per ray instead of 5.2 with quadrant fields.
#pragma once
#include <cmath>
#include <cstdint>

#ifdef __HIPCC__
#define MCL_HD __host__ __device__
#else
#define MCL_HD
#endif

namespace mcl {

#ifndef MCL_KWEDGES
#define MCL_KWEDGES 16                   // (a build with another count is an experiment: include/mcl_hip_engine.h says 16)
#endif
constexpr int kWedges = MCL_KWEDGES;     // direction bins per turn (a power of two, multiple of 4)
constexpr int kWedgeShift = kWedges == 8 ? 1 : kWedges == 16 ? 2 : kWedges == 32 ? 3 : 4;   // log2(kWedges / 4): bin -> quadrant
static_assert(kWedges == (4 << kWedgeShift), "kWedges must be 8, 16, 32 or 64");
constexpr int kWedgeR = 255;             // offsets searched per axis: gaps beyond 254 are capped anyway

struct WedgeRow { int16_t xa, xb; };     // reachable offsets ox in row oy: xa..xb (none when xa > xb)

// rows[oy + kWedgeR] for oy = -kWedgeR..kWedgeR.  For every row the reachable ox form one interval (the three
// conditions are convex); it is found by testing every ox.
inline void wedge_rows(int k, WedgeRow *rows)
{
    const double two_pi = 6.283185307179586476925286766559;
    const double a0 = two_pi * k / kWedges, a1 = two_pi * (k + 1) / kWedges;
    const double n1x = -std::sin(a0), n1y = std::cos(a0);      // left of the lower edge
    const double n2x = std::sin(a1), n2y = -std::cos(a1);      // right of the upper edge
    const double e1 = std::fabs(n1x) + std::fabs(n1y), e2 = std::fabs(n2x) + std::fabs(n2y);
    const int q = k >> kWedgeShift;
    const int sx = (q == 0 || q == 3) ? 1 : -1, sy = (q == 0 || q == 1) ? 1 : -1;
    for (int oy = -kWedgeR; oy <= kWedgeR; ++oy) {
        int xa = 1, xb = 0;
        bool any = false;
        for (int ox = -kWedgeR; ox <= kWedgeR; ++ox) {
            // 1e-9: the sums are either exactly zero in real arithmetic (square touches the edge: not reachable)
            // or larger than 1e-3 for |o| <= 255, so rounding cannot move a reachable offset below the threshold
            const bool in = sx * ox >= 0 && sy * oy >= 0 && (n1x * ox + n1y * oy + e1 > 1e-9) && (n2x * ox + n2y * oy + e2 > 1e-9);
            if (in) { if (!any) { xa = ox; any = true; } xb = ox; }
        }
        rows[oy + kWedgeR].xa = (int16_t)xa;
        rows[oy + kWedgeR].xb = (int16_t)xb;
    }
}

// nxt[y*Wp + x]: smallest x' >= x with stop(x', y), Wp when there is none; prv: largest x' <= x, -1 when none.
// Cells outside the padded grid are stops (cpp:632-636).
MCL_HD inline int wedge_skip_cell(const int32_t *nxt, const int32_t *prv, int Wp, int Hp, int x, int y, const WedgeRow *rows)
{
    if (nxt[(size_t)y * Wp + x] == x) return 0;                // a stop cell
    int64_t best = (int64_t)1 << 40;
    for (int d = 0; d <= kWedgeR; ++d) {
        const int64_t gy = d > 1 ? d - 1 : 0;
        if (gy * gy >= best) break;
        for (int sgn = (d == 0 ? 1 : -1); sgn <= 1; sgn += 2) {
            const int oy = sgn * d;
            const WedgeRow r = rows[oy + kWedgeR];
            if (r.xa > r.xb) continue;
            const int yy = y + oy, lo = x + r.xa, hi = x + r.xb;
            const bool row_in = yy >= 0 && yy < Hp;
            int ox = 1 << 20;                                   // smallest |ox| of a reachable stop in this row
            if (r.xa <= 0 && r.xb >= 0) {                       // the interval contains ox = 0: look both ways
                const int xr = row_in ? nxt[(size_t)yy * Wp + x] : x;
                const int xl = row_in ? prv[(size_t)yy * Wp + x] : x;
                if (xr <= hi) ox = xr - x;
                if (xl >= lo && x - xl < ox) ox = x - xl;
            } else if (r.xa > 0) {                              // strictly to the right
                const int xr = (!row_in || lo >= Wp) ? lo : nxt[(size_t)yy * Wp + lo];
                if (xr <= hi) ox = xr - x;
            } else {                                            // strictly to the left
                const int xl = (!row_in || hi < 0) ? hi : prv[(size_t)yy * Wp + hi];
                if (xl >= lo) ox = x - xl;
            }
            if (ox < (1 << 20)) {
                const int64_t gx = ox > 1 ? ox - 1 : 0;
                const int64_t g2 = gx * gx + gy * gy;
                if (g2 < best) best = g2;
            }
        }
    }
    if (best >= (int64_t)254 * 254) return 255;
    int64_t rt = (int64_t)sqrt((double)best);
    while (rt * rt > best) --rt;
    while ((rt + 1) * (rt + 1) <= best) ++rt;
    return (int)(rt + 1);
}

// Layout of the sixteen mirrored, ringed wedge fields k_rays_sweep<.., GLOBAL> probes in place (mcl_rays_sweep.h, k_ring_field),
// and the proof obligation that goes with it: the kernel forms the byte offset  row * pitch + column  from the START of the
// allocation, with column = k * stride + cell column.  A walk of field k starts in a cell of the ringed grid (row < Hp + 4,
// column < Wp + 4 <= pitch), runs towards +x, +y only and advances by at most P samples of at most one cell per axis in total
// (a skip larger than the samples left ends the walk before the jump is made; the start's fraction and the guard bias add
// less than one more cell), so the largest offset it can form is
//   k * stride + (Hp + 4 + P) * pitch + (Wp + 4 + P)   =  sweep_global_max_offset(k)
// which must lie inside field k's own stride (rows of stop bytes behind the ringed grid: the tail), and the whole allocation
// below 2^32 (the offset is a 32-bit VGPR); the multiplicands of the kernel's v_mad_u32_u24 (row, pitch) must fit 24 bits.
struct SweepGlobalLayout { bool ok; int pitch; int rows; size_t stride, alloc; };
inline SweepGlobalLayout sweep_global_layout(int Wp, int Hp, int P)
{
    SweepGlobalLayout g{};
    g.pitch = (Wp + 4 + 63) & ~63;
    const int tail = P + 2 + (P + 4 + g.pitch - 1) / g.pitch;        // rows the walk can run past the ringed grid, column overflow included
    g.rows = Hp + 4 + tail;
    g.stride = (size_t)g.rows * (size_t)g.pitch;
    g.alloc = g.stride * (size_t)kWedges;
    g.ok = g.alloc < ((size_t)1 << 32) && g.rows + P < (1 << 24) && g.pitch < (1 << 24) && P >= 1;
    return g;
}
inline size_t sweep_global_max_offset(const SweepGlobalLayout &g, int Wp, int Hp, int P, int k)
{
    return (size_t)k * g.stride + (size_t)(Hp + 4 + P) * (size_t)g.pitch + (size_t)(Wp + 4 + P);
}

}  // namespace mcl
