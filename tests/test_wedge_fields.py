"""Wedge (direction-binned) skip fields of MCL_RAYS_CELL (csrc/mcl_wedge.h), checked on the CPU:
* against an independent brute-force construction that decides reachability geometrically
  (open offset square vs. closed wedge cone), and
* by the property that matters: a skipping march driven by the wedge fields returns the same step index as
  the literal fixed-step march of the oracle (cast_ray, cpp:611-650) for random rays."""
import math

import numpy as np
import pytest

from test_host_precompute import padded_stops


def _seg_intersects_ray(p, q, d):
    """closed segment p-q vs. ray {r*d, r >= 0} (2-D), tolerant to touching (touching counts as hit)."""
    px, py = p; qx, qy = q
    ex, ey = qx - px, qy - py
    den = d[0] * ey - d[1] * ex
    if abs(den) < 1e-15:
        return False
    # r*d = p + s*e
    r = (px * ey - py * ex) / den
    s = (px * d[1] - py * d[0]) / den
    return r >= -1e-12 and -1e-12 <= s <= 1 + 1e-12


def open_square_meets_wedge(ox, oy, a0, a1, eps=1e-7):
    """Does the OPEN square (ox,oy) + (-1,1)^2 contain a point of the closed cone of directions [a0, a1]?
    Decided on the closed square shrunk by eps (independent of the half-plane formulation in mcl_wedge.h)."""
    lo_x, hi_x, lo_y, hi_y = ox - 1 + eps, ox + 1 - eps, oy - 1 + eps, oy + 1 - eps
    if lo_x <= 0 <= hi_x and lo_y <= 0 <= hi_y:
        return True                                  # apex inside
    corners = [(lo_x, lo_y), (hi_x, lo_y), (hi_x, hi_y), (lo_x, hi_y)]
    for cx, cy in corners:                           # a corner inside the cone
        ang = math.atan2(cy, cx) % (2 * math.pi)
        lo = a0 % (2 * math.pi)
        if (ang - lo) % (2 * math.pi) <= (a1 - a0):
            return True
    for a in (a0, a1):                               # a cone edge crossing a square side
        d = (math.cos(a), math.sin(a))
        for i in range(4):
            if _seg_intersects_ray(corners[i], corners[(i + 1) % 4], d):
                return True
    return False


def brute_wedge(grid, k, K):
    S = padded_stops(grid)
    Hp, Wp = S.shape
    pad = max(Hp, Wp) + 2
    Sb = np.pad(S, pad, constant_values=True)
    ys, xs = np.nonzero(Sb)
    ys = ys - pad; xs = xs - pad
    a0, a1 = 2 * math.pi * k / K, 2 * math.pi * (k + 1) / K
    reach = {}
    out = np.zeros((Hp, Wp), np.int64)
    for cy in range(Hp):
        for cx in range(Wp):
            if S[cy, cx]:
                continue
            best = None
            for tx, ty in zip(xs, ys):
                o = (int(tx - cx), int(ty - cy))
                g2 = max(abs(o[0]) - 1, 0) ** 2 + max(abs(o[1]) - 1, 0) ** 2
                if best is not None and g2 >= best:
                    continue
                if o not in reach:
                    reach[o] = open_square_meets_wedge(o[0], o[1], a0, a1)
                if reach[o]:
                    best = g2
            r = int(math.isqrt(best))
            out[cy, cx] = min(r + 1, 255)
    return out


@pytest.mark.parametrize("seed", [0, 1])
def test_wedge_fields_vs_geometric_bruteforce(engine_mod, seed):
    rng = np.random.default_rng(seed)
    g = np.where(rng.random((14, 17)) < [0.05, 0.15][seed], 100, 0).astype(np.int8)
    K = engine_mod.WEDGES
    iso = engine_mod.host_skip_field(g).astype(np.int64)
    n_eq = n_all = 0
    for k in range(K):
        got = engine_mod.host_skip_field_wedge(g, k).astype(np.int64)
        want = brute_wedge(g, k, K)
        assert ((got == 0) == (want == 0)).all()
        # never longer than the exact wedge bound (safety), never shorter than the isotropic or quadrant bound
        assert (got <= want).all(), (k, np.argwhere(got > want)[:5])
        assert (got >= iso).all()
        assert (got >= engine_mod.host_skip_field_dir(g, k // (K // 4)).astype(np.int64)).all()
        n_eq += int((got == want).sum()); n_all += got.size
    # the half-plane formulation only loses corner cases near the apex
    assert n_eq >= 0.97 * n_all, (n_eq, n_all)


def test_wedge_march_equals_literal_march(engine_mod, orc, sibal1):
    """Skipping along the reference's sample lattice with the wedge field of the ray's direction bin returns the
    oracle's step index (fp64 positions; rays whose samples come within 1e-9 px of a cell boundary are skipped,
    the kernels resolve those through their guard levels)."""
    sub = np.ascontiguousarray(sibal1.data[40:140, 80:200])
    H, W = sub.shape
    res, ox, oy = sibal1.resolution, 0.0, 0.0
    om = orc.OracleMap(sub, res, ox, oy)
    P = om.max_range_px
    K = engine_mod.WEDGES
    fields = [engine_mod.host_skip_field_wedge(sub, k).astype(np.int64) for k in range(K)]
    rng = np.random.default_rng(5)
    free = np.argwhere(sub == 0)
    n = 4000
    pick = free[rng.integers(0, len(free), n)]
    x = (pick[:, 1] + rng.random(n)) * om.resolution
    y = (pick[:, 0] + rng.random(n)) * om.resolution
    ang = rng.uniform(-math.pi, math.pi, n)
    ang[:64] = np.round(ang[:64] / (math.pi / 8)) * (math.pi / 8)          # exactly on wedge edges / axes
    ranges, steps = orc.cast_many(om, x, y, ang)
    checked = 0
    for i in range(n):
        k = int(math.floor(ang[i] * K / (2 * math.pi))) % K
        F = fields[k]
        px, py = x[i] / om.resolution, y[i] / om.resolution
        ux, uy = math.cos(ang[i]), math.sin(ang[i])
        s = 0
        ambiguous = False
        r = P
        c0 = F[int(math.floor(py)) + 1, int(math.floor(px)) + 1]
        s = max(int(c0), 1)
        while s <= P:
            sx_, sy_ = px + s * ux, py + s * uy
            fx, fy = sx_ - math.floor(sx_), sy_ - math.floor(sy_)
            if min(fx, 1 - fx, fy, 1 - fy) < 1e-9:
                ambiguous = True
                break
            cx, cy = int(math.floor(sx_)) + 1, int(math.floor(sy_)) + 1       # padded coordinates
            if cx < 0 or cy < 0 or cx > W or cy > H:
                r = s - 1
                break
            d = F[cy, cx]
            if d == 0:
                r = s - 1
                break
            s += int(d)
        if ambiguous:
            continue
        checked += 1
        assert r == steps[i], (i, k, r, steps[i], x[i], y[i], ang[i])
    assert checked > 0.95 * n
