"""Host-side precomputation of the engine (runs without a GPU): the sensor table must equal the
oracle's restatement of cpp:233-292 bit for bit, and the skip-distance field must equal an independent
construction (scipy EDT of the 3x3-dilated stop set) and respect the safety property the exactness
argument of DESIGN.md §4.2 needs."""
import numpy as np
import pytest
from scipy import ndimage


def padded_stops(grid):
    occ = grid > 50
    H, W = occ.shape
    S = np.ones((H + 1, W + 1), bool)
    S[1:, 1:] = occ; S[0, 1:] = occ[0]; S[1:, 0] = occ[:, 0]; S[0, 0] = occ[0, 0]
    return S


def reference_skip(grid):
    S = padded_stops(grid)
    Sp = np.pad(S, 2, constant_values=True)
    Sd = ndimage.binary_dilation(Sp, structure=np.ones((3, 3), bool))
    g2 = np.rint(ndimage.distance_transform_edt(~Sd) ** 2).astype(np.int64)[2:-2, 2:-2]
    r = np.floor(np.sqrt(g2.astype(np.float64))).astype(np.int64)
    r = np.where(r * r > g2, r - 1, r)
    r = np.where((r + 1) * (r + 1) <= g2, r + 1, r)
    skip = np.minimum(r + 1, 255)
    skip[S] = 0
    return skip.astype(np.uint8), S


@pytest.mark.parametrize("P", [207, 239, 50])
def test_sensor_table_bit_exact_vs_oracle(orc, engine_mod, P):
    assert np.array_equal(engine_mod.host_sensor_table(P), orc.sensor_table(P))
    cfg = engine_mod.default_config(z_hit=0.6, z_short=0.1, z_max=0.1, z_rand=0.2, sigma_hit=3.5)
    assert np.array_equal(engine_mod.host_sensor_table(P, cfg), orc.sensor_table(P, 0.6, 0.1, 0.1, 0.2, 3.5))


@pytest.mark.parametrize("mapname", ["sibal1", "spielberg"])
def test_skip_field_matches_independent_construction(request, engine_mod, mapname):
    m = request.getfixturevalue(mapname)
    got = engine_mod.host_skip_field(m.data)
    want, S = reference_skip(m.data)
    assert got.shape == want.shape
    assert np.array_equal(got, want)
    assert ((got == 0) == S).all()


def test_skip_field_safety_property_bruteforce(engine_mod, orc):
    """skip(c) - 1 < gap(c, t) for every stop cell t: no stop cell is reachable in fewer than skip(c)
    unit steps from the interior of c (brute force on a small random map)."""
    rng = np.random.default_rng(0)
    g = np.where(rng.random((40, 50)) < 0.06, 100, 0).astype(np.int8)
    g[rng.random(g.shape) < 0.03] = -1
    skip = engine_mod.host_skip_field(g).astype(np.int64)
    S = padded_stops(g)
    Sb = np.pad(S, 30, constant_values=True)
    ys, xs = np.nonzero(Sb)
    for (y, x), s in np.ndenumerate(skip):
        if s == 0:
            continue
        dx = np.maximum(np.abs(xs - (x + 30)) - 1, 0)
        dy = np.maximum(np.abs(ys - (y + 30)) - 1, 0)
        gap2 = (dx * dx + dy * dy).min()
        assert (s - 1) ** 2 <= gap2 < s * s or s == 255, (y, x, s, gap2)
    # it dominates the Chebyshev field used by the first version of the kernel
    D = ndimage.distance_transform_cdt(~np.pad(S, 1, constant_values=True), metric="chessboard")[1:-1, 1:-1]
    assert (skip >= np.minimum(D, 255)).all()


def brute_directional(grid, q):
    """skip_q(c) = floor(min gap(c,t)) + 1 over stop cells t a quadrant-q ray can still reach
    (sx*(tx-cx) >= 0 and sy*(ty-cy) >= 0), everything outside the padded grid being a stop."""
    sx, sy = [(1, 1), (-1, 1), (-1, -1), (1, -1)][q]
    S = padded_stops(grid)
    Hp, Wp = S.shape
    pad = max(Hp, Wp) + 2
    Sb = np.pad(S, pad, constant_values=True)
    ys, xs = np.nonzero(Sb)
    ys = ys - pad; xs = xs - pad
    out = np.zeros((Hp, Wp), np.int64)
    for cy in range(Hp):
        for cx in range(Wp):
            if S[cy, cx]:
                continue
            m = (sx * (xs - cx) >= 0) & (sy * (ys - cy) >= 0)
            dx = np.maximum(np.abs(xs[m] - cx) - 1, 0)
            dy = np.maximum(np.abs(ys[m] - cy) - 1, 0)
            g2 = int((dx * dx + dy * dy).min())
            r = int(np.floor(np.sqrt(g2)))
            while r * r > g2:
                r -= 1
            while (r + 1) * (r + 1) <= g2:
                r += 1
            out[cy, cx] = min(r + 1, 255)
    return out


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_directional_skip_fields_match_bruteforce(engine_mod, seed):
    rng = np.random.default_rng(seed)
    g = np.where(rng.random((28, 37)) < [0.03, 0.08, 0.2][seed], 100, 0).astype(np.int8)
    g[rng.random(g.shape) < 0.05] = -1
    iso = engine_mod.host_skip_field(g).astype(np.int64)
    for q in range(4):
        got = engine_mod.host_skip_field_dir(g, q).astype(np.int64)
        want = brute_directional(g, q)
        assert np.array_equal(got, want), (q, np.argwhere(got != want)[:5])
        assert (got >= iso).all()                       # fewer constraints -> never a shorter jump
    # the isotropic field is the minimum of the four directional ones
    allq = np.minimum.reduce([engine_mod.host_skip_field_dir(g, q).astype(np.int64) for q in range(4)])
    assert np.array_equal(allq, iso)


def test_directional_field_on_real_map_region(engine_mod, sibal1):
    sub = np.ascontiguousarray(sibal1.data[60:110, 100:160])
    for q in range(4):
        assert np.array_equal(engine_mod.host_skip_field_dir(sub, q).astype(np.int64), brute_directional(sub, q))
