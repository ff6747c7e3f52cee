"""Memory safety of k_rays_sweep's global-field form as arithmetic (no GPU): the kernel forms the byte offset
row * pitch + column from the START of the allocation of the sixteen mirrored, ringed wedge fields (csrc/mcl_wedge.h:
sweep_global_layout; csrc/mcl_rays_sweep.h: MCL_SWG_TRIP), with column = k * stride + cell column.  A walk starts in a cell of
field k's ringed grid, runs towards +x, +y only (the fields are mirrored per quadrant) and advances by at most P samples of at
most one cell per axis in total -- whatever bytes it reads: a skip larger than the samples left ends it before the jump.  So
every offset it can form must lie inside field k's own stride and the allocation below 2^32 (the offset is a 32-bit register).
cpp:195 puts no bound on MAX_RANGE_PX, cpp:632-636 none on the map size."""
import ctypes as C
import itertools

import numpy as np
import pytest

from monte_carlo_localization_amd import engine

KW = 16


def layout(w, h, p):
    lib = engine.load_library()
    out = (C.c_int64 * 6)()
    lib.mcl_host_sweep_global_layout.argtypes = [C.c_uint32, C.c_uint32, C.c_int32, C.POINTER(C.c_int64)]
    assert lib.mcl_host_sweep_global_layout(w, h, p, out) == 0
    return dict(ok=bool(out[0]), pitch=out[1], rows=out[2], stride=out[3], alloc=out[4], max_offset=out[5])


@pytest.mark.parametrize("w,h,p", [(60, 400, 479), (400, 60, 479), (1, 1, 2000), (3, 5000, 999), (2000, 2000, 200), (2000, 2000, 479),
                                   (4000, 4000, 479), (350, 177, 599), (17, 17, 17), (64, 64, 1), (5000, 3, 2037), (8000, 8000, 300)])
def test_every_offset_a_walk_can_form_lies_inside_its_field(w, h, p):
    g = layout(w, h, p)
    wp, hp = w + 1, h + 1
    if not g["ok"]:
        assert g["alloc"] >= 2 ** 32 or g["rows"] + p >= 2 ** 24        # too large for 32-bit offsets: the form is not used
        return
    assert g["pitch"] >= wp + 4 and g["pitch"] % 64 == 0 and g["rows"] >= hp + 4
    assert g["stride"] == g["rows"] * g["pitch"] and g["alloc"] == KW * g["stride"] < 2 ** 32
    assert g["rows"] + p < 2 ** 24 and g["pitch"] < 2 ** 24               # multiplicands of the kernel's v_mad_u32_u24
    # corner cells of the ringed grid (the start of a walk: any cell of rows 0 .. hp + 3, columns 0 .. wp + 3), every field;
    # the walk adds up to p + 1 cells per axis (p samples, the start's fraction and the guard bias)
    for k, r0, c0 in itertools.product(range(KW), (0, hp + 3), (0, wp + 3)):
        lo = k * g["stride"] + r0 * g["pitch"] + c0
        hi = k * g["stride"] + (r0 + p + 1) * g["pitch"] + (c0 + p + 1)
        assert 0 <= lo and hi < (k + 1) * g["stride"] <= g["alloc"]
        assert hi <= g["max_offset"] or k < KW - 1
    assert g["max_offset"] < g["alloc"]


def test_brute_force_on_a_small_map():
    """Every start cell x every advance (dx, dy) in [0, P + 1]^2 on a map small enough to enumerate."""
    w, h, p = 9, 6, 23
    g = layout(w, h, p)
    assert g["ok"]
    r0, c0 = np.meshgrid(np.arange(h + 1 + 4), np.arange(w + 1 + 4), indexing="ij")
    dx = np.arange(p + 2)
    off = (r0[..., None, None] + dx[:, None]) * g["pitch"] + (c0[..., None, None] + dx[None, :])
    assert off.min() >= 0 and off.max() < g["stride"]


# ---- the hybrid form's LDS window (csrc/mcl_rays_sweep.h, the window loader under HYB): ranges beyond the window -------------------
# The rule the loader applies, restated: in a 256 x 256 window whose rays run towards +x, +y, the last row and the last column are
# EXIT cells (0xFE) where they lie inside the grid, and every skip byte (1 .. 127; stops, 0xFF, are left alone) is clamped to
# min(255 - row, 255 - column).  A jump of k samples moves a ray by at most k cells along either axis (a direction component is
# at most one cell per sample), so from any cell a walk that only ever advances by the byte of the cell it stands in stays inside the
# window until it reads a stop, an exit, or runs out of samples -- whatever the field holds.
S, EXIT, STOP = 256, 0xFE, 0xFF


def hybrid_window(field, in_grid):
    """field: S x S bytes in the window's (mirrored) frame, 1 .. 127 or STOP; in_grid: which cells lie inside the map."""
    w = field.copy()
    rr, cc = np.meshgrid(np.arange(S), np.arange(S), indexing="ij")
    lim = np.minimum(S - 1 - rr, S - 1 - cc)
    skip = w != STOP
    w[skip] = np.minimum(w[skip], lim[skip])
    edge = ((rr == S - 1) | (cc == S - 1)) & in_grid
    w[edge] = EXIT
    return w


@pytest.mark.parametrize("seed", range(4))
def test_hybrid_window_no_walk_leaves_it_unnoticed(seed):
    rng = np.random.default_rng(seed)
    field = rng.integers(1, 128, (S, S)).astype(np.uint8)
    field[rng.random((S, S)) < 0.002] = STOP
    in_grid = np.ones((S, S), bool)
    if seed & 1:                                    # the grid ends inside the window: beyond it a window is stop bytes
        in_grid[:, 200:] = False
        in_grid[230:, :] = False
        field[~in_grid] = STOP
    w = hybrid_window(field, in_grid)
    assert (w[in_grid & (w != STOP)] != 0).all()                       # no zero skip anywhere a ray can stand
    assert (w[-1, :][in_grid[-1, :]] == EXIT).all() and (w[:, -1][in_grid[:, -1]] == EXIT).all()
    P = 479
    n = 4000
    y = rng.uniform(2.0, 44.0, n); x = rng.uniform(2.0, 44.0, n)       # origins in the play corner
    th = rng.uniform(0.0, np.pi / 2, n)
    dy, dx = np.sin(th), np.cos(th)                                     # both components in [0, 1]: the mirrored frame
    left = np.full(n, P)
    alive = np.ones(n, bool)
    for _ in range(2 * S):
        r, c = np.floor(y).astype(int), np.floor(x).astype(int)
        assert (r[alive] <= S - 1).all() and (c[alive] <= S - 1).all()  # every read lies inside the window
        b = w[np.minimum(r, S - 1), np.minimum(c, S - 1)].astype(int)
        ended = alive & ((b == STOP) | (b == EXIT) | (b > left))
        alive &= ~ended
        if not alive.any():
            break
        y = np.where(alive, y + b * dy, y); x = np.where(alive, x + b * dx, x)
        left = np.where(alive, left - b, left)
    assert not alive.any()
