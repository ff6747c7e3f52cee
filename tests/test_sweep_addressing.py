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


# ---- the hybrid form's LDS window (csrc/mcl_rays_sweep.h: the budget of a lane's walk under HYB): ranges beyond the window ----------
# The rule, restated: in the 256 x 256 window every ray of a wedge runs towards +x, +y, and a unit step of a ray of mirrored wedge
# wm (directions wm * 22.5 .. (wm + 1) * 22.5 degrees) moves it by at most cos(wm * 22.5) cells in x and sin((wm + 1) * 22.5) in y.
# A lane whose origin is (x0, y0) gets budget = floor(min((255 - x0) / dx_max, (255 - y0) / dy_max)) samples (at most the range):
# within them every sample of every ray of the wedge lies in a cell of the window, whatever the field holds; a ray that is
# unstopped when they are spent goes on in the global fields.
S = 256
INV_DX = [1.0, 1.0823922002923940, 1.4142135623730951, 2.6131259297527530]
INV_DY = [2.6131259297527530, 1.4142135623730951, 1.0823922002923940, 1.0]


@pytest.mark.parametrize("wm", range(4))
def test_hybrid_budget_keeps_every_sample_inside_the_window(wm):
    rng = np.random.default_rng(wm)
    assert abs(INV_DX[wm] - 1.0 / np.cos(np.radians(22.5 * wm))) < 1e-15 * INV_DX[wm] * 4
    assert abs(INV_DY[wm] - 1.0 / np.sin(np.radians(22.5 * (wm + 1)))) < 1e-15 * INV_DY[wm] * 4
    n, P = 20000, 479
    x0 = rng.uniform(2.0, 45.0, n); y0 = rng.uniform(2.0, 45.0, n)                  # origins in the play corner (40 cells + margins)
    # directions of the wedge, and up to one beam (a quarter of a degree) beyond either end: the virtual beams of a scan-edge lane
    psi = np.radians(rng.uniform(22.5 * wm - 0.25, 22.5 * (wm + 1) + 0.25, n))
    dx, dy = np.abs(np.cos(psi)), np.abs(np.sin(psi))
    budget = np.minimum(np.floor(np.minimum((S - 1 - x0) * INV_DX[wm], (S - 1 - y0) * INV_DY[wm])), P)
    assert (budget >= 209).all()                                                     # the window is laid out for 211 px of reach
    xe, ye = x0 + budget * dx, y0 + budget * dy                                      # the farthest sample a walk can stand on
    assert (xe < S).all() and (ye < S).all() and (np.floor(xe) <= S - 1).all() and (np.floor(ye) <= S - 1).all()
    inside = (psi >= np.radians(22.5 * wm)) & (psi <= np.radians(22.5 * (wm + 1)))
    assert (xe[inside] <= S - 1 + 1e-9).all() and (ye[inside] <= S - 1 + 1e-9).all()   # real beams keep a whole cell of margin
