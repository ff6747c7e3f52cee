"""The level-1 arithmetic of k_rays_sweep's walk restated in numpy (no GPU): the claims csrc/mcl_rays_sweep.h makes about its
fixed-point directions, checked on the numbers instead of in a comment.

* A direction component is carried as an integer X of 2^-32 px per sample: X = round(round(x + 1) - nu e y) with x, y the scaled
  (2^32 - 3), mirrored GRID direction of the beam -- stepped with the three-term recurrence d(j+2) = 2 cos(inc) d(j+1) - d(j) --
  and e the beam's own offset from the grid (MCL_SW_INTS_REC, MCL_SW_STEP_REC).  Claim: 0 <= X <= 2^32 - 1 and
  -3 <= X - d 2^32 <= +2 for the TRUE direction d = |cos / sin (theta + (double)angle_f32[j])| (cpp:533, 616-617), over a full
  wedge of 90+ steps -- which is what sweep_guard_units(P) = 3 (P + 1) + (P + 1) / 64 + 8 budgets for (plus half a unit for the
  origin and 0.2 for the reference's own accumulated rounding).
* On an evenly spaced scan the first beam of a direction wedge is ceil((m W - theta - a0) / inc) unless that lies within 4e-6 rad
  of a whole beam (first_beam_in_wedge<GRID>): the guess equals the exact classification floor((theta + angle) K / 2 pi) >= m.
The GPU tests compare the kernel's results with the oracle bit for bit; this file pins WHY they can be equal."""
import numpy as np
import pytest

SCALE = 4294967296.0 - 3.0
KW = 16


def lidar_angles(B=1081, a_min=-2.3561945, inc=0.004363323):
    """float32 angles as a driver publishes them: a_min + j * inc evaluated in float32"""
    return (np.float32(a_min) + np.arange(B, dtype=np.float32) * np.float32(inc)).astype(np.float32)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_turned_directions_stay_within_three_units_of_the_true_direction(seed):
    rng = np.random.default_rng(seed)
    ang = lidar_angles()
    B = ang.size
    a0 = float(ang[0])
    inc = (float(ang[-1]) - a0) / (B - 1)
    grid = a0 + np.arange(B) * inc
    e = ang.astype(np.float64) - grid
    assert np.abs(e).max() <= 2e-6                      # the REC condition of mcl_set_beam_angles
    K = 2.0 * np.cos(inc)
    worst_lo, worst_hi = 0.0, 0.0
    for _ in range(400):
        th = rng.uniform(-np.pi, np.pi)
        j0 = int(rng.integers(0, B - 130))
        n = 128                                         # more than a wedge's 90 beams
        # mirrored frame of the wedge the first beam lies in: component signs of its direction
        sx = 1.0 if np.cos(th + grid[j0 + n // 2]) >= 0 else -1.0
        sy = 1.0 if np.sin(th + grid[j0 + n // 2]) >= 0 else -1.0
        nu = sx * sy
        aq, bq = np.cos(th) * SCALE * sx, np.sin(th) * SCALE * sx
        def start(j):
            c, s = np.cos(grid[j]), np.sin(grid[j])
            x = aq * c - bq * s
            y = (bq * c + aq * s) * (1.0 if sx == sy else -1.0)
            return x, y
        xa, ya = start(j0)
        xb, yb = start(j0 + 1)
        for t in range(n):
            j = j0 + t
            x, y = (xa, ya) if t % 2 == 0 else (xb, yb)
            Xx = np.rint(np.rint(x + 1.0) - nu * e[j] * y)
            Xy = np.rint(np.rint(y + 1.0) + nu * e[j] * x)
            dx = sx * np.cos(th + float(ang[j])) * 4294967296.0
            dy = sy * np.sin(th + float(ang[j])) * 4294967296.0
            if dx >= 0 and dy >= 0:                     # beams of THIS quadrant (the kernel only walks those; the others are discarded)
                for X, d in ((Xx, dx), (Xy, dy)):
                    assert 0.0 <= X <= 4294967295.0
                    worst_lo, worst_hi = min(worst_lo, X - d), max(worst_hi, X - d)
            if t % 2 == 0:
                xa, ya = K * xb - xa, K * yb - ya
            else:
                xb, yb = K * xa - xb, K * ya - yb
    assert -3.01 <= worst_lo and worst_hi <= 2.01, (worst_lo, worst_hi)


@pytest.mark.parametrize("P", [57, 200, 207, 239, 243, 479, 999, 2037])
def test_guard_budget_covers_the_error_of_every_sample(P):
    guard = 3 * (P + 1) + ((P + 1) >> 6) + 8            # sweep_guard_units
    per_sample, origin, reference = 3.01, 0.5 + 2 ** -13, 0.2
    assert guard >= per_sample * P + origin + reference
    assert 2 * guard < 2 ** 16                          # the guard stays a sliver of the 2^32-unit cell: 2^-17 px at most


@pytest.mark.parametrize("seed", [5, 6])
def test_grid_guess_of_a_wedges_first_beam_is_the_exact_classification(seed):
    rng = np.random.default_rng(seed)
    ang = lidar_angles()
    B = ang.size
    a0 = float(ang[0])
    inv_inc = (B - 1) / (float(ang[-1]) - a0)
    W = 2.0 * np.pi / KW
    angd = ang.astype(np.float64)
    skipped = 0
    for _ in range(3000):
        th = rng.uniform(-np.pi, np.pi)
        wedge = np.floor((th + angd) * (KW * 0.15915494309189533577)).astype(int)       # beam_wedge of every beam
        m = int(rng.integers(wedge[0] + 1, wedge[-1] + 1))
        exact = int(np.searchsorted(wedge, m, side="left"))                              # first beam with wedge >= m
        x = (m * W - th - a0) * inv_inc
        if abs(x - np.rint(x)) > 4e-6 * inv_inc + 1e-6:
            assert min(max(int(np.ceil(x)), 0), B) == exact
        else:
            skipped += 1                                # the kernel takes the exact path there
    assert skipped < 30
