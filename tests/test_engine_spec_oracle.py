"""CPU checks of the engine-spec half of the oracle (orc_eng_*): the pieces the HIP kernels are compared
with bit-for-bit where the reference's own arithmetic is sequential or degenerate."""
import math
import os

import numpy as np
import pytest

from conftest import GOLDEN


def test_philox_known_answers(orc):
    # Random123 kat_vectors, philox4x32 10 rounds
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        assert tuple(int(v) for v in orc.eng_philox4x32(ctr, key)) == want


def test_det_exp_accuracy_and_edges(orc):
    xs = np.concatenate([-np.logspace(-12, 2.8, 4000), [0.0, -1e-300, -744.9, -745.1, -800.0, -1e9]])
    got = orc.eng_det_exp(xs)
    want = np.exp(xs)
    ok = want > 1e-300
    rel = np.abs(got[ok] - want[ok]) / want[ok]
    assert rel.max() < 4.5e-16                     # <= 2 ulp
    assert got[xs == 0.0][0] == 1.0
    assert (got[xs < -745.0] == 0.0).all()
    assert math.isnan(orc.eng_det_exp([float("nan")])[0])


def test_log_sums_are_exact_in_any_order(orc):
    """E4: fp32 table entries have |value| in [0.5, 8) -> multiples of 2^-24; 1081-term sums need < 53 bits."""
    L = orc.eng_log_table(orc.sensor_table(207))
    assert np.isfinite(L).all() and (np.abs(L) >= 0.5).all() and (np.abs(L) < 8.0).all()
    rng = np.random.default_rng(0)
    v = L.ravel()[rng.integers(0, L.size, 1081)].astype(np.float64)
    from fractions import Fraction
    exact = sum(Fraction(float(t)) for t in v)
    for perm in range(5):
        s = 0.0
        for t in rng.permutation(v):
            s += t
        assert Fraction(s) == exact


def test_fixed_point_resampling_matches_reference_discrete_distribution(orc):
    """With the same uniforms the exact integer CDF picks the same parents as libstdc++'s
    discrete_distribution on the G3 fixtures (they can differ only when a uniform falls within
    ~2^-36 of a CDF step)."""
    for B in (61, 121):
        z = np.load(os.path.join(GOLDEN, f"g3_mcl_step_B{B}.npz"))
        w, u = z["weights_in"], z["uniforms"]
        ref = orc.resample_indices(w, u)
        assert np.array_equal(ref, z["idx"])
        q = orc.eng_quantize_weights(w)
        k53 = np.floor(u * 2.0 ** 53).astype(np.uint64)
        eng = orc.eng_resample_indices(q, 0, k53=k53)
        assert np.array_equal(eng, ref)


def test_resampling_degenerate_inputs(orc):
    n = 100
    u = np.linspace(0, 0.999, n)
    with np.errstate(all="ignore"):
        assert (orc.resample_indices(np.zeros(n), u) == 0).all()      # G6: NaN CDF -> index 0
    assert (orc.eng_resample_indices(np.zeros(n, np.uint64), 0, k53=np.zeros(n, np.uint64)) == 0).all()
    # one-hot weights
    w = np.zeros(n); w[37] = 1.0
    ref = orc.resample_indices(w, u)
    # u == 0.0 exactly (probability 2^-53): lower_bound(cp, 0.0) returns slot 0 even though w[0] == 0;
    # the engine's strict comparison skips zero-weight slots — the one documented difference
    assert ref[0] == 0 and (ref[1:] == 37).all()
    q = orc.eng_quantize_weights(w)
    assert (orc.eng_resample_indices(q, 0, k53=np.floor(u * 2.0 ** 53).astype(np.uint64)) == 37).all()
    assert (orc.eng_resample_indices(q, 1, k0=12345) == 37).all()


def test_systematic_resampling_counts(orc):
    rng = np.random.default_rng(3)
    n = 5000
    w = rng.random(n) ** 8
    q = orc.eng_quantize_weights(w)
    idx = orc.eng_resample_indices(q, 1, k0=0x80000000)
    assert (np.diff(idx) >= 0).all()                         # sorted
    cnt = np.bincount(idx, minlength=n)
    expect = q.astype(np.float64) / q.sum() * n
    assert np.abs(cnt - expect).max() <= 1.0 + 1e-9          # systematic: |count - n*p| < 1
    # order independence: a partition of the children gives the same indices
    a = orc.eng_resample_indices(q, 1, k0=0x80000000, n_children=n)
    assert np.array_equal(a, idx)


def test_chebyshev_and_skip_equivalence(orc, sibal1, sibal1_oracle):
    """The empty-space-skipping march returns the same step as the literal march:
    scalar restatement in numpy on sibal1, against orc_ref_cast_ray."""
    from scipy import ndimage
    m, om = sibal1, sibal1_oracle
    W, H, P = m.width, m.height, om.max_range_px
    occ = m.data > 50
    S = np.ones((H + 1, W + 1), bool)
    S[1:, 1:] = occ; S[0, 1:] = occ[0]; S[1:, 0] = occ[:, 0]; S[0, 0] = occ[0, 0]
    D = ndimage.distance_transform_cdt(~np.pad(S, 1, constant_values=True), metric="chessboard")[1:-1, 1:-1]
    small = S[:40, :40].astype(np.uint8)
    brute = orc.eng_chebyshev_bruteforce(small, 64)
    Dsmall = ndimage.distance_transform_cdt(~np.pad(small.astype(bool), 1, constant_values=True), metric="chessboard")[1:-1, 1:-1]
    assert np.array_equal(brute, np.minimum(Dsmall, 64))
    rng = np.random.default_rng(5)
    n = 3000
    fy, fx = np.nonzero(m.data == 0)
    k = rng.integers(0, fy.size, n)
    x = om.origin_x + (fx[k] + rng.random(n)) * om.resolution
    y = om.origin_y + (fy[k] + rng.random(n)) * om.resolution
    th = rng.uniform(-np.pi, np.pi, n)
    _, want = orc.cast_many(om, x, y, th)
    px0 = (x - om.origin_x) / om.resolution
    py0 = (y - om.origin_y) / om.resolution
    ux, uy = np.cos(th), np.sin(th)
    got = np.full(n, P)
    for i in range(n):
        c0x, c0y = int(np.floor(px0[i])) + 1, int(np.floor(py0[i])) + 1
        s = max(1, min(int(D[c0y, c0x]), 15))
        while s <= P:
            cx = int(np.floor(px0[i] + s * ux[i])) + 1
            cy = int(np.floor(py0[i] + s * uy[i])) + 1
            d = min(int(D[cy, cx]), 15) if (0 <= cx <= W and 0 <= cy <= H) else 0
            if d == 0:
                got[i] = s - 1
                break
            s += d
    assert np.array_equal(got, want)
