"""Regression net for the reference-faithful half of the CPU oracle (oracle/mcl_oracle.c orc_ref_*, refdraws.cpp): the
known answers recorded in SURVEY.md Appendix B, which the survey produced by running the reference's own
src/particle_filter.cpp against stub headers.  The reference ships no tests or golden vectors (CMakeLists.txt:126-133)
and a stand-in build is not an admissible pin, so the oracle's status stays PARITY UNPINNED (oracle/mcl_oracle.c header,
DESIGN.md §7); these literals are the only reference-derived numbers there are, and the oracle reproduces every one."""
import numpy as np
import pytest


def test_max_range_px_goes_through_float32(orc, spielberg_oracle, sibal1_oracle):
    # 0.05796 -> float32 -> 207 ; 0.05 -> 0.05000000074505806 -> 239 (not 240)
    assert spielberg_oracle.resolution == 0.057959999889135361
    assert spielberg_oracle.max_range_px == 207
    assert sibal1_oracle.resolution == 0.05000000074505806
    assert sibal1_oracle.max_range_px == 239


def test_sensor_table_known_answers_spielberg(orc):
    T = orc.sensor_table(207)          # T[d, r] == sensor_model_table_(r, d)
    known = {(0, 0): 0.066356471331281711, (10, 10): 0.039510871674593294, (50, 60): 0.01386122001072697,
             (100, 100): 0.020236969092535362, (207, 207): 0.040853676069455724, (207, 0): 0.11476404821172194,
             (0, 207): 0.0076506002804536816}
    for (r, d), v in known.items():
        assert T[d, r] == v, (r, d)
    assert abs(T.sum() - 208.0) < 1e-11          # 208 unit columns (Appendix B: 207.99999999999599)
    np.testing.assert_allclose(T.sum(axis=1), 1.0, rtol=0, atol=1e-13)


def test_sensor_table_known_answers_sibal1(orc):
    T = orc.sensor_table(239)
    assert T[0, 0] == 0.066229217520362121
    assert abs(T.sum() - 240.0) < 1e-11


def _f32(x):
    return float(np.float32(x))


def test_synthetic_scan_known_answers(orc, spielberg_oracle, sibal1_oracle):
    ang = orc.beam_angles()
    assert ang.size == 1081 and ang.dtype == np.float32
    z = np.zeros(ang.size)
    r, _ = orc.cast_many(spielberg_oracle, z, z, ang.astype(np.float64))
    want = [2.20248008, 1.04328001, 4.23108006, 1.39103997, 1.27512002]
    got = [float(r[i]) for i in (0, 270, 540, 810, 1080)]
    assert [f"{g:.8f}" for g in got] == [f"{w:.8f}" for w in want]
    assert f"{float(r.astype(np.float64).sum()):.5f}" == "2513.58912"
    r, _ = orc.cast_many(sibal1_oracle, z, z, ang.astype(np.float64))
    want = [1.55000007, 1.20000005, 9.10000038, 1.20000005, 1.64999998]          # Appendix B
    got = [float(r[i]) for i in (0, 270, 540, 810, 1080)]
    assert [f"{g:.8f}" for g in got] == [f"{w:.8f}" for w in want]
    assert f"{float(r.astype(np.float64).sum()):.5f}" == "2099.25004"


def test_full_chain_known_answers(orc, spielberg_oracle):
    """Appendix B 'Full chain': N=2000, angle_step=18 (61 beams), seed 42, init cloud at (0,0,0),
    one MCL(action=(0.05,0,0.01)) with the origin scan — includes the libstdc++ draw semantics."""
    om = spielberg_oracle
    full = orc.beam_angles()
    z = np.zeros(full.size)
    scan, _ = orc.cast_many(om, z, z, full.astype(np.float64))
    ang, obs = orc.beam_angles(angle_step=18), scan[::18].copy()
    assert ang.size == 61
    N = 2000
    s = orc.RefStream(42)
    p, w = orc.init_particles_pose(s, (0.0, 0.0, 0.0), N)
    assert list(p[:, 0]) == [-0.27511724721024677, 0.25771653484560064, 0.18954434226648892]
    u, nrm = s.uniforms(N), s.normals(3 * N).reshape(N, 3)
    T = orc.sensor_table(om.max_range_px)
    out = orc.mcl_step(om, p, w, (0.05, 0.0, 0.01), ang, obs, T, u, nrm)
    P, W = out["particles"], out["weights"]
    assert list(P[:, 0]) == [-0.033223950649317921, 0.34465170716028698, -0.74871940806065618]
    assert W[0] == 1.0946110511321877e-14
    assert W.max() == 0.22817876493613812
    # first three ranges of particle 0 / particle 1 (float metres in the reference: step*res, 12 on a miss)
    res = om.resolution
    r0 = [_f32(s_ * res) for s_ in out["steps"][0, :3]]
    assert [f"{v:.5f}" for v in r0] == ["3.30372", "5.10048", "10.83852"]
    assert out["steps"][1, 0] == om.max_range_px                                   # "12"
    r1 = [_f32(s_ * res) for s_ in out["steps"][1, 1:3]]
    assert [f"{v:.5f}" for v in r1] == ["11.07036", "7.07112"]
    pose = orc.expected_pose(P, W)
    assert list(pose) == [0.02491833902169947, 0.011643671276452626, 9.4908712839202129e-05]
    seq = 0.0
    for v in W:
        seq += v
    assert seq == 0.99999999999999878                                              # sequential sum, as printed


def test_underflow_at_1081_beams(orc, spielberg_oracle):
    """the double product underflows to 0 for every particle at 1081 beams."""
    z = np.load(__import__("os").path.join(__import__("conftest").GOLDEN, "g4_underflow_B1081.npz"))
    assert z["ref_raw_weights"].max() == 0.0 and z["ref_weights"].sum() == 0.0
    assert np.isfinite(z["eng_logw"]).all()
