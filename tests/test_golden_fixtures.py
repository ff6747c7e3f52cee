"""The committed fixtures under tests/golden/ are oracle outputs; this keeps oracle and fixtures in step
(regenerate with tools/make_golden.py)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN


def test_g1_tables(orc):
    for P in (207, 239):
        z = np.load(os.path.join(GOLDEN, f"g1_sensor_table_P{P}.npz"))
        assert np.array_equal(z["table"], orc.sensor_table(P))


@pytest.mark.parametrize("name", ["Spielberg_map", "sibal1"])
def test_g2_cast_ray(orc, maps_mod, name):
    m = maps_mod.load_npz(os.path.join(GOLDEN, f"map_{name}.npz"))
    om = orc.OracleMap(m.data, m.resolution, m.origin_x, m.origin_y)
    z = np.load(os.path.join(GOLDEN, f"g2_cast_ray_{name}.npz"))
    r, s = orc.cast_many(om, z["x"], z["y"], z["theta"])
    assert np.array_equal(r, z["ranges"]) and np.array_equal(s, z["steps"])
    # the identity the engine relies on: range_idx == step, P on a miss
    P = om.max_range_px
    rpx = (r.astype(np.float64) / om.resolution).astype(np.float32)
    rpx = np.minimum(rpx, np.float32(P))
    assert np.array_equal(np.clip(np.round(rpx).astype(np.int64), 0, P), s)


@pytest.mark.parametrize("B", [61, 121])
def test_g3_step(orc, spielberg_oracle, B):
    z = np.load(os.path.join(GOLDEN, f"g3_mcl_step_B{B}.npz"))
    T = orc.sensor_table(spielberg_oracle.max_range_px)
    r = orc.mcl_step(spielberg_oracle, z["particles_in"], z["weights_in"], z["action"], z["angles"], z["obs"], T,
                     z["uniforms"], z["normals"])
    assert np.array_equal(r["idx"], z["idx"])
    assert np.array_equal(r["particles"], z["particles_out"])
    assert np.array_equal(r["steps"], z["steps"])
    assert np.array_equal(r["weights"], z["weights_out"])
    assert np.array_equal(orc.expected_pose(r["particles"], r["weights"]), z["pose"])


def test_scan_fixtures(orc, spielberg_oracle):
    z = np.load(os.path.join(GOLDEN, "scan_Spielberg_map_origin.npz"))
    ang = orc.beam_angles()
    assert np.array_equal(ang, z["angles"])
    r, s = orc.cast_many(spielberg_oracle, np.zeros(ang.size), np.zeros(ang.size), ang.astype(np.float64))
    assert np.array_equal(r, z["ranges"]) and np.array_equal(s, z["steps"])
