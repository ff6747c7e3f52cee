import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "lds_windows: (tests/test_gpu_sweep_global.py) the test sets MCL_SWEEP_GLOBAL itself, if at all")


@pytest.fixture(scope="session")
def orc():
    from oracle import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def maps_mod():
    from monte_carlo_localization_amd import maps
    return maps


@pytest.fixture(scope="session")
def spielberg(maps_mod):
    return maps_mod.load_npz(os.path.join(GOLDEN, "map_Spielberg_map.npz"))


@pytest.fixture(scope="session")
def sibal1(maps_mod):
    return maps_mod.load_npz(os.path.join(GOLDEN, "map_sibal1.npz"))


@pytest.fixture(scope="session")
def spielberg_oracle(orc, spielberg):
    return orc.OracleMap(spielberg.data, spielberg.resolution, spielberg.origin_x, spielberg.origin_y)


@pytest.fixture(scope="session")
def sibal1_oracle(orc, sibal1):
    return orc.OracleMap(sibal1.data, sibal1.resolution, sibal1.origin_x, sibal1.origin_y)


@pytest.fixture(scope="session")
def engine_mod():
    """The HIP engine binding; building is part of the fixture so a stale .so never hides a change."""
    import __graft_entry__ as g
    g.build()
    from monte_carlo_localization_amd import engine
    return engine


def make_engine(engine_mod, m, angles, n, **cfg):
    e = engine_mod.Engine(max_particles=n, **cfg)
    e.set_map(m.data, m.resolution, m.origin_x, m.origin_y)
    e.set_beam_angles(angles)
    return e


def tracking_cloud(rng, n, pose=(0.0, 0.0, 0.0), sig=(0.5, 0.5, 0.4)):
    p = np.empty((3, n))
    p[0] = pose[0] + rng.normal(0, sig[0], n)
    p[1] = pose[1] + rng.normal(0, sig[1], n)
    th = pose[2] + rng.normal(0, sig[2], n)
    p[2] = (th + np.pi) % (2 * np.pi) - np.pi
    return p


def block_digests(a, block=1 << 20):
    """crc32 of every `block` entries along the last axis (rows of a 2-D array separately): arrays of tens of millions of
    entries are compared through these instead of being written out.  Shards that are a whole number of blocks long
    concatenate to the digests of the unsharded array."""
    import zlib
    a = np.ascontiguousarray(a)
    rows = a.reshape(-1, a.shape[-1])
    return np.array([[zlib.crc32(r[i:i + block].tobytes()) for i in range(0, r.size, block)] for r in rows], dtype=np.uint32)
