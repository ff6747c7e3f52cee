"""The C-ABI library loads on a CPU-only box, exports every symbol include/mcl_hip_engine.h declares,
its config struct matches the ctypes mirror, and creating an engine without a GPU fails loudly
(no silent CPU fallback)."""
import ctypes
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "mcl_hip_engine.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mcl_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(engine_mod):
    lib = engine_mod.load_library()
    names = declared_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in mcl_hip_engine.h but not exported"
    assert set(engine_mod.EXPORTS) == set(names)
    lib.mcl_abi_version.restype = ctypes.c_int
    assert lib.mcl_abi_version() == 1


def test_config_struct_layout_matches_header(engine_mod, tmp_path):
    probe = tmp_path / "probe.c"
    probe.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "mcl_hip_engine.h"\n'
                     'int main(void){printf("%zu %zu %zu %zu %zu %zu\\n", sizeof(mcl_config_t), offsetof(mcl_config_t, seed),'
                     'offsetof(mcl_config_t, squash_factor), offsetof(mcl_config_t, resample_mode),'
                     'offsetof(mcl_config_t, keep_ray_steps), offsetof(mcl_config_t, reserved));return 0;}\n')
    exe = tmp_path / "probe"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(probe), "-o", str(exe)])
    got = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    C = engine_mod.Config
    want = [ctypes.sizeof(C), C.seed.offset, C.squash_factor.offset, C.resample_mode.offset, C.keep_ray_steps.offset,
            C.reserved.offset]
    assert got == want


def test_default_config_is_the_reference_yaml(engine_mod):
    c = engine_mod.default_config()
    assert (c.max_particles, c.max_range_m, c.squash_factor) == (2000, 12.0, 2.2)
    assert (c.z_hit, c.z_short, c.z_max, c.z_rand, c.sigma_hit) == (0.80, 0.01, 0.07, 0.12, 8.0)
    assert (c.motion_dispersion_x, c.motion_dispersion_y, c.motion_dispersion_theta) == (0.05, 0.025, 0.25)
    assert c.resample_mode == engine_mod.RESAMPLE_MULTINOMIAL and c.weight_mode == engine_mod.WEIGHT_LOG


def test_no_gpu_means_loud_failure(engine_mod):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(engine_mod.EngineError) as ei:
        engine_mod.Engine(max_particles=16)
    assert "no HIP device" in str(ei.value) or "rc=-4" in str(ei.value)


def test_particle_total_bound_is_checked_before_anything_else(engine_mod):
    """weights are 2^-36 fixed point summed in 64 bits: 2^27 particles (or more) per engine / per group are refused with an
    argument error, on a box without a GPU too (the check precedes the device probe)"""
    with pytest.raises(engine_mod.EngineError) as ei:
        engine_mod.Engine(max_particles=1 << 27)
    assert "rc=-1" in str(ei.value)
    with pytest.raises(engine_mod.EngineError) as ei:
        engine_mod.Group([0, 0], max_particles=1 << 26)
    assert "rc=-1" in str(ei.value)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "monte_carlo_localization_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp")):
                txt = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", txt, flags=re.M), f
                assert "mcl_oracle" not in txt or f.endswith("mcl_device_math.h"), f


def test_legacy_build_exports_the_same_abi_and_the_product_refuses_its_kernels(engine_mod):
    """libmcl_hip_engine_legacy.so (the same sources with -DMCL_LEGACY_RAY_KERNELS: k_rays_quad / k_rays_cell, test
    infrastructure) exports the same symbols; the PRODUCT library knows the two kernel classes only as 'not built in'."""
    leg = engine_mod.load_library(legacy=True)
    for n in declared_functions():
        assert hasattr(leg, n), n
    out = subprocess.check_output(["nm", "-D", "--defined-only", engine_mod.LIB_PATH]).decode()
    assert "k_rays_sweep" in out and "k_rays_quad" not in out and "k_rays_cell" not in out
    out = subprocess.check_output(["nm", "-D", "--defined-only", engine_mod.LEGACY_LIB_PATH]).decode()
    assert "k_rays_quad" in out and "k_rays_cell" in out
