"""The host patch as an artefact: integration/particle_filter_hip.patch is a unified diff against the
reference's src/particle_filter.cpp, include/particle_filter_cpp/particle_filter.hpp and CMakeLists.txt.  Where the
reference checkout exists (this container) the diff must apply cleanly; everywhere, every C-ABI symbol it calls must be
declared in include/mcl_hip_engine.h and exported by the built library, and the class surface must be untouched apart
from the four private additions."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATCH = os.path.join(ROOT, "integration", "particle_filter_hip.patch")
REFERENCE = "/root/reference"


def patch_text():
    return open(PATCH).read()


def test_patch_touches_only_the_three_files():
    files = re.findall(r"^\+\+\+ b/(\S+)", patch_text(), flags=re.M)
    assert files == ["CMakeLists.txt", "include/particle_filter_cpp/particle_filter.hpp", "src/particle_filter.cpp"]


def test_every_engine_call_is_declared_in_the_header():
    added = "\n".join(l[1:] for l in patch_text().splitlines() if l.startswith("+") and not l.startswith("+++"))
    used = set(re.findall(r"\b(mcl_[a-z_]+)\s*\(", added))
    header = open(os.path.join(ROOT, "include", "mcl_hip_engine.h")).read()
    declared = set(re.findall(r"\b(mcl_[a-z_0-9]+)\s*\(", header))
    assert used and used <= declared, sorted(used - declared)
    assert {"mcl_create", "mcl_destroy", "mcl_set_map", "mcl_set_beam_angles", "mcl_set_particles", "mcl_update",
            "mcl_expected_pose", "mcl_get_stage_timings", "mcl_sample_particles", "mcl_particle_mean"} <= used


def test_class_surface_unchanged():
    """No public member, no existing private declaration and no signature is removed from the header: the only removed
    lines of the whole diff are the bodies of MCL() / expected_pose() and the visualisation draw they replace."""
    hunks = patch_text().split("+++ b/")
    hpp = next(h for h in hunks if h.startswith("include/particle_filter_cpp/particle_filter.hpp"))
    removed = [l for l in hpp.splitlines() if l.startswith("-") and not l.startswith("---")]
    assert removed == []
    added = [l[1:].strip() for l in hpp.splitlines() if l.startswith("+") and not l.startswith("+++") and l[1:].strip()]
    assert [a for a in added if not a.startswith("//")] == [
        '#include "mcl_hip_engine.h"', "~ParticleFilter() override;",
        "mcl_engine_t *engine_ = nullptr;  // libmcl_hip_engine.so: resample, motion, ray cast, table lookup, normalise, pose",
        "bool host_state_stale_ = false;   // particles_ / weights_ lag behind the engine until sync_host_state()",
        "void sync_host_state();", "void push_host_state();"]


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="the reference checkout only exists in the build container")
def test_patch_applies_to_the_reference():
    r = subprocess.run(["git", "apply", "--check", "--verbose", PATCH], cwd=REFERENCE, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
