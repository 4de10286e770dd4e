"""Build-time properties of the HIP kernels, read from hipcc's kernel-resource-usage remarks (gdb-nerf_amd/build.py writes
them to csrc/obj/resource_usage.json on every build; no GPU needed)."""
import json
import os

import pytest

from gdb_nerf_amd import build


@pytest.fixture(scope="module")
def usage():
    build.build()
    path = os.path.join(build.CSRC, "obj", "resource_usage.json")
    if not os.path.exists(path):
        build.build(force=True)
    return json.load(open(path))


def test_fused_kernels_keep_their_data_out_of_private_memory(usage):
    """Round 1's "packed f32 corrupts lanes 48..63" was a private-memory (scratch) round trip of a weight struct that hipcc
    introduced to form op_sel operand pairs (reproducer -DGDB_XP_PK=1).  EVERY instantiation of every fused
    kernel must have a zero-size private segment: no vector spill and no struct parked in scratch (SGPR spills go to VGPR lanes,
    not to memory, and are bounded below so that a regression shows)."""
    fused = {k: v for k, v in usage["gdb_fused.hip"].items() if "k_render" in k}
    assert len(fused) >= 12, sorted(fused)     # 3 precisions x (3 slot-wave variants + segment-wave + dense kernels)
    for name, u in fused.items():
        assert u["scratch_bytes_per_lane"] == 0, (name, u)
        assert u["vgpr_spill"] == 0, (name, u)


# SGPR spills per kernel family (to VGPR lanes: no memory traffic): today's values as upper bounds (k_render_fused<true, ...> is the
# slot loop for S_max > 8, where every wave-uniform value lives across the loop body; the segment-wave kernel re-derives them per slot)
SGPR_SPILL_BOUND = {"k_render_fusedILb0": 0, "k_render_dense": 0, "k_render_solo": 40, "k_render_fusedILb1": 176}


def test_scalar_spills_stay_bounded_and_the_compiler_is_recorded(usage):
    for name, u in usage["gdb_fused.hip"].items():
        for fam, bound in SGPR_SPILL_BOUND.items():
            if fam in name:
                assert u["sgpr_spill"] <= bound, (name, u)
    assert "hipcc" in usage.get("_toolchain", {}) and "clang" in usage["_toolchain"]["hipcc"].lower(), usage.get("_toolchain")


def test_occupancy_the_schedules_are_designed_for(usage):
    """Registers per lane decide waves per SIMD (MI355X guide: <= 168 -> 3 waves, <= 256 -> 2): the one-slot-per-wave kernels
    run three waves per SIMD at all three precisions, and so does the dense kernel c2 / c3 / c4 take at fp32 and at f16
    (its split-f16 form is built in both a 3- and a 2-wave variant)."""
    for name, u in usage["gdb_fused.hip"].items():
        if "k_render_fusedILb0ELi4E" in name or "k_render_denseILi1E" in name or "k_render_denseILi0ELi3E" in name or "k_render_denseILi2ELi3E" in name:
            assert u["vgprs"] <= 168 and u["waves_per_simd"] >= 3, (name, u)
