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
    introduced to form op_sel operand pairs (reproducer -DGDB_XP_PK=1).  Every instantiation of the c2-class
    kernel must have a zero-size private segment, and no fused kernel may spill vector registers."""
    fused = {k: v for k, v in usage["gdb_fused.hip"].items() if "k_render" in k}
    assert len(fused) >= 12, sorted(fused)     # 3 precisions x (3 slot-wave variants + segment-wave + dense kernels)
    for name, u in fused.items():
        if "k_render_fused" in name:
            assert u["scratch_bytes_per_lane"] == 0, (name, u)
        assert u["vgpr_spill"] == 0, (name, u)


def test_occupancy_the_schedules_are_designed_for(usage):
    """Registers per lane decide waves per SIMD (MI355X guide: <= 168 -> 3 waves, <= 256 -> 2): the one-slot-per-wave kernels
    run three waves per SIMD at all three precisions."""
    for name, u in usage["gdb_fused.hip"].items():
        if "k_render_fusedILb0ELi4E" in name:
            assert u["vgprs"] <= 168 and u["waves_per_simd"] >= 3, (name, u)
