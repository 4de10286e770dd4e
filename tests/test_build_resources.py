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
# (k_render_dense became a persistent tile loop in round 4: the tile cursor and the row / window / bundle range of the tile live in
# SGPRs across gather + MLP, a few of which the allocator parks in VGPR lanes: 9-13 today)
SGPR_SPILL_BOUND = {"k_render_fusedILb0": 0, "k_render_dense": 24, "k_render_flat": 32, "k_render_solo": 40, "k_render_fusedILb1": 176}


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
    seen = set()
    for name, u in usage["gdb_fused.hip"].items():
        if ("k_render_fusedILb0ELi4E" in name or "k_render_denseILi1E" in name or "k_render_denseILi0ELi3E" in name or "k_render_denseILi2ELi3E" in name
                or "k_render_soloILi0ELi3E" in name):   # (the last: the f16 segment-wave build c5 takes - V = 5 stages 11,520 B per wave: its 12 waves per CU are set by these registers, not by LDS)
            assert u["vgprs"] <= 168 and u["waves_per_simd"] >= 3, (name, u)
            seen.add(name.split("I")[0])
    assert len(seen) >= 3, seen


def test_staging_sizes_the_launcher_counts_on():
    """LDS per wave (gdb_fused.hip stage_v<>): rows per view x 128 B.  f16: 18 rows (6 packed colour rows + 10 packed rows of feature
    channel pairs + 2 packed direction rows), fp32 / split-f16: 35.  With 1280-byte LDS granules: V = 3 holds 12 waves per CU at every
    precision (two-wave workgroups at fp32), V = 5 at f16 holds 14 by LDS (one-wave workgroups) - so the 12 its registers allow run;
    with fp32 feature rows (27 rows, rounds 3-4) it was 9 - the numbers DESIGN.md quotes."""
    gran, cap = 1280, 160 * 1024
    waves = lambda lds, n: n * (cap // (-(-n * lds // gran) * gran))
    f16, f32 = 18 * 128, 35 * 128
    assert max(waves(3 * f32, 1), waves(3 * f32, 2)) == 12 and waves(3 * f32, 1) == 11
    assert waves(3 * f16, 1) >= 12
    assert waves(5 * f16, 1) == 14 and waves(5 * 27 * 128, 1) == 9 and waves(5 * f32, 1) == 7
    src = open(os.path.join(build.CSRC, "gdb_fused.hip")).read()
    assert "PREC == GDB_PREC_F16 ? 6 : 12" in src and "PREC == GDB_PREC_F16 ? 10 : GDB_CFR" in src   # row_feat<> / feat_rows<>: the layout the arithmetic above assumes
