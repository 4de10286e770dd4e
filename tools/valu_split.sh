#!/bin/bash
# VALU / VMEM / LDS instruction counts of k_render_fused with parts ablated (GDB_FUSED_SKIP bits: 1 colours, 2 features,
# 4 volume, 8 MLP).  The ablation switches exist only in a diagnostic build, so the fused TU is rebuilt with
# -DGDB_DEBUG_SKIP first (run on the GPU box through gpurun; the box's copy of the library is what changes).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/valu_split; mkdir -p $OUT
touch $ROOT/gdb-nerf_amd/csrc/gdb_fused.hip
(cd $ROOT && GDB_HIPCC_EXTRA="-DGDB_DEBUG_SKIP" python3 gdb-nerf_amd/build.py > $OUT/build.log 2>&1) || { echo "diagnostic build failed"; exit 1; }
cd /tmp; export TMPDIR=/tmp
for sk in 0 7 8 15; do
  GDB_FUSED_SKIP=$sk timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --kernel-trace --output-format csv -d $OUT/skip$sk -- python3 $ROOT/bench.py --no-cpu-baseline --steps 5 --warmup 1 --prewarm-ms 0 > /dev/null 2>&1
  echo "skip=$sk"; python3 $ROOT/tools/pmc_summary.py $OUT/skip$sk 2>/dev/null | grep -A9 k_render_fused | grep -v "=="
done
