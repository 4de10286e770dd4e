#!/bin/bash
# VALU / VMEM / LDS instruction counts of k_render_fused with parts ablated (GDB_FUSED_SKIP bits: 1 colours, 2 features,
# 4 volume, 8 MLP).  The ablation switches exist only in the diagnostic build (-DGDB_DIAG -DGDB_DEBUG_SKIP), which is built
# BESIDE the product library as libgdbnerf_hip.skip.so and selected through GDB_NERF_LIB (run on the GPU box through gpurun).
# Usage: tools/valu_split.sh [bench args, e.g. --precision f16]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/valu_split; mkdir -p $OUT
(cd $ROOT && python3 gdb-nerf_amd/build.py --tag skip --extra "-DGDB_DIAG -DGDB_DEBUG_SKIP" > $OUT/build.log 2>&1) || { echo "diagnostic build failed"; exit 1; }
export GDB_NERF_LIB=$ROOT/gdb-nerf_amd/libgdbnerf_hip.skip.so
cd /tmp; export TMPDIR=/tmp
for sk in 0 7 8 15; do
  GDB_FUSED_SKIP=$sk timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --kernel-trace --output-format csv -d $OUT/skip$sk -- python3 $ROOT/bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 1 --prewarm-ms 0 "$@" > /dev/null 2>&1
  echo "skip=$sk"; python3 $ROOT/tools/pmc_summary.py $OUT/skip$sk 2>/dev/null | grep -A9 k_render_fused | grep -v "=="
done
