#!/bin/bash
# Hardware-counter passes over `bench.py` on the GPU box (run through gpurun).  One rocprofv3 run
# per counter group (PMC slots are limited; FETCH_SIZE / WRITE_SIZE need passes of their own), only
# --kernel-trace beside --pmc.  Usage: tools/prof_pmc.sh <tag> [bench args...]
set -u
TAG=${1:-pmc}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
GROUPS_=(
 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM"
 "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_SMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS"
 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM"
 "FETCH_SIZE"
 "WRITE_SIZE TCC_HIT TCC_MISS"
 "TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TCP_PENDING_STALL_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES"
 "TA_TA_BUSY TA_FLAT_READ_WAVEFRONTS GRBM_GUI_ACTIVE GRBM_TA_BUSY"
 "SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_ACTIVE_INST_VALU2 SQ_THREAD_CYCLES_VALU SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_MISC SQ_LEVEL_WAVES SQ_INSTS_VALU_MFMA_MOPS_F32"
)
i=0
for g in "${GROUPS_[@]}"; do
  timeout -k 10 240 rocprofv3 --pmc $g --kernel-trace --output-format csv -d $OUT/pass$i -- python3 $ROOT/bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 --prewarm-ms 0 "$@" > $OUT/pass$i.log 2>&1 || echo "pass $i failed (see pass$i.log)"
  i=$((i+1))
done
python3 $ROOT/tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
