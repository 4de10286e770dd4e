#!/bin/bash
# Hardware-counter passes over the decoder's kernels (tools/run_decoder.py) on the GPU box.  Usage: tools/pmc_decoder.sh <tag>
set -u
TAG=${1:-pmc_dec}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
GROUPS_=(
 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM"
 "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_SMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS"
 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES"
 "TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TCP_PENDING_STALL_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES"
 "TA_TA_BUSY TA_FLAT_READ_WAVEFRONTS GRBM_GUI_ACTIVE TCC_HIT TCC_MISS"
)
i=0
for g in "${GROUPS_[@]}"; do
  timeout -k 10 240 rocprofv3 --pmc $g --kernel-trace --output-format csv -d $OUT/pass$i -- python3 $ROOT/tools/run_decoder.py 6 1 > $OUT/pass$i.log 2>&1 || echo "pass $i failed (see pass$i.log)"
  i=$((i+1))
done
PMC_KERNELS="k_conv k_se" python3 $ROOT/tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
find $OUT -name '*.csv' -not -name '*counter_collection.csv' -delete
cat $OUT/summary.txt
