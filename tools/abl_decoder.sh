#!/bin/bash
# Ablation timing of the decoder's kernels: rocprofv3 kernel stats of tools/run_decoder.py for the product library and each
# libgdbnerf_hip.<tag>.so given.  Usage: tools/abl_decoder.sh tag1 tag2 ...   -> gpurun_out/dec/abl_<tag>.csv
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/dec
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for TAG in product "$@"; do
  if [ $TAG = product ]; then unset GDB_NERF_LIB; else export GDB_NERF_LIB=$ROOT/gdb-nerf_amd/libgdbnerf_hip.$TAG.so; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t_$TAG -- python3 $ROOT/tools/run_decoder.py 20 1 > $OUT/abl_$TAG.log 2>&1
  cp $(find $OUT/t_$TAG -name '*kernel_stats.csv' | head -1) $OUT/abl_$TAG.csv
  rm -rf $OUT/t_$TAG
  python3 $ROOT/tools/kstats.py $OUT/abl_$TAG.csv | grep -v "k_prepare"
done
