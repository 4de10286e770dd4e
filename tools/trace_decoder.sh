#!/bin/bash
# Per-kernel times of the decoder (N1) on the GPU box: rocprofv3 kernel stats of tools/bench_decoder.py.
# Usage: tools/trace_decoder.sh <tag> [lib]   -> gpurun_out/dec/<tag>.{json,csv}
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-base}
[ -n "${2:-}" ] && export GDB_NERF_LIB=$2
OUT=$ROOT/gpurun_out/dec
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$TAG -- python3 $ROOT/tools/bench_decoder.py > $OUT/$TAG.json 2> $OUT/$TAG.err
cp $(find $OUT/trace_$TAG -name '*kernel_stats.csv' | head -1) $OUT/$TAG.csv
rm -rf $OUT/trace_$TAG
python3 $ROOT/tools/kstats.py $OUT/$TAG.csv
cat $OUT/$TAG.json
