#!/bin/bash
# A/B helper: link libgdbnerf_hip.<tag>.so from the product objects with ONE source file replaced by another version of it
# (e.g. an earlier commit's).  Usage: tools/build_variant.sh <tag> <file.hip in csrc> <path of the replacement source>
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TAG=$1; NAME=$2; SRC=$3
CS=$ROOT/gdb-nerf_amd/csrc
TMP=$(mktemp -d)
cp $SRC $TMP/$NAME
OBJ=$TMP/${NAME%.hip}.o
# (the contraction mode of each translation unit as gdb-nerf_amd/build.py has it: the fused kernels' is fast-honor-pragmas, the rest off)
CONTRACT=off; [ "$NAME" = gdb_fused.hip ] && CONTRACT=fast-honor-pragmas
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -Wall -Wno-unused-function -ffp-contract=$CONTRACT -I$CS -I$ROOT/include -c $TMP/$NAME -o $OBJ
OBJS=""
for f in gdb_ops gdb_mlp gdb_fused gdb_costvol gdb_merge gdb_decoder; do
  if [ $f.hip = $NAME ]; then OBJS="$OBJS $OBJ"; else OBJS="$OBJS $CS/obj/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o $ROOT/gdb-nerf_amd/libgdbnerf_hip.$TAG.so
rm -rf $TMP
echo $ROOT/gdb-nerf_amd/libgdbnerf_hip.$TAG.so
