#!/bin/bash
# Build the library as it stood at an earlier commit, beside the product library, for same-box A/B runs (tools/ab_libs.py --libs TAG,product):
#   tools/build_baseline.sh <commit> <tag>     ->  gdb-nerf_amd/libgdbnerf_hip.<tag>.so   (git-ignored; travels to the GPU box)
set -eu
COMMIT=${1:?commit}; TAG=${2:?tag}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d /tmp/gdb_baseline.XXXXXX)
mkdir -p $TMP/gdb-nerf_amd/csrc $TMP/include
for f in $(git -C $ROOT ls-tree --name-only $COMMIT gdb-nerf_amd/csrc/ | grep -E '\.(hip|h)$'); do git -C $ROOT show $COMMIT:$f > $TMP/$f; done
git -C $ROOT show $COMMIT:include/gdb_nerf_hip.h > $TMP/include/gdb_nerf_hip.h
git -C $ROOT show $COMMIT:gdb-nerf_amd/build.py > $TMP/gdb-nerf_amd/build.py
(cd $TMP && python3 -c "import sys; sys.path.insert(0, \"gdb-nerf_amd\"); import build; print(build.build(force=True))" > $TMP/build.log 2>&1) || { tail -20 $TMP/build.log; exit 1; }
cp $TMP/gdb-nerf_amd/libgdbnerf_hip.so $ROOT/gdb-nerf_amd/libgdbnerf_hip.$TAG.so
echo "built $ROOT/gdb-nerf_amd/libgdbnerf_hip.$TAG.so from $COMMIT"
rm -rf $TMP
