#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r06f; mkdir -p $OUT; cd $ROOT
echo "== stagger"; GDB_NERF_LIB=$ROOT/gdb-nerf_amd/libgdbnerf_hip.diag.so timeout -k 10 500 python3 tools/xp_stagger.py 300 3 > $OUT/xp_start_stagger.txt 2>&1; cat $OUT/xp_start_stagger.txt | grep -v amdgpu
echo "== batch"; timeout -k 10 300 python3 tools/xp_batch.py 200 > $OUT/xp_frames_per_launch.txt 2>&1; cat $OUT/xp_frames_per_launch.txt | grep -v amdgpu
echo "== new tests"; timeout -k 10 400 python3 -m pytest tests/test_hip_parity.py -m gpu -q -k "prepare_rows" 2>&1 | tail -2
