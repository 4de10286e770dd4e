#!/bin/bash
# k_prepare's GPU time (rocprofv3 kernel stats of tools/time_prepare.py) for builds of the library whose launch of it carries extra dynamic
# LDS - i.e. fewer workgroups resident per CU, so that the tiles run in several rounds and one round's stores overlap the next one's loads.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/prep_res; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for v in product "$@"; do
  if [ "$v" = product ]; then unset GDB_NERF_LIB; else export GDB_NERF_LIB=$ROOT/gdb-nerf_amd/libgdbnerf_hip.$v.so; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$v -- python3 $ROOT/tools/time_prepare.py > $OUT/$v.log 2>&1
  echo "== $v: $(grep k_prepare $(ls $OUT/$v/*/*kernel_stats.csv | head -1) | cut -d, -f1-4)"
done
