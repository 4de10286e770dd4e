#!/bin/bash
# k_plan alone and k_prepare under the bench, per library tag:  tools/prof_plan.sh product [tag ...]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out
for t in "$@"; do
  if [ "$t" = product ]; then unset GDB_NERF_LIB; else export GDB_NERF_LIB=$ROOT/gdb-nerf_amd/libgdbnerf_hip.$t.so; fi
  rm -rf /tmp/tp_$t /tmp/tr_$t; cd /tmp
  TMPDIR=/tmp timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tp_$t -- python3 $ROOT/tools/time_plan.py > /dev/null 2> $ROOT/gpurun_out/prof_plan_$t.err
  TMPDIR=/tmp timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_$t -- python3 $ROOT/bench.py --no-cpu-baseline --no-extras --steps 1000 > /dev/null 2>> $ROOT/gpurun_out/prof_plan_$t.err
  echo "== $t"; grep -h "k_plan" /tmp/tp_$t/*/*kernel_stats.csv > /tmp/o_$t.txt; sed -n 2,3p /tmp/tr_$t/*/*kernel_stats.csv >> /tmp/o_$t.txt; cut -d, -f1-4 /tmp/o_$t.txt
done
