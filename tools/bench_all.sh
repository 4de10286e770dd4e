#!/bin/bash
# All BASELINE workloads x the three precisions on the GPU box -> one JSON array (copy into profiles/<round>/bench_all_workloads.json).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
echo "["
first=1
for wl in c1 c2 c3 c3p c4 c5; do for P in f32 f32x f16; do
  [ $first = 1 ] || echo ","
  first=0
  python3 $ROOT/bench.py --no-cpu-baseline --no-extras --workload $wl --precision $P --steps ${BENCH_STEPS:-500} --warmup 100 2>/dev/null
done; done
echo "]"
