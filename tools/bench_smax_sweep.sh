#!/bin/bash
# S_max sweep on the c2 shape ("sweep S_max in {3, 6, 8} and state which is reported"): adaptive and fixed
# counts x the three precisions -> one JSON array (copy into profiles/<round>/smax_sweep.json).  The headline stays c2 as
# configs/dtu_eval.yaml has it: S_max 3 adaptive.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
echo "["
first=1
for S in 3 6 8; do for M in adaptive fixed; do for P in f32 f32x f16; do
  [ $first = 1 ] || echo ","
  first=0
  python3 $ROOT/bench.py --no-cpu-baseline --no-extras --workload c2 --smax $S --sampling $M --precision $P --steps ${BENCH_STEPS:-300} --warmup 50 2>/dev/null
done; done; done
echo "]"
