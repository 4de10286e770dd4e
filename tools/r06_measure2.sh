#!/bin/bash
# part 2: decoder / network benches, rocprofv3 kernel stats + PMC of the default bench command at fp32 and f16 (c2)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
bash tools/round_measure_b.sh r06
