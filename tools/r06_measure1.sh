#!/bin/bash
# round 6 final measurements, part 1: bench lines, all workloads, stamps
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
bash tools/round_measure_a.sh r06
