#!/bin/bash
# Second half of a round's measurements (the first: tools/round_measure_a.sh): decoder / whole-network benches, then the rocprofv3
# kernel-trace stats + PMC passes of the default bench command at fp32 and f16 (tools/profile_round.sh).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-r04}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT
echo "== decoder / network";      timeout -k 10 300 python3 tools/bench_decoder.py > $OUT/bench_decoder.json 2> $OUT/bench_decoder.err
timeout -k 10 300 python3 tools/bench_network.py > $OUT/bench_network.json 2> $OUT/bench_network.err
echo "== profile";                PRECS="f32 f16" bash tools/profile_round.sh $TAG c2 > $OUT/profile.log 2>&1
tail -3 $OUT/profile.log
