#!/bin/bash
# Everything a round's DESIGN / profiles numbers come from, in one gpurun call:  tools/round_measure.sh <tag>
#   -> gpurun_out/<tag>/: driver-like and default bench lines, all workloads x precisions, rocprofv3 kernel stats + PMC passes
#      (tools/profile_round.sh), occupancy / phase stamps of the fp32 kernel, the decoder and whole-network benches.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r04}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
echo "== driver-like (20 steps)"; timeout -k 10 500 python3 bench.py --steps 20 --warmup 5 > $OUT/bench_driver_like_20_steps.json 2> $OUT/bench_driver_like.err
echo "== default";                timeout -k 10 500 python3 bench.py --no-cpu-baseline > $OUT/bench_default.json 2> $OUT/bench_default.err
echo "== all workloads";          BENCH_STEPS=300 timeout -k 10 900 bash tools/bench_all.sh > $OUT/bench_all_workloads.json 2> $OUT/bench_all.err
echo "== stamps";                 for s in 1 3 4; do timeout -k 10 120 python3 tools/stamps.py --schedule=$s 2>&1 | grep -v -i warn > $OUT/stamps_f32_schedule$s.txt; done
timeout -k 10 120 python3 tools/stamps.py --schedule=3 f16 2>&1 | grep -v -i warn > $OUT/stamps_f16_schedule3.txt
echo "== small-operand probe";    timeout -k 10 200 python3 tools/probe_split_f16_small.py > $OUT/split_f16_small_operands.txt 2>&1
echo "== decoder / network";      timeout -k 10 300 python3 tools/bench_decoder.py > $OUT/bench_decoder.json 2> $OUT/bench_decoder.err
timeout -k 10 300 python3 tools/bench_network.py > $OUT/bench_network.json 2> $OUT/bench_network.err
echo "== profile";                bash tools/profile_round.sh $TAG c2 > $OUT/profile.log 2>&1
tail -3 $OUT/profile.log
python3 - <<PY
import json
for f in ("bench_driver_like_20_steps", "bench_default"):
    try:
        d = json.load(open("$OUT/" + f + ".json"))
        print(f, "value", round(d["value"] / 1e9, 3), "G rays/s  ms/step", round(d["ms_per_step"], 4), "kernel_ms", round(d["roofline"]["kernel_ms"], 4), "frac", round(d["roofline"]["frac"], 3))
    except Exception as e:
        print(f, "unreadable:", e)
PY
