#!/bin/bash
# First half of a round's measurements: the driver-like and default bench lines, all workloads x precisions, phase / residency stamps.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-r04}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT
echo "== driver-like (20 steps)"; timeout -k 10 500 python3 bench.py --steps 20 --warmup 5 > $OUT/bench_driver_like_20_steps.json 2> $OUT/bench_driver_like.err
echo "== default";                timeout -k 10 500 python3 bench.py --no-cpu-baseline > $OUT/bench_default.json 2> $OUT/bench_default.err
echo "== all workloads";          BENCH_STEPS=300 timeout -k 10 900 bash tools/bench_all.sh > $OUT/bench_all_workloads.json 2> $OUT/bench_all.err
echo "== stamps";                 for s in 3 4; do timeout -k 10 120 python3 tools/stamps.py --schedule=$s 2>&1 | grep -v -i "warn\|amdgpu.ids" > $OUT/stamps_f32_schedule$s.txt; done
timeout -k 10 120 python3 tools/stamps.py --schedule=3 f16 2>&1 | grep -v -i "warn\|amdgpu.ids" > $OUT/stamps_f16_schedule3.txt
python3 - <<PY
import json
for f in ("bench_driver_like_20_steps", "bench_default"):
    try:
        d = json.load(open("$OUT/" + f + ".json"))
        print(f, "value", round(d["value"] / 1e9, 3), "G rays/s  ms/step", round(d["ms_per_step"], 4), "kernel_ms", round(d["roofline"]["kernel_ms"], 4), "frac", round(d["roofline"]["frac"], 3))
    except Exception as e:
        print(f, "unreadable:", e)
PY
