#!/bin/bash
# round 6, second GPU call: the GPU test suite with the round's new tests, record-padding / acquire A/B, the alternating-frame soak, a bench line
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r06b; mkdir -p $OUT; cd $ROOT
echo "== pytest -m gpu"; timeout -k 10 700 python3 -m pytest tests -m gpu -x -q -s > $OUT/pytest_gpu.txt 2>&1; echo "rc=$?"; tail -4 $OUT/pytest_gpu.txt
grep -h "smooth\|alternating\|bias" $OUT/pytest_gpu.txt | head -20
echo "== ab record padding / acquire"; timeout -k 10 400 python3 tools/ab_libs.py --libs rec24,product,acq --cases c2:f32:4,c2:f16:4,c2@6f:f32:4,c4:f32:4 --steps 400 --reps 3 > $OUT/ab_flat_record_128B_and_acquire.txt 2>&1
tail -6 $OUT/ab_flat_record_128B_and_acquire.txt
echo "== soak"; timeout -k 10 300 python3 tools/soak.py 30000 > $OUT/soak.txt 2>&1; tail -6 $OUT/soak.txt
echo "== bench"; timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 > $OUT/bench_driver_like.json 2> $OUT/bench.err; tail -2 $OUT/bench.err
python3 - <<PY
import json
d = json.load(open("$OUT/bench_driver_like.json"))
print("value", d["value"] / 1e9, "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "scaling", d["scaling"])
for k in ("bundle_size_4", "prepare_sources_ready", "prepare_sources_ready_f16", "hipgraph_replay"):
    print(k, json.dumps(d.get(k))[:600])
PY
