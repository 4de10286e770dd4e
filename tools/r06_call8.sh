#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r06h; mkdir -p $OUT; cd $ROOT
echo "== full gpu suite"; timeout -k 10 900 python3 -m pytest tests -m gpu -q -x -s > $OUT/pytest_gpu.txt 2>&1; echo "rc=$?"; tail -3 $OUT/pytest_gpu.txt
echo "== smoke"; timeout -k 10 200 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -6
echo "== bench"; timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 > $OUT/bench_driver_like.json 2> $OUT/bench.err; tail -1 $OUT/bench.err
python3 - <<PY
import json
d = json.load(open("$OUT/bench_driver_like.json"))
print("value", d["value"] / 1e9, "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "traffic", d["roofline"]["traffic"], d["roofline"]["traffic_source"][:60], "scaling", d["scaling"])
PY
