#!/bin/bash
# round 6, fourth GPU call: new tests (sources-ready, bundle_size 1 / 4 fused, partial pyramid for strips), strip-prepare timing, then the full GPU suite
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r06d; mkdir -p $OUT; cd $ROOT
echo "== pytest new tests"; timeout -k 10 700 python3 -m pytest tests/test_hip_parity.py tests/test_network_surface.py -m gpu -q -s -k "bundle_size or prepare_rows or sources_ready or pyr16_without or row_strips" > $OUT/pytest_new.txt 2>&1; echo "rc=$?"; tail -5 $OUT/pytest_new.txt
grep -h "fused bundle_size\|F7d\|partial pyramid" $OUT/pytest_new.txt | head -60
echo "== strip prepare timing"; timeout -k 10 400 python3 tools/time_prepare_rows.py 8 200 > $OUT/time_prepare_rows_world8.json 2> $OUT/time_prepare_rows.err; tail -2 $OUT/time_prepare_rows.err
python3 - <<PY
import json
d = json.load(open("$OUT/time_prepare_rows_world8.json"))
for k, v in d.items():
    if isinstance(v, dict): print(k, "whole", round(v["whole_frame_prepare_us"], 1), "strip mean", round(v["strip_prepare_us_mean"], 1), "max", round(v["strip_prepare_us_max"], 1), "ratio", round(v["strip_over_whole"], 2), "shares", [round(s["pyramid_share_written"], 2) for s in v["strips"]])
PY
echo "== full gpu suite"; timeout -k 10 800 python3 -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.txt 2>&1; echo "rc=$?"; tail -4 $OUT/pytest_gpu.txt
