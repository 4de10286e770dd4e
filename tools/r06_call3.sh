#!/bin/bash
# round 6, third GPU call: the round's new tests (smooth c4 / c5 frames, prepare_rows, sources-ready, bundle_size 1 / 4 fused), the f16 gather
# A/B (select-free level offsets) against the library at HEAD~, c5 f16 with the sources-ready record, bundle_size 4 pricing
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r06c; mkdir -p $OUT; cd $ROOT
echo "== pytest new tests"; timeout -k 10 700 python3 -m pytest tests/test_hip_parity.py tests/test_network_surface.py -m gpu -x -q -s -k "bundle_size or smooth_feature_c4_c5 or prepare_rows or sources_ready or pyr16_without or schedules_agree or row_strips" > $OUT/pytest_new.txt 2>&1; echo "rc=$?"; tail -3 $OUT/pytest_new.txt
grep -h "smooth c\|fused bundle_size\|F7d" $OUT/pytest_new.txt | head -40
echo "== b4"; timeout -k 10 200 python3 tools/bench_b4.py 200 > $OUT/bench_b4.json 2> $OUT/bench_b4.err; tail -2 $OUT/bench_b4.err; grep -h "ms_per_step\|per_ray\"" $OUT/bench_b4.json
echo "== ab f16 level_off"; timeout -k 10 500 python3 tools/ab_libs.py --libs base,product --cases c2:f16:0,c3:f16:0,c4:f16:0,c5:f16:0,c2:f32:0 --steps 300 --reps 3 > $OUT/ab_f16_select_free_level_offsets.txt 2>&1
tail -7 $OUT/ab_f16_select_free_level_offsets.txt
echo "== c5 f16 bench with sources-ready record"; timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-extras --sources-ready-record --workload c5 --precision f16 --steps 200 --warmup 20 > $OUT/bench_c5_f16.json 2> $OUT/bench_c5.err; tail -2 $OUT/bench_c5.err
python3 - <<PY
import json
d = json.load(open("$OUT/bench_c5_f16.json"))
print("c5 f16 ms/step", d["ms_per_step"], "kernel_ms", d["roofline"]["kernel_ms"], "sources_ready", json.dumps(d.get("prepare_sources_ready"))[:300])
PY
