#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r06g; mkdir -p $OUT; cd $ROOT
echo "== decoder / b4 tests"; timeout -k 10 600 python3 -m pytest tests/test_decoder.py tests/test_network_surface.py -m gpu -q -s > $OUT/pytest_dec.txt 2>&1; echo "rc=$?"; tail -3 $OUT/pytest_dec.txt; grep -h "upscale_factor 4\|F7d" $OUT/pytest_dec.txt
echo "== b4 bench"; timeout -k 10 300 python3 tools/bench_b4.py 200 > $OUT/bench_b4.json 2> $OUT/bench_b4.err; tail -2 $OUT/bench_b4.err; python3 -c "
import json; d=json.load(open('$OUT/bench_b4.json')); print(d.get('b4_decoder')); print(d['b4'].get('fused_ms_per_step'))"
echo "== full gpu suite"; timeout -k 10 800 python3 -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.txt 2>&1; echo "rc=$?"; tail -3 $OUT/pytest_gpu.txt
