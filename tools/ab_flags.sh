#!/bin/bash
# A/B of compile-time variants of gdb_fused.hip on the GPU box: for each flag set rebuild the fused TU, run the fused
# parity + determinism tests, then three bench runs.  Usage: tools/ab_flags.sh "<flags A>" "<flags B>" ...
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
i=0
for F in "$@"; do
  touch gdb-nerf_amd/csrc/gdb_fused.hip
  GDB_HIPCC_EXTRA="$F" python gdb-nerf_amd/build.py > gpurun_out/ab_build_$i.log 2>&1 || { echo "[$F] build failed"; tail -5 gpurun_out/ab_build_$i.log; exit 1; }
  timeout -k 10 300 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "fused" > gpurun_out/ab_tests_$i.log 2>&1
  rc=$?
  echo "[$F] tests: $(tail -1 gpurun_out/ab_tests_$i.log)"
  [ $rc -eq 0 ] || { grep -E "assert|Error|FAILED" gpurun_out/ab_tests_$i.log | head -5; i=$((i+1)); continue; }
  for r in 1 2 3; do
    timeout -k 10 120 python bench.py --no-cpu-baseline ${AB_BENCH_ARGS:-} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('[$F]  G rays/s', round(d['value']/1e9,3), ' kernel us', round(d['roofline']['kernel_ms']*1e3,1))"
  done
  i=$((i+1))
done
