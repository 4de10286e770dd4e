#!/bin/bash
# A/B of compile-time variants on the GPU box: each flag set is built BESIDE the product library (libgdbnerf_hip.ab<i>.so,
# selected through GDB_NERF_LIB; an empty flag set "" is the product flags), then the fused parity + determinism tests, then
# three bench runs.  Usage: [AB_BENCH_ARGS="--precision f16"] tools/ab_flags.sh "<flags A>" "<flags B>" ...
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
i=0
for F in "$@"; do
  python gdb-nerf_amd/build.py --tag ab$i --extra "$F" > gpurun_out/ab_build_$i.log 2>&1 || { echo "[$F] build failed"; tail -5 gpurun_out/ab_build_$i.log; exit 1; }
  export GDB_NERF_LIB=$PWD/gdb-nerf_amd/libgdbnerf_hip.ab$i.so
  timeout -k 10 300 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "fused" > gpurun_out/ab_tests_$i.log 2>&1
  rc=$?
  echo "[$F] tests: $(tail -1 gpurun_out/ab_tests_$i.log)"
  [ $rc -eq 0 ] || { grep -E "assert|Error|FAILED" gpurun_out/ab_tests_$i.log | head -5; i=$((i+1)); continue; }
  for r in 1 2 3; do
    timeout -k 10 120 python bench.py --no-cpu-baseline --no-extras ${AB_BENCH_ARGS:-} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('[$F]  G rays/s', round(d['value']/1e9,3), ' kernel us', round(d['roofline']['kernel_ms']*1e3,1))"
  done
  i=$((i+1))
done
