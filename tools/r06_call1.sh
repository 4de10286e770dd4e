#!/bin/bash
# round 6, first GPU call: hand-off acquire A/B, HIP-graph probe, bundle_size 4 pricing, soak of the acquire build
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r06a; mkdir -p $OUT; cd $ROOT
echo "== ab acquire"; timeout -k 10 500 python3 tools/ab_libs.py --libs product,acq --cases c2:f32:4,c2:f16:4,c2@6f:f32:4,c4:f32:4,c3:f32:4,c2:f32:3 --steps 400 --reps 3 > $OUT/ab_flat_acquire.txt 2>&1
tail -8 $OUT/ab_flat_acquire.txt
echo "== graph probe"; timeout -k 10 200 python3 tools/graph_probe.py f32 1000 > $OUT/graph_probe_f32.json 2> $OUT/graph_probe.err; tail -3 $OUT/graph_probe.err
echo "== b4"; timeout -k 10 200 python3 tools/bench_b4.py 200 > $OUT/bench_b4.json 2> $OUT/bench_b4.err; tail -3 $OUT/bench_b4.err
echo "== soak (acquire build)"; GDB_NERF_LIB=$ROOT/gdb-nerf_amd/libgdbnerf_hip.acq.so timeout -k 10 200 python3 tools/soak.py 30000 > $OUT/soak_acquire_build.txt 2>&1; tail -2 $OUT/soak_acquire_build.txt
