#!/bin/bash
# part 3: rocprofv3 kernel stats + PMC of c5 f16 and c4 f32; soak, stress, two-rank gloo rehearsal
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r06; mkdir -p $OUT; cd $ROOT
echo "== c5 f16 profile"; PRECS=f16 bash tools/profile_round.sh r06_c5 c5 > $OUT/profile_c5.log 2>&1; tail -2 $OUT/profile_c5.log
echo "== c4 f32 profile"; PRECS=f32 bash tools/profile_round.sh r06_c4 c4 > $OUT/profile_c4.log 2>&1; tail -2 $OUT/profile_c4.log
echo "== stress"; timeout -k 10 300 python3 tools/stress_fused.py 100 7 > $OUT/stress_fused.txt 2>&1; tail -1 $OUT/stress_fused.txt
timeout -k 10 300 python3 tools/stress_bundle_sizes.py 100 11 > $OUT/stress_bundle_sizes.txt 2>&1; tail -2 $OUT/stress_bundle_sizes.txt
echo "== rehearsal"; GDB_BENCH_REHEARSE=1 timeout -k 10 500 python3 bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_gpus2_gloo_rehearsal.json 2> $OUT/rehearsal.err; tail -2 $OUT/rehearsal.err
python3 - <<PY
import json
d = json.load(open("$OUT/bench_gpus2_gloo_rehearsal.json"))
print("rehearsal: scaling", d["scaling"], "n_gpus", d["n_gpus"], "gathered", d.get("gathered_equals_full_render"), "prepare_ms", d.get("prepare_ms"), [(k, v.get("prepare_ms"), v.get("kernel_ms")) for k, v in d.items() if k.startswith("rows_c")])
PY
