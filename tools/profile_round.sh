#!/bin/bash
# Round profile on the GPU box (run through gpurun): kernel-trace stats of the default bench command,
# then the PMC passes (tools/prof_pmc.sh), then the per-launch HBM traffic of the fused kernel
# (FETCH_SIZE doubled for gfx950's wide-read undercount, WRITE_SIZE as is; MI355X_MICROARCH.md §HBM).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r01}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --no-cpu-baseline > $OUT/bench_under_trace.json 2> $OUT/trace.log
bash $ROOT/tools/prof_pmc.sh $TAG/pmc > $OUT/pmc.log 2>&1
python3 $ROOT/bench.py > $OUT/bench.json 2> $OUT/bench.err
python3 - <<PY
import csv, glob, json, os
out = "$OUT"
st = glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True)
rows = list(csv.DictReader(open(st[0]))) if st else []
summ = [{"kernel": r["Name"].split("(")[0], "calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "pct": float(r["Percentage"])} for r in rows]
vals = {}
for line in open(os.path.join(out, "pmc", "summary.txt")):
    if line.startswith("=="):
        cur = line[3:].strip()
    elif "avg/dispatch" in line:
        vals.setdefault(cur, {})[line.split()[0]] = float(line.split("=")[-1])
k = next((k for k in vals if "k_render_fused" in k), None)
traffic = None
if k and "FETCH_SIZE" in vals[k] and "WRITE_SIZE" in vals[k]:
    traffic = (2.0 * vals[k]["FETCH_SIZE"] + vals[k]["WRITE_SIZE"]) * 1024.0   # KB -> bytes, read side doubled (gfx950)
json.dump({"kernel_stats": summ, "pmc_per_dispatch": vals, "traffic_bytes": {"c2:k_render_fused": traffic}}, open(os.path.join(out, "summary.json"), "w"), indent=1)
print("fused kernel HBM traffic per launch (bytes):", traffic)
PY
cat $OUT/bench.json
