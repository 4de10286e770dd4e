#!/bin/bash
# Round profile on the GPU box (run through gpurun): kernel-trace stats of the default bench command (fp32 headline) and of
# the f16 path, then the PMC passes of both (tools/prof_pmc.sh), then the per-launch HBM traffic of the fused kernel
# (FETCH_SIZE doubled for gfx950's wide-read undercount, WRITE_SIZE as is; MI355X_MICROARCH.md §HBM).
# Usage: tools/profile_round.sh <tag> [workload]      -> gpurun_out/<tag>/ ; copy what is to be judged into profiles/<tag>/
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r02}
WL=${2:-c2}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for P in ${PRECS:-f32 f32x f16}; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$P -- python3 $ROOT/bench.py --no-cpu-baseline --no-extras --workload $WL --precision $P > $OUT/bench_under_trace_$P.json 2> $OUT/trace_$P.log
  bash $ROOT/tools/prof_pmc.sh $TAG/pmc_$P --workload $WL --precision $P > $OUT/pmc_$P.log 2>&1
done
python3 - <<PY
import csv, glob, json, os
out, wl = "$OUT", "$WL"
res = {"kernel_stats": {}, "pmc_per_dispatch": {}, "traffic_bytes": {}}
for p in ("f32", "f32x", "f16"):
    st = glob.glob(os.path.join(out, "trace_" + p, "**", "*kernel_stats.csv"), recursive=True)
    rows = list(csv.DictReader(open(st[0]))) if st else []
    res["kernel_stats"][p] = [{"kernel": r["Name"].split("(")[0], "calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "pct": float(r["Percentage"])} for r in rows]
    vals, cur = {}, None
    sp = os.path.join(out, "pmc_" + p, "summary.txt")
    for line in (open(sp) if os.path.exists(sp) else []):
        if line.startswith("=="):
            cur = line[3:].strip()
        elif "avg/dispatch" in line:
            vals.setdefault(cur, {})[line.split()[0]] = float(line.split("=")[-1])
    res["pmc_per_dispatch"][p] = vals
    for k in vals:
        if "k_render" in k and "FETCH_SIZE" in vals[k] and "WRITE_SIZE" in vals[k]:
            name = "k_render_solo" if "solo" in k else "k_render_dense" if "dense" in k else "k_render_flat" if "flat" in k else "k_render_fused"
            res["traffic_bytes"][f"{wl}:{name}:{p}"] = (2.0 * vals[k]["FETCH_SIZE"] + vals[k]["WRITE_SIZE"]) * 1024.0   # KB -> bytes, read side doubled (gfx950)
json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1)
print("fused kernel HBM traffic per launch (bytes):", res["traffic_bytes"])
# traffic.json for bench.py's roofline.traffic, stamped with the kernel source it was measured on (bench.py refuses it otherwise)
import sys
sys.path.insert(0, "$ROOT")
from bench import kernel_source_sha16   # gdb_fused.hip + gdb_internal.h + gdb_ops.hip, comments stripped: what bench.py checks
sha = kernel_source_sha16()
tj = dict(res["traffic_bytes"])
tj["_kernel_source_sha256_16"] = sha
tj["_source"] = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py, per launch: 2*FETCH_SIZE + WRITE_SIZE, KB -> bytes (MI355X guide, HBM section); per key: _sources"
# the file each key's counters were read from (bench.py's roofline.traffic_source names the one of ITS key: VERDICT r05 item 10)
tj["_sources"] = {k: "profiles/$TAG/pmc_" + k.rsplit(":", 1)[1] + "/summary.txt (workload " + k.split(":")[0] + ")" for k in res["traffic_bytes"]}
json.dump(tj, open(os.path.join(out, "traffic.json"), "w"), indent=1)
print("traffic.json written to", out, "- copy it to profiles/traffic.json (the file bench.py reads) together with the summaries")
PY
