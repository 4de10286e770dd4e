cd /tmp; export TMPDIR=/tmp; OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_costvol; mkdir -p $OUT
i=0
for g in "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "FETCH_SIZE" "WRITE_SIZE TCC_HIT TCC_MISS" "TA_TA_BUSY GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TCP_PENDING_STALL_CYCLES"; do
  timeout -k 10 200 rocprofv3 --pmc $g --kernel-trace --output-format csv -d $OUT/p$i -- python3 $GRAFT_REPO_ROOT/tools/bench_costvol.py > /dev/null 2>&1; i=$((i+1))
done
python3 - <<PY
import csv, glob, os
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for path in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    per = defaultdict(float); names = {}; grid = {}
    for r in csv.DictReader(open(path)):
        per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"]); names[r["Dispatch_Id"]] = r["Kernel_Name"]; grid[r["Dispatch_Id"]] = r["Grid_Size"]
    for (d, c), v in per.items():
        if "k_costvol<" in names[d]: acc[grid[d]][c].append(v)
for g in acc:
    print("grid", g)
    for c in sorted(acc[g]): print("   %-32s %14.1f" % (c, sum(acc[g][c]) / len(acc[g][c])))
PY
