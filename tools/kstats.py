#!/usr/bin/env python3
"""Print the k_* rows of a rocprofv3 kernel_stats.csv (name, calls, average / min / max in us)."""
import csv, sys
for path in sys.argv[1:]:
    print("==", path)
    for r in csv.DictReader(open(path)):
        n = r["Name"]
        if n.startswith("void k_") or n.startswith("k_"):
            print(f"{n.split('(')[0][:44]:44s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.2f} us  min {float(r['MinNs'])/1e3:8.2f}  max {float(r['MaxNs'])/1e3:8.2f}")
