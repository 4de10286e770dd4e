#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc CSVs (one directory per pass) into per-kernel, per-dispatch averages."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))  # kernel -> counter -> [per-dispatch values]
for path in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
    per = defaultdict(float)
    names = {}
    for r in csv.DictReader(open(path)):
        key = (r["Dispatch_Id"], r["Counter_Name"])
        per[key] += float(r["Counter_Value"])
        names[r["Dispatch_Id"]] = r["Kernel_Name"]
    for (d, c), v in per.items():
        acc[names[d].split("(")[0]][c].append(v)
for k in sorted(acc, key=lambda k: -sum(len(v) for v in acc[k].values())):
    if not any(s in k for s in os.environ.get("PMC_KERNELS", "k_render k_pyr k_cam k_mlp k_encode k_prep k_flat").split()):
        continue
    print(f"== {k}")
    for c in sorted(acc[k]):
        v = acc[k][c]
        print(f"   {c:36s} n={len(v):4d}  avg/dispatch = {sum(v) / len(v):16.1f}")
