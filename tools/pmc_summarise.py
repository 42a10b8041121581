#!/usr/bin/env python3
"""Average every counter of one kernel over the launches found in rocprofv3
counter_collection.csv files below a directory.
   python tools/pmc_summarise.py <dir> [kernel-substring]"""
import collections, csv, glob, os, sys
root, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "spmv_rowblock")
for sub in sorted(os.listdir(root)):
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if pat in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    if acc:
        print(sub, {k: round(sum(v) / len(v), 1) for k, v in sorted(acc.items())}, "launches", len(next(iter(acc.values()))), flush=True)
