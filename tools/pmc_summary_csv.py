#!/usr/bin/env python3
"""rocprofv3 counter_collection.csv -> kernel,counter,dispatches,mean,max (one line per kernel and counter).
   python tools/pmc_summary_csv.py <counter_collection.csv> > summary.csv"""
import collections, csv, re, sys
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"(\w+)\s*(<|\()", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", ""))
    acc[(m.group(1) if m else r["Kernel_Name"][:40], r["Counter_Name"])].append(float(r["Counter_Value"]))
print("kernel,counter,dispatches,mean,max")
for (k, c), v in sorted(acc.items()):
    print(f"{k},{c},{len(v)},{sum(v) / len(v)},{max(v)}")
