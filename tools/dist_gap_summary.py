#!/usr/bin/env python3
"""rocprofv3 kernel_trace.csv of tools/dist_overhead.py -> per kernel: launches, mean duration, mean gap before it (idle time of the
stream between the previous kernel's end and this one's start), over the LAST 2000 dispatches (the distributed code path's loop)."""
import collections, csv, re, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))[-2000:]
acc = collections.OrderedDict()
prev_end = None
for r in rows:
    name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); name = re.sub(r"[<(].*", "", name).replace("void ", "")
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    a = acc.setdefault(name, [0, 0, 0])
    a[0] += 1; a[1] += e - s; a[2] += (s - prev_end) if prev_end is not None else 0
    prev_end = e
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
busy = sum(a[1] for a in acc.values())
print(f"last {len(rows)} dispatches: span {span / 1e6:.3f} ms, kernels busy {busy / 1e6:.3f} ms ({100.0 * busy / span:.1f} %)")
for k, (n, d, g) in acc.items():
    print(f"  {k:40s} x{n:5d}  mean {d / n / 1e3:8.2f} us   mean gap before {g / n / 1e3:7.2f} us")
