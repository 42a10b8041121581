#!/usr/bin/env python3
"""Turn the rocprofv3 --pmc passes (separate runs for FETCH_SIZE and
WRITE_SIZE, MI355X_MICROARCH.md "HBM"/"rocprofv3 PMC slots") into per-launch
HBM bytes for every SpMV kernel of the run, keyed by kernel name, with the
gfx950 correction calibrated on this repo's own streaming kernels
(cg_p_update / cg_update: known byte counts).  A run without those kernels
(stand-alone SpMV passes) takes the correction of an earlier file.

  python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_csv> [tcc_csv] <size> [--correction-from traffic.json]
      > profiles/spmv_traffic.json
"""
import collections
import csv
import json
import re
import sys


def load(path):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        m = re.search(r"(\w+_kernel)", r["Kernel_Name"])
        d[(m.group(1) if m else r["Kernel_Name"][:40], r["Counter_Name"])].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in d.items()}


def main():
    argv = sys.argv[1:]
    prev = None
    if "--correction-from" in argv:
        i = argv.index("--correction-from")
        prev = json.load(open(argv[i + 1]))
        del argv[i:i + 2]
    fetch, write = load(argv[0]), load(argv[1])
    tcc = load(argv[2]) if len(argv) > 3 else {}
    size = int(argv[-1])
    N = size ** 3
    vec_kib = 8 * N / 1024
    # calibration: FETCH_SIZE (KiB) vs the true read bytes of 16 B/lane streaming kernels
    # (no preconditioner): pass C reads z (= r), p, x and writes x, p; pass B reads tmp, r and writes r
    cal = {}
    for k, n_read, n_write in (("cg_p_update_kernel", 3, 2), ("cg_update_kernel", 2, 1)):
        if (k, "FETCH_SIZE") in fetch and (k, "WRITE_SIZE") in write:
            cal[k] = dict(fetch_factor=n_read * vec_kib / fetch[(k, "FETCH_SIZE")],
                          write_factor=n_write * vec_kib / write[(k, "WRITE_SIZE")])
    if cal:
        ff = sum(c["fetch_factor"] for c in cal.values()) / len(cal)
        wf = sum(c["write_factor"] for c in cal.values()) / len(cal)
    elif prev:
        ff, wf = prev["fetch_correction"], prev["write_correction"]
    else:
        raise SystemExit("no calibration kernels in this run and no --correction-from file")
    out = dict(size=size, calibration=cal or "taken from an earlier file", fetch_correction=ff, write_correction=wf, kernels={})
    for K in sorted({k for k, c in fetch if k.startswith("spmv_") and c == "FETCH_SIZE"}):
        if (K, "WRITE_SIZE") not in write:
            continue
        rd = fetch[(K, "FETCH_SIZE")] * 1024 * ff
        wr = write[(K, "WRITE_SIZE")] * 1024 * wf
        e = dict(fetch_size_kib=fetch[(K, "FETCH_SIZE")], write_size_kib=write[(K, "WRITE_SIZE")],
                 hbm_read_bytes_per_launch=rd, hbm_write_bytes_per_launch=wr, hbm_bytes_per_launch=rd + wr)
        if (K, "TCC_HIT_sum") in tcc:
            h, m = tcc[(K, "TCC_HIT_sum")], tcc[(K, "TCC_MISS_sum")]
            e["l2_hit_rate"] = h / (h + m)
        out["kernels"][K] = e
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
