#!/usr/bin/env python3
"""Turn the rocprofv3 --pmc passes (separate runs for FETCH_SIZE and
WRITE_SIZE, MI355X_MICROARCH.md "HBM"/"rocprofv3 PMC slots") into per-launch
HBM bytes for the SpMV kernel, with the gfx950 correction calibrated on this
repo's own streaming kernels (cg_p_update / cg_update: known byte counts).

  python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_csv> [tcc_csv] <size> > profiles/spmv_traffic.json
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
    fetch, write = load(sys.argv[1]), load(sys.argv[2])
    tcc = load(sys.argv[3]) if len(sys.argv) > 4 else {}
    size = int(sys.argv[-1])
    N = size ** 3
    vec_kib = 8 * N / 1024
    # calibration: FETCH_SIZE (KiB) vs the true read bytes of 16 B/lane streaming kernels
    cal = {}
    # round-2 schedule (no preconditioner): pass C reads z (= r), p, x and writes x, p; pass B reads tmp, r and writes r
    for k, n_read, n_write in (("cg_p_update_kernel", 3, 2), ("cg_update_kernel", 2, 1)):
        cal[k] = dict(fetch_factor=n_read * vec_kib / fetch[(k, "FETCH_SIZE")],
                      write_factor=n_write * vec_kib / write[(k, "WRITE_SIZE")])
    ff = sum(c["fetch_factor"] for c in cal.values()) / len(cal)
    wf = sum(c["write_factor"] for c in cal.values()) / len(cal)
    # the SpMV kernel of the run: the value-dictionary kernel where the matrix has one, else the CRS-value kernel
    K = next(k for k in ("spmv_rowmajor_vd_kernel", "spmv_rowblock_vd_kernel", "spmv_rowblock_kernel") if (k, "FETCH_SIZE") in fetch)
    rd = fetch[(K, "FETCH_SIZE")] * 1024 * ff
    wr = write[(K, "WRITE_SIZE")] * 1024 * wf
    nnz = (3 * size - 2) ** 3
    out = dict(size=size, kernel=K,
               fetch_size_kib=fetch[(K, "FETCH_SIZE")],
               write_size_kib=write[(K, "WRITE_SIZE")],
               calibration=cal, fetch_correction=ff, write_correction=wf,
               hbm_read_bytes_per_launch=rd, hbm_write_bytes_per_launch=wr,
               hbm_bytes_per_launch=rd + wr, algorithmic_bytes_per_launch=12 * nnz + 20 * N,
               ratio_to_algorithmic=(rd + wr) / (12 * nnz + 20 * N))
    if tcc:
        h, m = tcc[(K, "TCC_HIT_sum")], tcc[(K, "TCC_MISS_sum")]
        out["l2_hit_rate"] = h / (h + m)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
