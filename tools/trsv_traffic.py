#!/usr/bin/env python3
"""gpurun_out/trsv_traffic_<tag>/ (tools/trsv_traffic.sh) -> profiles/trsv_traffic.json: HBM bytes per sweep of every
workload / direction of bench.py's `sweeps` legs.  Counted: every dispatch of the sweep kernels and their sentinel fills
(plan-building kernels are setup), summed over the run and divided by the sweeps made.  FETCH_SIZE is corrected by the factor
calibrated on this repo's streaming kernels (profiles/spmv_traffic.json: gfx950 tallies 128-byte requests at 64 bytes); the
sweeps' 8-byte polls are narrower accesses, for which that factor is an upper bound -- so `hbm_bytes_per_sweep` is an upper
estimate of the fabric bytes, and `fetch_uncorrected` is kept beside it.
    python tools/trsv_traffic.py <dir> <sweeps per run> <spmv_traffic.json>"""
import collections
import csv
import json
import os
import re
import sys

SWEEP = re.compile(r"(trsv_tiled_kernel|trsv_chain_kernel|chain_fill_kernel|sptrsv_wave_kernel|sptrsv_syncfree_kernel|fill_sentinel_kernel|"
                   r"trsv_level_kernel|tiled_fill\w*|spmv_rowblock_kernel)")


def total(path):
    per = collections.defaultdict(float)
    n = collections.Counter()
    for r in csv.DictReader(open(path)):
        m = SWEEP.search(r["Kernel_Name"])
        if m:
            per[m.group(1)] += float(r["Counter_Value"])
            n[m.group(1)] += 1
    return per, n


def main():
    d, n_sweeps, cal = sys.argv[1], int(sys.argv[2]), json.load(open(sys.argv[3]))
    ff, wf = cal["fetch_correction"], cal["write_correction"]
    out = {"sweeps_per_run": n_sweeps, "fetch_correction": ff, "write_correction": wf, "sweeps": {}}
    for f in sorted(os.listdir(d)):
        m = re.match(r"(\w+)_(forward|backward)_FETCH_SIZE\.csv$", f)
        if not m:
            continue
        key, direction = m.groups()
        wfile = os.path.join(d, f"{key}_{direction}_WRITE_SIZE.csv")
        if not os.path.exists(wfile):
            continue
        fe, nf = total(os.path.join(d, f))
        wr, _ = total(wfile)
        rd_kib, wr_kib = sum(fe.values()) / n_sweeps, sum(wr.values()) / n_sweeps
        out["sweeps"].setdefault(key, {})[direction] = {
            "kernels": {k: int(v / n_sweeps) if v >= n_sweeps else v / n_sweeps for k, v in nf.items()},
            "fetch_size_kib_per_sweep": rd_kib, "write_size_kib_per_sweep": wr_kib,
            "fetch_uncorrected_bytes_per_sweep": rd_kib * 1024,
            "hbm_read_bytes_per_sweep": rd_kib * 1024 * ff, "hbm_write_bytes_per_sweep": wr_kib * 1024 * wf,
            "hbm_bytes_per_sweep": rd_kib * 1024 * ff + wr_kib * 1024 * wf}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
