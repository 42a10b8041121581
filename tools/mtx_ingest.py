#!/usr/bin/env python3
"""Large `.mtx` ingestion figure (round-4 verdict, item 8; reference: sparse_matrix.hpp:225-357 -- MatrixCOO::read_from_mtx,
symmetric expansion, stable sort by row -- and utilities.hpp:326-367 convert_coo_to_crs): config 5 names a SuiteSparse file
"if supplied on the box"; this writes the unstructured config-5 input itself (unstr:80,80,80: 1,536,000 rows, 1.04e8
entries) as MatrixMarket -- general, and symmetric-lower in the column-major layout of the SuiteSparse files -- and times the
host CLI's input phase on it: read + parse -> COO -> CRS -> binary cache -> upload, then the `-cache` reload.  The CRS the CLI
built (its cache file) must equal the generator's bit for bit.

    python tools/mtx_ingest.py [nx ny nz] [--dir DIR] [--keep]       prints one JSON line
"""
import ctypes as C
import json
import os
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from oracle import pyoracle  # noqa: E402

BIN = os.path.join(ROOT, "basic_iterative_solvers_amd", "host", "basic_iterative_solvers")


def read_cache(path):
    """the CLI's binary CRS cache: 6 x int64 header, row_ptr int64[n + 1], col int32[nnz], val float64[nnz]"""
    with open(path, "rb") as f:
        h = np.fromfile(f, dtype=np.int64, count=6)
        n, nnz = int(h[1]), int(h[3])
        rp = np.fromfile(f, dtype=np.int64, count=n + 1)
        col = np.fromfile(f, dtype=np.int32, count=nnz)
        val = np.fromfile(f, dtype=np.float64, count=nnz)
    return rp, col, val


def run_cli(mtx, cache):
    t0 = time.time()
    out = subprocess.run([BIN, mtx, "-cg", "-p", "j", "-cache", cache], capture_output=True, text=True, timeout=1200)
    wall = time.time() - t0
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
    line = next(ln for ln in out.stdout.splitlines() if ln.startswith("Matrix input:"))
    nums = {k.strip(" ,").replace(" ", "_"): float(v) for k, v in re.findall(r"([A-Za-z.][A-Za-z .+>\-]*?) ([0-9]+\.[0-9]+) s", line.split(":", 1)[1])}
    m = re.search(r"(converged in: |did not converge after )(\d+) iterations", out.stdout)
    return {"wall_s": wall, "input_line": line, "phases_s": nums, "iterations": int(m.group(2)) if m else None}


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    shape = tuple(int(a) for a in args[:3]) if len(args) >= 3 else (80, 80, 80)
    keep = "--keep" in sys.argv
    d = sys.argv[sys.argv.index("--dir") + 1] if "--dir" in sys.argv else tempfile.mkdtemp(prefix="bis_mtx_", dir="/tmp")
    os.makedirs(d, exist_ok=True)
    threads = int(os.environ.get("BIS_CPU_THREADS", "16"))
    pyoracle.set_omp_threads(threads)
    orc = pyoracle.Oracle()
    orc.lib.orc_write_mtx.restype = C.c_int64
    t0 = time.time()
    A = orc.gen_unstr(*shape)
    rec = {"matrix": "unstr:%d,%d,%d" % shape, "rows": A.n_rows, "nnz": A.nnz, "generate_on_host_s": time.time() - t0, "writer_threads": threads}
    S = A.to_scipy()
    assert abs(S - S.T).max() == 0.0  # symmetric in pattern and values: the symmetric file can represent it
    del S
    for kind, sym in (("general", 0), ("symmetric", 1)):
        mtx, cache = os.path.join(d, f"unstr_{kind}.mtx"), os.path.join(d, f"unstr_{kind}.crs")
        for p in (mtx, cache):
            if os.path.exists(p):
                os.remove(p)
        t0 = time.time()
        stored = orc.lib.orc_write_mtx(mtx.encode(), C.c_int64(A.n_rows), A.row_ptr.ctypes, A.col.ctypes, A.val.ctypes, C.c_int(sym))
        assert stored > 0
        r = {"file_bytes": os.path.getsize(mtx), "stored_entries": int(stored), "write_s": time.time() - t0}
        r["first_run"] = run_cli(mtx, cache)          # .mtx -> COO -> CRS -> cache -> upload
        rp, col, val = read_cache(cache)
        r["crs_bit_identical_to_generator"] = bool(np.array_equal(rp, A.row_ptr) and np.array_equal(col, A.col) and np.array_equal(val, A.val))
        del rp, col, val
        r["cache_bytes"] = os.path.getsize(cache)
        r["second_run_from_cache"] = run_cli(mtx, cache)
        rec[kind] = r
        if not keep:
            os.remove(mtx); os.remove(cache)
    print(json.dumps(rec))
    assert rec["general"]["crs_bit_identical_to_generator"] and rec["symmetric"]["crs_bit_identical_to_generator"]


if __name__ == "__main__":
    main()
