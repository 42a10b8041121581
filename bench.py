#!/usr/bin/env python3
"""bench.py -- CG iterations/s (+ SpMV GFLOP/s and % of the HBM roofline) on
the HPCG 27-point operator, the metric BASELINE.json names.

    python bench.py --gpus N --steps K --warmup W [--size 256] [--precond none|j]

A "step" is one CG iteration (methods/cg.hpp:6-54 + the residual sample of
:162-166) on synthetic HPCG-`size` (size^3 rows, 27-point, b = 1, x0 = 0.1),
generated directly in HBM; inputs are resident before the timed region.
N > 1: one process per GPU (torch.distributed.run), the matrix 1-D
row-partitioned (z-slabs), strong scaling: the global problem is fixed.

Prints ONE JSON line (rank 0).  The timed region -- `value`, `ms_per_step`,
`roofline` -- is the path BASELINE.json's north_star names: CRS SpMV inside the
CG loop with the CRS value array streamed (8 bytes per non-zero; option
spmv_valdict = 0), the SpMV launches priced with HIP events recorded on the
library's stream: `achieved` = SURVEY 8d's algorithmic bytes (12 nnz + 20 N)
over the average launch time, `traffic` = the PMC-measured HBM bytes per launch
(profiles/spmv_traffic*.json), `moved_*` = the same time priced on those bytes.
`compressed_stream` = the same loop on the same arrays with the library's
default stream format for this matrix (a lossless re-encoding for matrices with
<= 256 distinct values: bit-identical y, fewer bytes -- a speed-up figure, not
a statement about CRS bandwidth); `target_512` = both at the north-star size,
`config5_spmv` = the unstructured stand-in's SpMV; `cpu_baseline`
(+ `cpu_baseline_socket`, `cpu_baseline_first_touch`) time the reference's own
CG (oracle/_ref) on the host cores for a bounded number of iterations, and
`parity_max_dr_over_r0` compares a GPU run of the same length with that history
(N = 1, rank 0 only).

The PRECONDITIONED half of the path (round 5): `sweeps` = forward / backward
bis_sptrsv (native_sptrsv / native_bsptrsv, kernels.hpp:54-117) on HPCG-256,
Anderson-256, fem:80,80,81 and unstr:80,80,80 (as generated and RCM-ordered),
HIP-event timed on the library's stream, each with a roofline on 12 nnz_T + 28 N
(SURVEY 8d), the kernel that ran and the PMC traffic of profiles/trsv_traffic.json;
`cpu_baseline_sptrsv` = the reference's serial native_sptrsv on HPCG-128 beside
it; `configs` = BASELINE configs 2, 4, 5 through the host CLI (the library's
GMRES / BiCGSTAB schedules) with iterations, iterate time and the timer tree's
SpMV / preconditioner / BLAS-1 split; `options` = the library options in effect.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# the CPUs this process may use, read before any OpenMP runtime binds the main thread to its first place
try:
    CPUS_ALLOWED = len(os.sched_getaffinity(0))
except Exception:
    CPUS_ALLOWED = None

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--size", type=int, default=256, help="HPCG grid edge (256 = metric size)")
    ap.add_argument("--precond", default="none", choices=["none", "j"])
    ap.add_argument("--matrix", default="hpcg", choices=["hpcg", "anderson"],
                    help="hpcg: the BASELINE metric; anderson (+ --precond j): BASELINE configs 2/3")
    ap.add_argument("--shift", type=float, default=0.0,
                    help="Anderson diagonal shift: 0 = the config as named (indefinite, timing only), "
                         "9 = the conditioned twin (SPD)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-target-512", action="store_true",
                    help="skip the HPCG-512 sub-record (north-star target size, int64 row pointers)")
    ap.add_argument("--target-size", type=int, default=512,
                    help="N > 1: grid edge of the north-star sub-record (`target_<size>`: the partitioned solve on HPCG "
                         "<size>^3, both stream formats); 0 = none")
    ap.add_argument("--target-steps", type=int, default=10)
    ap.add_argument("--tune-placement", type=int, default=0,
                    help="setup: keep the fastest of K re-allocations of the matrix' streamed arrays "
                         "(bis_mat_tune_placement; 0 = off)")
    ap.add_argument("--no-sweeps", action="store_true", help="skip the `sweeps` legs (natural-order SpTRSV per workload)")
    ap.add_argument("--no-configs", action="store_true", help="skip the `configs` legs (BASELINE configs 2-5 through the host CLI)")
    ap.add_argument("--cpu-iters", type=int, default=0, help="0: sized for ~10-30 s")
    ap.add_argument("--traffic-json", default=None,
                    help="per-launch HBM bytes from the rocprofv3 PMC passes (default: profiles/spmv_traffic[_<size>].json)")
    ap.add_argument("--headline", default="crs", choices=["crs", "default"],
                    help="what the timed region streams: crs = the CRS value array (north_star's CRS SpMV; the default), "
                         "default = the library's default format for the matrix (compressed where it applies)")
    return ap.parse_args()


def device_state():
    """Clocks, package power and temperatures of GPU 0 as rocm-smi reports them right now (called right behind a timed
    region, never inside one): a record whose kernel ran slower than on another box says whether the device was
    throttling.  None where rocm-smi is not there or does not answer."""
    import subprocess
    try:
        txt = subprocess.run(["rocm-smi", "-d", "0", "--showclocks", "--showpower", "--showtemp"], capture_output=True, text=True, timeout=20).stdout
    except Exception:  # noqa: BLE001
        return None
    state = {}
    for line in txt.splitlines():
        if ":" not in line or "GPU[" not in line:
            continue
        key, _, val = line.split(":", 1)[1].rpartition(":")
        key, val = key.strip(), val.strip()
        for tag, name in (("sclk", "sclk"), ("mclk", "mclk"), ("fclk", "fclk"), ("socclk", "socclk"), ("Power", "power_W"),
                          ("junction", "temp_junction_C"), ("memory", "temp_memory_C"), ("edge", "temp_edge_C")):
            if tag in key and name not in state:
                state[name] = val
    return state or None


def host_topology():
    """Sockets / physical cores / logical CPUs of this box and the CPUs this process may run on."""
    topo = {"logical_cpus": os.cpu_count()}
    topo["cpus_allowed"] = CPUS_ALLOWED
    try:
        phys, socks, model = set(), set(), None
        pid = cid = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                pid = line.split(":")[1].strip()
                socks.add(pid)
            elif line.startswith("core id"):
                cid = line.split(":")[1].strip()
                phys.add((pid, cid))
            elif line.startswith("model name") and model is None:
                model = line.split(":")[1].strip()
        topo.update(sockets=len(socks) or None, physical_cores=len(phys) or None, cpu_model=model)
    except Exception:
        pass
    return topo


def measured_stream(ctx, N):
    """Streaming ceilings of THIS box with the library's own kernels over N-vectors, wall-clocked over 100 queued launches:
    triad (sum_vectors, 24 B per element: two reads, one write), copy (16 B) and read (the sum of squares, 8 B in, nothing
    out).  The operands ROTATE through eight vectors (1 GB at N = 16.8 M), so that no launch finds its inputs in the 256 MB
    Infinity Cache: with three fixed vectors the same kernels measure 5-15 % more, which is cache, not HBM."""
    import ctypes as C_
    v = [ctx.alloc(N) for _ in range(8)]
    for i, w in enumerate(v):
        ctx.init_vector(w, 1.0 + i)
    scratch = ctx.alloc(8)
    k = [0]

    def triad():
        ctx.sum_vectors(v[k[0] % 8], v[(k[0] + 3) % 8], v[(k[0] + 6) % 8], 0.5); k[0] += 1

    def copy():
        ctx.copy_vector(v[k[0] % 8], v[(k[0] + 3) % 8]); k[0] += 1

    def read():
        ctx.check(ctx.lib.bis_sumsq_dev(ctx.h, C_.c_void_p(v[k[0] % 8].ptr), C_.c_int64(N), C_.c_void_p(scratch.ptr))); k[0] += 1

    out = {}
    for name, fn, nbytes in (("triad", triad, 24 * N), ("copy", copy, 16 * N), ("read", read, 8 * N)):
        for _ in range(8):
            fn()
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(100):
            fn()
        ctx.sync()
        out[name] = 100 * nbytes / (time.perf_counter() - t0) / 1e9
    # (`read` is two launches per pass -- the partial sums and their 1-workgroup finish: at 21 us per pass the boundary
    # between them is ~10 % of the wall time, so this figure understates the kernel's own rate, 6.3-6.4 TB/s under rocprofv3)
    for w in v:
        w.free()
    scratch.free()
    return out


KERNEL_OF_FORM = ("spmv_rowblock_kernel", "spmv_rowblock_vd_kernel", "spmv_rowmajor_vd_kernel", "spmv_rowmajor_vd_kernel",
                  "spmv_sellwin_kernel", "spmv_sellwin_kernel", "spmv_win8_kernel",
                  "spmv_rowblock_kernel (one pass per column slab)")


def stream_format(A):
    """What the SpMV streams for this matrix (bis_mat_spmv_stream_info / bis_mat_spmv_streamed_bytes)."""
    col_b, val_b, n_dict, form = A.spmv_stream_info()
    return {"col_bytes": col_b, "val_bytes": val_b, "dictionary_values": n_dict, "form": form,
            "kernel": "spmv_sellmask_kernel" if form >= 4 and col_b == 0 else KERNEL_OF_FORM[form],  # (form 4 without per-non-zero codes: the row-mask kernel)
            "streamed_bytes_per_nnz": col_b + val_b,
            "streamed_bytes_per_launch": A.spmv_streamed_bytes()}


def traffic_file(size):
    """profiles/spmv_traffic_<size>.json where a PMC pass at that size was committed, else the HPCG-256 file."""
    f = os.path.join(ROOT, "profiles", f"spmv_traffic_{size}.json")
    return f if os.path.exists(f) else os.path.join(ROOT, "profiles", "spmv_traffic.json")


def load_traffic(path, size, kernel, streamed):
    """Per-launch HBM bytes of `kernel` from the rocprofv3 PMC passes of this command (tools/pmc_traffic.py writes
    the file from separate FETCH_SIZE / WRITE_SIZE runs); None when no pass covers this kernel at this size -- or
    when the measured bytes are not those of the stream format running now (a kernel name covers several formats,
    and a regression that re-reads data must not hide behind an old pass: a pass is accepted between 0.95x and
    1.25x the format's own bytes)."""
    try:
        tj = json.load(open(path))
    except Exception:
        return None
    if tj.get("size") != size:
        return None
    if "kernels" in tj:
        k = tj["kernels"].get(kernel)
        t = k.get("hbm_bytes_per_launch") if k else None
    else:
        t = tj.get("hbm_bytes_per_launch") if tj.get("kernel", "spmv_rowblock_kernel") == kernel else None
    if t is None or not 0.95 * streamed <= t <= 1.25 * streamed:
        return None
    return t


def spmv_roofline(A, avg_s, launches, traffic_path, size):
    """The roofline record of the SpMV launches of one leg.

    CRS value stream (the kernel reads the CRS `val` array: SURVEY 8d's "CRS SpMV"): `achieved` = the ALGORITHMIC
    bytes of kernels.hpp:22-42 per launch (12 nnz + 20 N; + 4 N with 64-bit row pointers) / the average launch
    time, `traffic` = the PMC-measured HBM bytes per launch (None without a pass for this kernel and size), and
    `moved_GBs` / `moved_frac` the same time priced on those (or, without a pass, on the format's own bytes -- the
    packed column stream is 2 bytes per non-zero, so the kernel moves less than the algorithmic count).
    Compressed stream formats (value dictionary / sliced ELL: a lossless re-encoding, fewer bytes): `achieved` prices
    the bytes the kernel MOVES (<= 1 by construction); the CRS byte count over the same time is kept as
    crs_equivalent_GBs -- a speed-up figure, not a bandwidth."""
    fmt = stream_format(A)
    N, nnz = A.n_rows, A.nnz
    crs_bytes = 12 * nnz + (24 if A.rp_width == 8 else 20) * N
    traffic = load_traffic(traffic_path, size, fmt["kernel"], fmt["streamed_bytes_per_launch"])
    moved = traffic if traffic else fmt["streamed_bytes_per_launch"]
    crs_stream = fmt["val_bytes"] == 8
    priced = crs_bytes if crs_stream else moved
    achieved = priced / avg_s / 1e9
    rec = {"bound": "hbm", "kernel": fmt["kernel"], "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
           "priced_on": ("algorithmic bytes of the CRS SpMV (12 nnz + 20 N, SURVEY 8d)" if crs_stream else
                         "PMC HBM traffic per launch" if traffic else
                         "bytes of the stream format (no PMC pass for this kernel/size)"),
           "traffic_source": "PMC HBM traffic per launch (profiles/)" if traffic else None,
           "avg_launch_ms": avg_s * 1e3, "launches": launches,
           "algorithmic_bytes_per_launch": crs_bytes,
           "moved_bytes_per_launch": moved, "moved_GBs": moved / avg_s / 1e9,
           "moved_frac": moved / avg_s / 1e9 / HBM_PEAK_GBS,
           "streamed_bytes_per_launch": fmt["streamed_bytes_per_launch"],
           "spmv_gflops": 2.0 * nnz / avg_s / 1e9, "spmv_stream": fmt}
    if fmt["form"] == 6:  # the library's own placement search for this stream (untimed setup, include/bis_hip.h bis_mat_win8_tuning)
        t, first_ms, kept_ms = A.win8_tuning()
        rec["placement_search"] = {"reallocations_tried": t, "kernel_ms_first_allocation": first_ms, "kernel_ms_kept": kept_ms}
    if not crs_stream:
        rec["crs_equivalent_GBs"] = crs_bytes / avg_s / 1e9
        rec["crs_equivalent_frac_of_peak"] = crs_bytes / avg_s / 1e9 / HBM_PEAK_GBS
    return rec


def crs_value_stream_record(leg):
    """The CRS-value leg once more under the key round 3's records used (`crs_value_stream`), with its roofline priced on the
    PMC-measured HBM traffic where a counter pass covers this kernel and size -- the same launches, the same time; the leg's
    own `roofline` stays priced on SURVEY 8d's algorithmic bytes."""
    r = leg["roofline"]
    pmc = r["traffic"] is not None
    return {"cg_iterations_per_s": leg["cg_iterations_per_s"], "ms_per_step": leg["ms_per_step"],
            "spmv_avg_launch_ms": leg["spmv_avg_launch_ms"],
            "roofline": {"bound": "hbm", "kernel": r["kernel"], "achieved": r["moved_GBs"], "peak": r["peak"], "unit": "GB/s",
                         "frac": r["moved_frac"], "traffic": r["traffic"],
                         "priced_on": "PMC HBM traffic per launch" if pmc else "bytes of the stream format (no PMC pass for this kernel/size)",
                         "algorithmic_frac": r["frac"], "avg_launch_ms": r["avg_launch_ms"], "launches": r["launches"]}}


def cg_leg(ctx, A, b, x, D, steps, warmup, traffic_path, size, valdict=None, win8=None):
    """`steps` timed CG iterations after `warmup` on (A, b, x = 0.1); valdict=0: with the value dictionary off (the
    kernel streams the 8-byte CRS values: SURVEY 8d's 'CRS SpMV'), None: the library's default stream format; win8=0: the
    row-block kernel on the CRS arrays in place instead of the window + sliced-ELL re-layout of the values."""
    ctx.set_option("spmv_valdict", -1 if valdict is None else valdict)
    ctx.set_option("spmv_win8", -1 if win8 is None else win8)
    try:
        ctx.init_vector(x, 0.1)
        cg = ctx.cg(A, b, x, D)
        r0 = cg.init(0.0)
        cg.iterate(warmup)
        ctx.sync()
        ctx.profile(True)
        t0 = time.perf_counter()
        cg.iterate(steps)
        ctx.sync()
        secs = time.perf_counter() - t0
        ctx.profile(False)
        launches, spmv_ms = ctx.profile_read()
        iters, conv, hist = cg.status(hist_cap=warmup + steps + 1)
        assert iters == warmup + steps
        cg.free()
        avg_s = spmv_ms * 1e-3 / max(launches, 1)
        roof = spmv_roofline(A, avg_s, launches, traffic_path, size)
    finally:
        ctx.set_option("spmv_valdict", -1)
        ctx.set_option("spmv_win8", -1)
    return {"steps": steps, "warmup": warmup, "cg_iterations_per_s": steps / secs, "ms_per_step": 1e3 * secs / steps,
            "spmv_avg_launch_ms": avg_s * 1e3, "spmv_frac_of_peak": roof["frac"], "roofline": roof,
            "residual_r0": r0, "residual_history": [float(h) for h in hist]}


IN_PLACE_NOTE = ("the same loop with the row-block kernel reading the CRS val array and the packed column codes IN PLACE (x gathered through "
                 "L1 / L2): round 4's headline kernel; bit-identical y")
W8_NOTE = ("8-byte CRS values + 2-byte window slots streamed in sliced-ELL order (a lossless re-layout of the CRS arrays built on the "
           "device), x window of each 1024-row block in LDS by LDS-DMA")


def target_512(ctx, steps=10, warmup=3):
    """North-star target size: CG on HPCG 512^3 (3.6e9 nnz, int64 row pointers), same fused schedule: the literal
    'CRS SpMV inside the CG loop on 512^3' (8-byte CRS values streamed), then (`compressed_stream`) the library's
    default stream format.  The reference's int CRS cannot hold this matrix, so there is no CPU leg
    (tests/test_gpu_kernels.py gates it through closed forms and the fused-vs-unfused history)."""
    n1 = 512
    N = n1 ** 3
    A = ctx.gen_hpcg(n1)
    b, x = ctx.alloc(N), ctx.alloc(N)
    ctx.init_vector(b, 1.0)
    rec = {"workload": "HPCG 512^3 27-point, -cg, b=1 x0=0.1, fused device schedule, int64 row_ptr, CRS values streamed",
           "rows": N, "nnz": A.nnz, "rp_width": A.rp_width}
    leg = cg_leg(ctx, A, b, x, None, steps, warmup, traffic_file(n1), n1, valdict=0)
    h1 = leg.pop("residual_history")
    rec.update(leg)
    rec["residual_last"] = h1[-1]
    rec["crs_value_stream"] = crs_value_stream_record(leg)
    inp = cg_leg(ctx, A, b, x, None, steps, warmup, traffic_file(n1), n1, valdict=0, win8=0)
    h3 = inp.pop("residual_history")
    inp["note"] = IN_PLACE_NOTE
    inp["history_max_dev_over_r0_vs_headline"] = max(abs(a - c) for a, c in zip(h1, h3)) / h1[0]
    rec["crs_arrays_in_place"] = inp
    cmp_ = cg_leg(ctx, A, b, x, None, steps, warmup, traffic_file(n1), n1)
    h2 = cmp_.pop("residual_history")
    cmp_["note"] = "the library's default stream format for this matrix (lossless re-encoding, bit-identical y)"
    cmp_["history_max_dev_over_r0_vs_crs_leg"] = max(abs(a - c) for a, c in zip(h1, h2)) / h1[0]
    rec["compressed_stream"] = cmp_
    A.free(); b.free(); x.free()
    return rec


def unstructured_spmv(ctx, launches=20, rcm=False, asis=False):
    """BASELINE config 5's SpMV, HIP-event timed: the stand-in with a grid (fem:80,80,80 -- the size the PMC passes of
    tools/spmv_pmc.sh profile; rows of 18-81 entries, more than 256 distinct values: no dictionary), or (rcm=True) the
    unstructured input as a real mesh is multiplied: unstr:80,80,80 RCM-ordered."""
    import numpy as np
    if rcm:
        A0 = ctx.gen_unstr(80, 80, 80)
        A = ctx.permute(A0, ctx.bfs_order(A0, rcm=True))
        A0.free()
    elif asis:  # a mesh numbered at random: no locality at all (the column-slab form, bis_spmv_slab.hip)
        A = ctx.gen_unstr(80, 80, 80)
    else:
        A = ctx.gen_fem(80, 80, 80)
    N = A.n_rows
    x, y = ctx.upload(np.random.default_rng(12345).uniform(-1, 1, N)), ctx.alloc(N)
    for _ in range(3):
        ctx.spmv(A, x, y)
    ctx.sync()
    ctx.profile(True)
    for _ in range(launches):
        ctx.spmv(A, x, y)
    ctx.sync()
    ctx.profile(False)
    n, ms = ctx.profile_read()
    rec = {"workload": ("unstr:80,80,80 RCM-ordered (config 5 as named, the order a mesh is solved in), y = A x" if rcm else
                        "unstr:80,80,80 as generated (config 5 as named: rows numbered at random, no locality), y = A x" if asis else
                        "fem:80,80,80 (stand-in for Flan_1565), y = A x"), "rows": N, "nnz": A.nnz,
           "roofline": spmv_roofline(A, ms * 1e-3 / max(n, 1), n, os.path.join(ROOT, "profiles", "spmv_traffic_unstr_rcm.json" if rcm else "spmv_traffic_unstr_asis.json" if asis else "spmv_traffic_fem.json"), 80)}
    if asis:
        k, one_ms, slab_ms = A.colslab_info()
        rec["column_slabs"] = {"slabs": k, "trial_one_pass_ms": one_ms, "trial_slab_passes_ms": slab_ms,
                               "note": "x (12.3 MB) does not fit an XCD's L2: K passes over column ranges whose x slices do, each continuing the rows' sums (bit-identical y); kept because the build-time trial measured them faster"}
    A.free(); x.free(); y.free()
    return rec


def cpu_baseline(size, precond, iters, threads=None, seconds=15.0, first_touch=False):
    """The reference's CG on the host cores of this box.

    kind "reference": oracle/_ref (the reference's own ConjugateGradientSolver,
    compiled from its sources by oracle/Makefile and shipped prebuilt), timed by
    the reference's own timer tree (iterate + sample).  Falls back to kind
    "port" (the oracle's OpenMP restatement) if the prebuilt reference is absent.
    """
    import numpy as np

    from oracle import pyoracle
    if threads is None:
        threads = int(os.environ.get("BIS_CPU_THREADS", "16"))  # the box's CPU share for one GPU
    pyoracle.set_omp_threads(threads)
    orc = pyoracle.Oracle()
    t0 = time.time()
    A = orc.gen_hpcg(size)
    gen_s = time.time() - t0
    D = np.full(A.n_rows, 26.0) if precond == "j" else None
    _, s1 = orc.cg_run(A, 1, D)
    if iters <= 0:  # size the sample for ~15 s of CPU work
        iters = int(max(3, min(400, seconds / max(s1, 1e-3))))
    if pyoracle.Ref.available() and A.nnz < 2 ** 31 - 1 and os.environ.get("BIS_CPU_KIND") != "port":
        ref = pyoracle.Ref()
        ref.set_first_touch(first_touch)
        try:
            r = ref.solve(A, "cg", "j" if precond == "j" else "none", max_iters=iters, tol=1e-300)
        finally:
            ref.set_first_touch(False)
        secs = r["iterate_s"] + r["sample_s"]
        n_it = r["iters"]
        return dict(value=n_it / secs, unit="CG iterations/s", cores=threads, kind="reference",
                    topology=host_topology(), omp_proc_bind=os.environ.get("OMP_PROC_BIND"),
                    omp_places=os.environ.get("OMP_PLACES"),
                    build=pyoracle.ref_build_info(),
                    matrix_first_touch=("MatrixCRS::operator= (OpenMP-parallel copy, sparse_matrix.hpp:92-128)" if first_touch
                                        else "calling thread (memcpy)"),
                    sample=f"HPCG {size}^3 ({A.nnz} nnz), {n_it} CG iterations of the reference's own "
                           f"ConjugateGradientSolver (oracle/_ref, g++ {pyoracle.ref_build_info().get('flags', '?')}, "
                           f"{threads} OpenMP threads), iterate+sample time from its timer tree; "
                           f"SpMV share {r['spmv_s'] / secs:.2f}",
                    ms_per_step=1e3 * secs / n_it), r["hist"]
    hist, secs = orc.cg_run(A, iters, D)
    return dict(value=iters / secs, unit="CG iterations/s", cores=threads, kind="port",
                topology=host_topology(), omp_proc_bind=os.environ.get("OMP_PROC_BIND"),
                omp_places=os.environ.get("OMP_PLACES"),
                sample=f"HPCG {size}^3 ({A.nnz} nnz), {iters} CG iterations of the oracle's OpenMP "
                       f"port (oracle/bis_oracle.c orc_cg_run), {threads} threads, "
                       f"matrix generated on host in {gen_s:.1f} s",
                ms_per_step=1e3 * secs / iters), hist


SWEEP_CASES = (
    # (key, workload text, generator, reorder)
    ("hpcg256", "HPCG 256^3 27-point: strict triangles of A, D = diag(A) (-cg -p sgs / gs sweeps)", lambda c: c.gen_hpcg(256), None),
    ("anderson256", "Anderson 256^3 7-point, shift 9 (config 4's GS sweep; same pattern as configs 2-3)", lambda c: c.gen_anderson(256, shift=9.0), None),
    ("fem80x80x81", "fem:80,80,81 (config-5 stand-in WITH a grid hint, 1.56 M rows)", lambda c: c.gen_fem(80, 80, 81), None),
    ("unstr80_asis", "unstr:80,80,80 as generated (config 5 as named: no grid, no locality, 135 wide levels)", lambda c: c.gen_unstr(80, 80, 80), None),
    ("unstr80_rcm", "unstr:80,80,80 RCM-ordered (config 5 as a real mesh is solved: ~7 thousand levels)", lambda c: c.gen_unstr(80, 80, 80), "rcm"),
)


def load_sweep_traffic(key, direction, algorithmic):
    """PMC HBM bytes per sweep from profiles/trsv_traffic.json (tools/trsv_traffic.py: separate FETCH_SIZE / WRITE_SIZE
    passes of the sweep kernel of this workload, sentinel fill included); accepted between 0.9x and 12x the algorithmic bytes
    (polls re-read operands; an ordering without locality fetches a line per 8-byte operand: unstr as generated moves 8x)."""
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "trsv_traffic.json")))
        t = tj["sweeps"][key][direction]["hbm_bytes_per_sweep"]
    except Exception:
        return None
    return t if t and 0.9 * algorithmic <= t <= 12.0 * algorithmic else None


def sweep_legs(ctx, sweeps=10, warm=3, only=None):
    """Forward and backward natural-order sweeps, x = (D + T)^-1 b, on the strict triangles of each workload: `sweeps` timed
    bis_sptrsv / bis_bsptrsv calls after `warm` (the first builds the plan), HIP events around every call on the library's
    stream (bis_profile_read_sweeps).  roofline: achieved = (12 nnz_T + 28 N) / average sweep time."""
    import numpy as np
    out = {}
    for key, text, gen, reorder in SWEEP_CASES:
        if only and key not in only:
            continue
        t_setup = time.perf_counter()
        A = gen(ctx)
        if reorder:
            perm = ctx.bfs_order(A, rcm=True)
            B = ctx.permute(A, perm)
            A.free()
            A = B
        N = A.n_rows
        Ls, Us, D, Dinv = ctx.split_strict(A)
        A.free(); Dinv.free()
        b = ctx.upload(np.random.default_rng(21).uniform(-1, 1, N))
        x = ctx.alloc(N)
        rec = {"workload": text, "rows": N}
        for direction, T, solve in (("forward", Ls, ctx.sptrsv), ("backward", Us, ctx.bsptrsv)):
            for _ in range(warm):
                solve(T, x, D, b)
            ctx.sync()
            ctx.profile(True)
            for _ in range(sweeps):
                solve(T, x, D, b)
            ctx.sync()
            ctx.profile(False)
            n, ms = ctx.profile_read_sweeps()
            ctx.profile_read()  # (drop the SpMV events a per-level path may have recorded)
            avg_s = ms * 1e-3 / max(n, 1)
            alg = 12 * T.nnz + 28 * N
            traffic = load_sweep_traffic(key, direction, alg)
            ach = alg / avg_s / 1e9
            rec[direction] = {"nnz_T": T.nnz, "avg_sweep_ms": avg_s * 1e3, "sweeps": n,
                              "roofline": {"bound": "hbm", "kernel": T.sweep_kernel(direction == "backward"), "achieved": ach,
                                           "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                                           "algorithmic_bytes_per_sweep": alg,
                                           "moved_frac": (traffic / avg_s / 1e9 / HBM_PEAK_GBS) if traffic else None,
                                           "note": "latency-bound: dependency levels x one wave's row latency (DESIGN.md section 4), not bandwidth"},
                              "sweep_gflops": (2.0 * T.nnz + 2.0 * N) / avg_s / 1e9}
        rec["setup_s"] = time.perf_counter() - t_setup - sum(rec[d]["avg_sweep_ms"] * 1e-3 * (sweeps + warm) for d in ("forward", "backward"))
        for v in (Ls, Us):
            v.free()
        for v in (D, b, x):
            v.free()
        out[key] = rec
    return out


CLI = os.path.join(ROOT, "basic_iterative_solvers_amd", "host", "basic_iterative_solvers")
CONFIG_RUNS = (
    ("config2", "Anderson 256^3 raw (indefinite: runs to MAX_ITERS), -cg", ["anderson:256", "-cg"]),
    ("config3_1gpu", "Anderson 256^3 shift 9, -cg -p j on ONE GPU (config 3 is the 8-GPU partition of this)", ["anderson:256,shift=9", "-cg", "-p", "j"]),
    ("config4", "Anderson 256^3 shift 9, -gm -p gs (GMRES(10) + natural-order GS sweep)", ["anderson:256,shift=9", "-gm", "-p", "gs"]),
    ("config5_standin", "fem:80,80,81 (grid-hinted stand-in), -bi -p ilu0", ["fem:80,80,81", "-bi", "-p", "ilu0"]),
    ("config5_unstr_asis", "unstr:80,80,80 as generated, -bi -p ilu0", ["unstr:80,80,80", "-bi", "-p", "ilu0"]),
    ("config5_unstr_rcm", "unstr:80,80,80 -perm rcm, -bi -p ilu0", ["unstr:80,80,80", "-bi", "-p", "ilu0", "-perm", "rcm"]),
)
_TIMER_KEYS = (("total_s", "Total elapsed time:"), ("preprocessing_s", "| Preprocessing time:"), ("factor_s", "| | Factor time:"),
               ("solve_s", "| Solve time:"), ("iterate_s", "| | Iterate time:"), ("spmv_s", "| | | SpMV time:"), ("precond_s", "| | | Precond. time:"),
               ("dot_s", "Dot time:"), ("sum_s", "Sum time:"), ("orthog_s", "| | | Orthog. time:"), ("sample_s", "| | Sample time:"))


def _run_cli(args, sync_timers, timeout=180):
    """One run of the host CLI (the C++ host layer over the C ABI, a child process); returns the parsed summary or an error."""
    import re
    import subprocess
    env = dict(os.environ)
    env["BIS_TIMERS_SYNC"] = "1" if sync_timers else "0"
    env.setdefault("OMP_NUM_THREADS", "1")
    try:
        r = subprocess.run([CLI] + list(args), capture_output=True, text=True, timeout=timeout, env=env)
    except Exception as e:  # missing binary, timeout
        return {"error": repr(e)[:200]}
    if r.returncode != 0:
        return {"error": (r.stderr or r.stdout)[-300:]}
    m = re.search(r"(converged in: |did not converge after )(\d+) iterations", r.stdout)
    out = {"iterations": int(m.group(2)) if m else None, "converged": bool(m and m.group(1).startswith("converged"))}
    last = {}
    for line in r.stdout.splitlines():
        for k, label in _TIMER_KEYS:
            if label in line:
                try:
                    last[k] = float(line.split(label)[1].strip().split("[")[0])
                except Exception:
                    pass
        if line.startswith("Device library options in effect:"):
            out["options"] = line.split(":", 1)[1].strip()
    out.update(last)  # (the timer tree is printed at the milestones too: the LAST print is the whole run)
    res = re.findall(r"\|\|A\*x_(\d+) - b\|\|_2 = (\S+)", r.stdout)
    if res:
        out["residual_first"], out["residual_last"] = float(res[0][1]), float(res[-1][1])
    return out


def config_legs(only=None):
    """BASELINE configs 2, 4, 5 (and config 3's one-GPU form) through the library's own solver schedules (host CLI): a run
    with asynchronous launches for the times that count (iterations, iterate time, preprocessing), and a second run with
    the timer tree draining the stream per call (the reference's TIME semantics) for the SpMV / preconditioner / BLAS-1 split."""
    out = {}
    for key, text, args in CONFIG_RUNS:
        if only and key not in only:
            continue
        a = _run_cli(args, sync_timers=False)
        rec = {"workload": text, "command": " ".join(["basic_iterative_solvers"] + args)}
        if "error" in a:
            rec["error"] = a["error"]
            out[key] = rec
            continue
        # solve_s = the reference's "Solve time" (iterate + sample + exchange, solver_harness.hpp:7-61): with asynchronous launches
        # the device-scalar schedules only drain the stream where the residual is sampled, so the split below it is not
        # meaningful in this run -- the second run gives it
        rec.update({k: a.get(k) for k in ("iterations", "converged", "solve_s", "preprocessing_s", "factor_s", "total_s",
                                          "residual_first", "residual_last", "options")})
        if a.get("iterations") and a.get("solve_s"):
            rec["ms_per_iteration"] = 1e3 * a["solve_s"] / a["iterations"]
        b = _run_cli(args, sync_timers=True)
        if "error" not in b and b.get("iterate_s"):
            blas1 = sum(b.get(k) or 0.0 for k in ("dot_s", "sum_s", "orthog_s"))
            rec["split_with_synchronous_timers"] = {
                "iterate_s": b["iterate_s"], "spmv_s": b.get("spmv_s"), "precond_s": b.get("precond_s"), "blas1_s": blas1,
                "spmv_share": (b.get("spmv_s") or 0.0) / b["iterate_s"], "precond_share": (b.get("precond_s") or 0.0) / b["iterate_s"],
                "blas1_share": blas1 / b["iterate_s"]}
        out[key] = rec
    return out


def cpu_sptrsv_leg(size=128, seconds=6.0):
    """The reference's serial native_sptrsv / native_bsptrsv (kernels.hpp:54-107, from oracle/_ref; the oracle's restatement
    where the prebuilt reference is absent) on HPCG-`size`: the CPU figure beside `sweeps` (one thread: the loop is serial)."""
    import numpy as np

    from oracle import pyoracle
    orc = pyoracle.Oracle()
    A = orc.gen_hpcg(size)
    L, Ls, U, Us = orc.split_LU(A)
    D, _, _ = orc.peel_diag(L)
    del L, U
    b = np.random.default_rng(21).uniform(-1, 1, A.n_rows)
    kind = "reference" if pyoracle.Ref.available() else "port"
    eng = pyoracle.Ref() if kind == "reference" else orc
    rec = {"kind": kind, "cores": 1, "unit": "ms per sweep", "sample": f"HPCG {size}^3 strict triangles ({Ls.nnz} non-zeros each), "
           "the reference's serial loop, best of the sweeps that fit a few seconds"}
    for direction, T, backward in (("forward", Ls, False), ("backward", Us, True)):
        best, t_end, n = None, time.time() + seconds / 2, 0
        while n < 2 or (time.time() < t_end and n < 20):
            t0 = time.perf_counter()
            eng.sptrsv(T, D, b, backward=backward)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
            n += 1
        alg = 12 * T.nnz + 28 * A.n_rows
        rec[direction] = {"ms_per_sweep": best * 1e3, "sweeps": n, "GBs_on_algorithmic_bytes": alg / best / 1e9}
    rec["value"] = rec["forward"]["ms_per_sweep"]
    return rec


def cpu_topology(threads_list=(16, 64)):
    """What explains the CPU leg: the cgroup's CPU quota, the NUMA nodes, where libgomp put the threads, and a host triad
    (first touch by the same static schedule) at the thread counts the CG legs use."""
    import ctypes as C

    from oracle import pyoracle
    topo = host_topology()
    try:
        topo["cgroup_cpu_max"] = open("/sys/fs/cgroup/cpu.max").read().strip()
    except Exception:
        topo["cgroup_cpu_max"] = None
    try:
        q, per = topo["cgroup_cpu_max"].split()
        topo["cgroup_cpu_quota_cores"] = None if q == "max" else float(q) / float(per)
    except Exception:
        topo["cgroup_cpu_quota_cores"] = None
    try:
        topo["numa_nodes"] = len([d for d in os.listdir("/sys/devices/system/node") if d.startswith("node") and d[4:].isdigit()])
    except Exception:
        topo["numa_nodes"] = None
    orc = pyoracle.Oracle()
    orc.lib.orc_host_triad.restype = C.c_double
    triad = {}
    for t in threads_list:
        if topo.get("logical_cpus") and t > topo["logical_cpus"]:
            continue
        pyoracle.set_omp_threads(t)
        places = (C.c_int * 256)(*([-1] * 256))
        gbs = orc.lib.orc_host_triad(C.c_int64(1 << 27), C.c_int(3), places, C.c_int(256))
        cpus = [p for p in list(places)[:t] if p >= 0]
        triad[str(t)] = {"GBs": gbs, "distinct_cpus": len(set(cpus)), "cpu_min": min(cpus) if cpus else None, "cpu_max": max(cpus) if cpus else None}
    topo["host_triad"] = triad
    topo["omp_proc_bind"] = os.environ.get("OMP_PROC_BIND")
    topo["omp_places"] = os.environ.get("OMP_PLACES")
    return topo


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher around it: start `python -m torch.distributed.run --nnodes=1
    --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py <the same arguments>` as a child process group,
    pass its stdout (rank 0's ONE JSON line) through and return its exit code."""
    import socket

    from basic_iterative_solvers_amd.watchdog import supervise
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")  # (torchrun would set it, with a warning on stderr)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    # the ranks run as a child process group under a watchdog (basic_iterative_solvers_amd/watchdog.py): a rank that outlives
    # a phase limit ends the run with ONE diagnostic JSON line {"error", "phase", "rank", ...} and a non-zero status
    return supervise(cmd, n, env=env)


def main():
    args = parse()
    # the CPU leg's OpenMP binding, fixed before libgomp starts
    os.environ.setdefault("OMP_PROC_BIND", "close")
    os.environ.setdefault("OMP_PLACES", "cores")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # invoked plainly (`python bench.py --gpus N ...`): become the launcher.  Nothing has touched the GPU yet (no
        # torch import, no HIP call), and the ranks run as CHILD processes -- never an exec of a process that has a
        # device open.  The child's single JSON line and its return code are relayed.
        raise SystemExit(launch_ranks(args.gpus))
    import torch  # plumbing: device sync + torch.distributed launcher contract

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} rank(s): "
                         "run `python bench.py --gpus N` (it launches its own ranks) or "
                         "`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`")
    if os.environ.get("BIS_BENCH_REHEARSE") == "1":
        local_rank = 0  # one-GPU rehearsal of the N > 1 path (tests/test_dist.py)
    torch.cuda.set_device(local_rank)

    from basic_iterative_solvers_amd import Context

    if world > 1 or os.environ.get("BIS_FORCE_DIST") == "1" or args.matrix != "hpcg":
        # the partitioned runner also serves N = 1 (a 1-rank communicator): Anderson configs, and
        # BIS_FORCE_DIST=1 to compare the distributed code path with the plain one on the same problem
        for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0"), ("MASTER_PORT", "29533")):
            os.environ.setdefault(k, v)
        from basic_iterative_solvers_amd.dist_bench import run_distributed
        return run_distributed(args, rank, world, local_rank)

    ctx = Context(local_rank)
    n1 = args.size
    N = n1 ** 3
    A = ctx.gen_hpcg(n1)
    nnz = A.nnz
    tuned = None
    if args.tune_placement > 0:  # untimed setup, a library feature (include/bis_hip.h)
        f_ms, b_ms = ctx.tune_placement(A, args.tune_placement)
        tuned = {"trials": args.tune_placement, "spmv_ms_first_allocation": f_ms, "spmv_ms_kept": b_ms}
    b, x = ctx.alloc(N), ctx.alloc(N)
    ctx.init_vector(b, 1.0)
    ctx.init_vector(x, 0.1)
    D = None
    if args.precond == "j":
        D = ctx.alloc(N)
        ctx.init_vector(D, 26.0)
    # The timed region runs what north_star names: the CRS SpMV with the CRS value array streamed.  (The library's
    # default for a matrix with <= 256 distinct values is a compressed re-encoding: the `compressed_stream` leg below.)
    headline_valdict = -1 if args.headline == "default" else 0
    ctx.set_option("spmv_valdict", headline_valdict)
    cg = ctx.cg(A, b, x, D)
    r0 = cg.init(0.0)  # tol 0: the timed iterations all execute (no early stop)
    cg.iterate(args.warmup)
    ctx.sync()
    torch.cuda.synchronize()

    ctx.profile(True)
    t0 = time.perf_counter()
    cg.iterate(args.steps)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    ctx.profile(False)
    launches, spmv_ms = ctx.profile_read()
    state_after_timed = device_state()
    iters, conv, hist = cg.status(hist_cap=args.warmup + args.steps + 1)
    if iters != args.warmup + args.steps:
        raise SystemExit(f"timed region invalid: {iters} iterations executed, "
                         f"expected {args.warmup + args.steps}")
    secs = t1 - t0
    its = args.steps / secs

    spmv_avg_s = spmv_ms * 1e-3 / max(launches, 1)
    roof = spmv_roofline(A, spmv_avg_s, launches, args.traffic_json or traffic_file(n1), n1)
    stream = measured_stream(ctx, N)
    # the streaming ceiling measured on this box: the library's own triad over N-vectors
    roof["measured_stream_GBs"] = stream["triad"]
    roof["measured_copy_GBs"] = stream["copy"]
    # the read-only stream (the sum of squares of an N-vector: 8 N bytes in, nothing out): the yardstick for a kernel that,
    # like the SpMV, reads almost everything it moves -- the triad (two reads, one write) understates what reads can reach
    roof["measured_read_GBs"] = stream["read"]
    roof["moved_frac_of_measured"] = roof["moved_GBs"] / stream["triad"]
    roof["moved_frac_of_measured_read"] = roof["moved_GBs"] / stream["read"]
    roof["algorithmic_frac"] = roof["frac"]  # (`frac` prices SURVEY 8d's algorithmic bytes; `moved_frac` the PMC traffic)
    vec_bytes = (80 if args.precond == "j" else 64) * N  # pass B 24 N (+16 N Jacobi), pass C 40 N
    out = {
        "metric": "CG iterations/sec + SpMV GFLOP/s (% HBM roofline), HPCG 256^3 at 1/2/4/8 GPUs",
        "value": its, "unit": "CG iterations/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * secs / args.steps,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"HPCG {n1}^3 27-point, -cg" +
                               (" -p j" if args.precond == "j" else "") +
                               ", b=1 x0=0.1, fused device schedule, " +
                               ("CRS values streamed (8 B per non-zero" + ("; " + W8_NOTE if roof["spmv_stream"]["form"] == 6 else "") + ")"
                                if roof["spmv_stream"]["val_bytes"] == 8 else "default (compressed) stream format"),
                   "rows": N, "nnz": nnz, "partition": "1 GPU"},
        "spmv_gflops": 2.0 * nnz / spmv_avg_s / 1e9,
        # the fused schedule's algorithmic bytes per iteration (12 nnz + 20 N + the two vector passes) per second
        "cg_algorithmic_GBs": (12 * nnz + 20 * N + vec_bytes) * its / 1e9,
        "cg_frac_of_peak": (12 * nnz + 20 * N + vec_bytes) * its / 1e9 / HBM_PEAK_GBS,
        "residual_r0": r0, "residual_last": float(hist[-1]),
        "roofline": roof,
        "device_state_after_timed_region": state_after_timed,
    }
    cg.free()
    ctx.set_option("spmv_valdict", -1)
    import numpy as np
    if roof["spmv_stream"]["val_bytes"] == 8:  # the headline once more, priced on the PMC traffic (see crs_value_stream_record)
        out["crs_value_stream"] = crs_value_stream_record({"cg_iterations_per_s": its, "ms_per_step": 1e3 * secs / args.steps,
                                                           "spmv_avg_launch_ms": spmv_avg_s * 1e3, "roofline": roof})
    if args.headline != "default" and roof["spmv_stream"]["form"] == 6:
        leg = cg_leg(ctx, A, b, x, D, min(args.steps, 50), min(args.warmup, 5), args.traffic_json or traffic_file(n1), n1, valdict=0, win8=0)
        h3 = np.array(leg.pop("residual_history"))
        m3 = min(len(h3), len(hist))
        leg["history_max_dev_over_r0_vs_timed_run"] = float(np.max(np.abs(h3[:m3] - np.array(hist[:m3]))) / h3[0])
        leg["note"] = IN_PLACE_NOTE
        out["crs_arrays_in_place"] = leg
    if args.headline != "default":
        # the same loop on the same arrays with the library's default stream format for this matrix
        leg = cg_leg(ctx, A, b, x, D, min(args.steps, 50), min(args.warmup, 5), args.traffic_json or traffic_file(n1), n1)
        if leg["roofline"]["spmv_stream"]["val_bytes"] < 8:
            h2 = np.array(leg.pop("residual_history"))
            m2 = min(len(h2), len(hist))
            # same y bit for bit; the fused (Ap, p) is summed over different row blocks, a different fixed order
            leg["history_max_dev_over_r0_vs_timed_run"] = float(np.max(np.abs(h2[:m2] - np.array(hist[:m2]))) / h2[0])
            leg["note"] = ("the library's default for this matrix: a lossless re-encoding of the CRS arrays (value dictionary / "
                           "sliced ELL with the x window in LDS), bit-identical y; not a CRS-bandwidth figure")
            out["compressed_stream"] = leg
    if not args.no_cpu_baseline:
        topo = cpu_topology()
        cb, cpu_hist = cpu_baseline(n1, args.precond, args.cpu_iters)
        cb["topology"] = topo
        out["cpu_baseline"] = cb
        # parity against the CPU path on the same input over the CPU leg's WHOLE history: a fresh GPU run of as many
        # iterations as the reference made (the timed run above covers only warmup + steps of them)
        n_cmp = len(cpu_hist) - 1
        ctx.set_option("spmv_valdict", headline_valdict)
        ctx.init_vector(x, 0.1)
        cgp = ctx.cg(A, b, x, D)
        cgp.init(0.0)
        cgp.iterate(n_cmp)
        _, _, gh = cgp.status(hist_cap=n_cmp + 1)
        cgp.free()
        ctx.set_option("spmv_valdict", -1)
        gh = np.array(gh)
        m = min(len(cpu_hist), len(gh))
        out["parity_max_dr_over_r0"] = float(np.max(np.abs(cpu_hist[:m] - gh[:m])) / cpu_hist[0])
        out["parity_samples"] = int(m)
        phys = (cb.get("topology") or {}).get("physical_cores") or 0
        socks = (cb.get("topology") or {}).get("sockets") or 1
        per_socket = phys // max(socks, 1)
        quota = topo.get("cgroup_cpu_quota_cores")
        if quota is not None and quota < per_socket:
            # the container's CPU quota is below one socket: more threads than the quota are throttled, not faster -- the
            # per-GPU share above IS the reference's best foot on this box
            out["cpu_baseline_socket"] = {"skipped": f"cgroup cpu.max allows {quota:g} cores; a {per_socket}-thread leg would be throttled"}
        elif os.environ.get("BIS_CPU_SOCKET_LEG", "1") != "0":
            if per_socket > cb["cores"]:
                # a second sample on one full socket (fewer iterations), beside the per-GPU share of the host
                cb2, _ = cpu_baseline(n1, args.precond, 0, threads=per_socket, seconds=5.0)
                out["cpu_baseline_socket"] = cb2
            # ... and the reference's best foot: its matrix copy made by MatrixCRS::operator= (OpenMP-parallel: the rows are
            # first touched by the threads that multiply them) on that socket
            cb3, _ = cpu_baseline(n1, args.precond, 0, threads=max(per_socket, cb["cores"]), seconds=5.0, first_touch=True)
            out["cpu_baseline_first_touch"] = cb3
    if tuned:
        out["placement_tuning"] = tuned
    A.free(); b.free(); x.free()
    if D is not None:
        D.free()
    if n1 == 256 and not args.no_target_512:
        info = ctx.device_info()
        if info["hbm_bytes"] >= 200e9:
            out["target_512"] = target_512(ctx)
            out["target_512"]["device_state_after_leg"] = device_state()
            out["config5_spmv"] = unstructured_spmv(ctx)
            out["config5_spmv_rcm"] = unstructured_spmv(ctx, rcm=True)
            out["config5_spmv_asis"] = unstructured_spmv(ctx, asis=True)
    if n1 == 256 and not args.no_sweeps:
        out["sweeps"] = sweep_legs(ctx)
        if not args.no_cpu_baseline:
            out["cpu_baseline_sptrsv"] = cpu_sptrsv_leg()
    out["options"] = ctx.options()
    out["options"]["placement_tuning"] = ("bis_mat_tune_placement: %d trials" % args.tune_placement) if args.tune_placement > 0 else \
        "bis_mat_tune_placement off; the library's own build-time search on the win8 stream is on (roofline.placement_search; option spmv_win8_tune)"
    ctx.close()  # (the child processes below get the whole device)
    if n1 == 256 and not args.no_configs:
        out["configs"] = config_legs()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
