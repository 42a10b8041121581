#!/usr/bin/env python3
"""bench.py -- CG iterations/s (+ SpMV GFLOP/s and % of the HBM roofline) on
the HPCG 27-point operator, the metric BASELINE.json names.

    python bench.py --gpus N --steps K --warmup W [--size 256] [--precond none|j]

A "step" is one CG iteration (methods/cg.hpp:6-54 + the residual sample of
:162-166) on synthetic HPCG-`size` (size^3 rows, 27-point, b = 1, x0 = 0.1),
generated directly in HBM; inputs are resident before the timed region.
N > 1: one process per GPU (torch.distributed.run), the matrix 1-D
row-partitioned (z-slabs), strong scaling: the global problem is fixed.

Prints ONE JSON line (rank 0).  `roofline` prices the dominant kernel (CRS
SpMV) with HIP events recorded on the library's stream inside the timed
region; `cpu_baseline` times the oracle's OpenMP port of the same CG loop on
the host cores for a bounded number of iterations (N = 1, rank 0 only).
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
    ap.add_argument("--tune-placement", type=int, default=0,
                    help="setup: keep the fastest of K re-allocations of the matrix' streamed arrays "
                         "(bis_mat_tune_placement; 0 = off)")
    ap.add_argument("--cpu-iters", type=int, default=0, help="0: sized for ~10-30 s")
    ap.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", "spmv_traffic.json"),
                    help="per-launch HBM bytes from the rocprofv3 PMC passes, if collected")
    return ap.parse_args()


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
    """Streaming ceiling of THIS box: the library's own axpy-class kernel (sum_vectors, 24 B per
    element: two reads, one write) and copy (16 B) over N-vectors, wall-clocked over 100 queued launches."""
    a, b, c = ctx.alloc(N), ctx.alloc(N), ctx.alloc(N)
    ctx.init_vector(a, 1.0); ctx.init_vector(b, 2.0)
    out = {}
    for name, fn, nbytes in (("triad", lambda: ctx.sum_vectors(c, a, b, 0.5), 24 * N),
                             ("copy", lambda: ctx.copy_vector(c, a), 16 * N)):
        for _ in range(5):
            fn()
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(100):
            fn()
        ctx.sync()
        out[name] = 100 * nbytes / (time.perf_counter() - t0) / 1e9
    for v in (a, b, c):
        v.free()
    return out


def stream_format(A):
    """What the SpMV streams per non-zero for this matrix (bis_mat_spmv_stream_info)."""
    col_b, val_b, n_dict, form = A.spmv_stream_info()
    return {"col_bytes": col_b, "val_bytes": val_b, "dictionary_values": n_dict,
            "kernel": ("spmv_rowblock_kernel", "spmv_rowblock_vd_kernel", "spmv_rowmajor_vd_kernel")[form],
            "streamed_bytes_per_nnz": col_b + val_b}


def crs_value_leg(ctx, A, b, x, D, steps, warmup):
    """The same CG on the same arrays with the value dictionary switched off (the kernel streams the 8-byte CRS
    values): the comparable of round 1's number, measured after the timed region."""
    ctx.set_option("spmv_valdict", 0)
    try:
        ctx.init_vector(x, 0.1)
        cg = ctx.cg(A, b, x, D)
        cg.init(0.0)
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
    finally:
        ctx.set_option("spmv_valdict", -1)
    N = A.n_rows
    avg_s = spmv_ms * 1e-3 / max(launches, 1)
    spmv_bytes = 12 * A.nnz + 20 * N
    return {"note": "option spmv_valdict=0: 8-byte CRS values streamed (2-byte column codes as before)",
            "steps": steps, "warmup": warmup, "cg_iterations_per_s": steps / secs, "ms_per_step": 1e3 * secs / steps,
            "spmv_avg_launch_ms": avg_s * 1e3, "spmv_GBs": spmv_bytes / avg_s / 1e9,
            "spmv_frac_of_peak": spmv_bytes / avg_s / 1e9 / HBM_PEAK_GBS, "spmv_gflops": 2.0 * A.nnz / avg_s / 1e9,
            "residual_history": [float(h) for h in hist]}


def target_512(ctx, steps=10, warmup=3):
    """North-star target size: CG on HPCG 512^3 (3.6e9 nnz, int64 row pointers), same fused schedule;
    the reference's int CRS cannot hold this matrix, so there is no CPU leg (tests/test_gpu_kernels.py
    gates it through closed forms and the fused-vs-unfused history)."""
    n1 = 512
    N = n1 ** 3
    A = ctx.gen_hpcg(n1)
    b, x = ctx.alloc(N), ctx.alloc(N)
    ctx.init_vector(b, 1.0)
    ctx.init_vector(x, 0.1)
    cg = ctx.cg(A, b, x)
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
    spmv_bytes = 12 * A.nnz + 24 * N  # + 4 N for the int64 row pointers
    avg_s = spmv_ms * 1e-3 / max(launches, 1)
    rec = {"workload": "HPCG 512^3 27-point, -cg, b=1 x0=0.1, fused device schedule, int64 row_ptr",
           "rows": N, "nnz": A.nnz, "rp_width": A.rp_width, "steps": steps, "warmup": warmup,
           "cg_iterations_per_s": steps / secs, "ms_per_step": 1e3 * secs / steps,
           "spmv_avg_launch_ms": avg_s * 1e3, "spmv_launches": launches,
           "spmv_algorithmic_bytes": spmv_bytes, "spmv_GBs": spmv_bytes / avg_s / 1e9,
           "spmv_frac_of_peak": spmv_bytes / avg_s / 1e9 / HBM_PEAK_GBS,
           "spmv_gflops": 2.0 * A.nnz / avg_s / 1e9, "residual_r0": r0, "residual_last": float(hist[-1]),
           "spmv_stream": stream_format(A)}
    cg.free(); A.free(); b.free(); x.free()
    return rec


def cpu_baseline(size, precond, iters):
    """The reference's CG on the host cores of this box.

    kind "reference": oracle/_ref (the reference's own ConjugateGradientSolver,
    compiled from its sources by oracle/Makefile and shipped prebuilt), timed by
    the reference's own timer tree (iterate + sample).  Falls back to kind
    "port" (the oracle's OpenMP restatement) if the prebuilt reference is absent.
    """
    import numpy as np

    from oracle import pyoracle
    threads = int(os.environ.get("BIS_CPU_THREADS", "16"))  # the box's CPU share for one GPU
    pyoracle.set_omp_threads(threads)
    orc = pyoracle.Oracle()
    t0 = time.time()
    A = orc.gen_hpcg(size)
    gen_s = time.time() - t0
    D = np.full(A.n_rows, 26.0) if precond == "j" else None
    _, s1 = orc.cg_run(A, 1, D)
    if iters <= 0:  # size the sample for ~15 s of CPU work
        iters = int(max(3, min(400, 15.0 / max(s1, 1e-3))))
    if pyoracle.Ref.available() and A.nnz < 2 ** 31 - 1 and os.environ.get("BIS_CPU_KIND") != "port":
        ref = pyoracle.Ref()
        r = ref.solve(A, "cg", "j" if precond == "j" else "none", max_iters=iters, tol=1e-300)
        secs = r["iterate_s"] + r["sample_s"]
        n_it = r["iters"]
        return dict(value=n_it / secs, unit="CG iterations/s", cores=threads, kind="reference",
                    topology=host_topology(), omp_proc_bind=os.environ.get("OMP_PROC_BIND"),
                    omp_places=os.environ.get("OMP_PLACES"),
                    sample=f"HPCG {size}^3 ({A.nnz} nnz), {n_it} CG iterations of the reference's own "
                           f"ConjugateGradientSolver (oracle/_ref, g++ -O3 -march=native -fopenmp, "
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


def main():
    args = parse()
    # the CPU leg's OpenMP binding, fixed before libgomp starts
    os.environ.setdefault("OMP_PROC_BIND", "close")
    os.environ.setdefault("OMP_PLACES", "cores")
    import torch  # plumbing: device sync + torch.distributed launcher contract

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run "
                             "(one process per GPU)")
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
    iters, conv, hist = cg.status(hist_cap=args.warmup + args.steps + 1)
    if iters != args.warmup + args.steps:
        raise SystemExit(f"timed region invalid: {iters} iterations executed, "
                         f"expected {args.warmup + args.steps}")
    secs = t1 - t0
    its = args.steps / secs

    spmv_bytes = 12 * nnz + 20 * N
    spmv_avg_s = spmv_ms * 1e-3 / max(launches, 1)
    achieved = spmv_bytes / spmv_avg_s / 1e9
    traffic = None
    if os.path.exists(args.traffic_json):
        try:
            tj = json.load(open(args.traffic_json))
            if tj.get("size") == n1 and tj.get("kernel", "spmv_rowblock_kernel") == stream_format(A)["kernel"]:
                traffic = tj.get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    stream = measured_stream(ctx, N)
    fmt = stream_format(A)
    streamed_bytes = (fmt["col_bytes"] + fmt["val_bytes"]) * nnz + 20 * N
    fused_bytes = 12 * nnz + (100 if args.precond == "j" else 84) * N  # SpMV 20 N, pass B 24 N (+16 N Jacobi), pass C 40 N
    out = {
        "metric": "CG iterations/sec + SpMV GFLOP/s (% HBM roofline), HPCG 256^3 at 1/2/4/8 GPUs",
        "value": its, "unit": "CG iterations/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * secs / args.steps,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"HPCG {n1}^3 27-point, -cg" +
                               (" -p j" if args.precond == "j" else "") +
                               ", b=1 x0=0.1, fused device schedule", "rows": N, "nnz": nnz,
                   "partition": "1 GPU"},
        "spmv_gflops": 2.0 * nnz / spmv_avg_s / 1e9,
        "cg_effective_GBs": fused_bytes * its / 1e9,
        "residual_r0": r0, "residual_last": float(hist[-1]),
        "roofline": {"bound": "hbm", "kernel": fmt["kernel"], "achieved": achieved,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic, "algorithmic_bytes_per_launch": spmv_bytes,
                     "avg_launch_ms": spmv_avg_s * 1e3, "launches": launches,
                     # the streaming ceiling measured on this box: the library's
                     # own triad over N-vectors; frac_of_measured prices the SpMV against it
                     "measured_stream_GBs": stream["triad"], "measured_copy_GBs": stream["copy"],
                     "frac_of_measured": achieved / stream["triad"],
                     # what the kernel streams: lossless re-encodings of the CRS arrays (16-bit column codes; 1-byte value
                     # codes against a dictionary when the matrix has <= 256 distinct values) -- `achieved` above is priced
                     # on the CRS byte count and can exceed the peak; this is the HBM rate of the bytes moved
                     "spmv_stream": fmt, "streamed_bytes_per_launch": streamed_bytes,
                     "streamed_GBs": streamed_bytes / spmv_avg_s / 1e9,
                     "streamed_frac_of_peak": streamed_bytes / spmv_avg_s / 1e9 / HBM_PEAK_GBS},
    }
    if not args.no_cpu_baseline:
        cb, cpu_hist = cpu_baseline(n1, args.precond, args.cpu_iters)
        out["cpu_baseline"] = cb
        # parity of the timed run against the CPU path on the same input
        m = min(len(cpu_hist), len(hist))
        import numpy as np
        out["parity_max_dr_over_r0"] = float(np.max(np.abs(cpu_hist[:m] - hist[:m])) / cpu_hist[0])
    if tuned:
        out["placement_tuning"] = tuned
    cg.free()
    if fmt["val_bytes"] == 1:
        import numpy as np
        leg = crs_value_leg(ctx, A, b, x, D, min(args.steps, 50), min(args.warmup, 5))
        h2 = np.array(leg.pop("residual_history"))
        m2 = min(len(h2), len(hist))
        # same y bit for bit; the fused (Ap, p) is summed over different row blocks, a different fixed order
        leg["history_max_dev_over_r0_vs_timed_run"] = float(np.max(np.abs(h2[:m2] - np.array(hist[:m2]))) / h2[0])
        out["crs_value_stream"] = leg
    A.free(); b.free(); x.free()
    if n1 == 256 and not args.no_target_512:
        info = ctx.device_info()
        if info["hbm_bytes"] >= 200e9:
            out["target_512"] = target_512(ctx)
    print(json.dumps(out), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
