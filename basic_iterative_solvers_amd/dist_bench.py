"""bench.py, N > 1: strong scaling of the fused CG on the row-partitioned operator
(z-slabs), one process per GPU, RCCL over xGMI.  The record's headline streams the CRS value
array (north_star's CRS SpMV); `compressed_stream` is the same solve with the library's default
stream format, and `target_512` both on the north-star problem (HPCG 512^3, strong scaling:
every rank generates its z-slab).  --matrix hpcg is the BASELINE metric
(HPCG 256^3, -cg); --matrix anderson --precond j is BASELINE config 3 (Anderson 256^3,
Jacobi-preconditioned CG, row-partitioned): shift 0 is the config as named (indefinite: a
timing workload), shift 9 its conditioned twin with the r0 check."""
import json
import os
import sys
import time

import numpy as np

HBM_PEAK_GBS = 8000.0


def anderson_diag(L, W, shift, seed, row0, row1):
    """Diagonal of the Anderson generator (bis_mat_gen_anderson / oracle orc_gen_anderson): the same
    counter hash in numpy uint64 arithmetic -- the partition-independent side of the r0 check."""
    i = np.arange(row0, row1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15) + (i + np.uint64(1)) * np.uint64(0xD1B54A32D192ED03)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    u = (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    return W * (u - 0.5) + shift


def r0_closed_form(matrix, shift, n1):
    """||b - A x0||_2 for b = 1, x0 = 0.1 from the row sums alone (independent of the partition)."""
    if matrix == "hpcg":  # row sum 27 - cx*cy*cz
        c = np.full(n1, 3.0); c[0] = c[-1] = 2.0
        rowsum = 27.0 - c[:, None, None] * c[None, :, None] * c[None, None, :]
        return float(np.sqrt(np.sum((1.0 - 0.1 * rowsum) ** 2)))
    # Anderson: diagonal + 6 off-diagonals of -t (t = 1); summed in chunks of 2^22 rows
    tot, N = 0.0, n1 ** 3
    for a in range(0, N, 1 << 22):
        d = anderson_diag(n1, 5.0, shift, 1, a, min(N, a + (1 << 22)))
        tot += float(np.sum((1.0 - 0.1 * (d - 6.0)) ** 2))
    return float(np.sqrt(tot))


KERNELS = ("spmv_rowblock_kernel", "spmv_rowblock_vd_kernel", "spmv_rowmajor_vd_kernel", "spmv_rowmajor_vd_kernel",
           "spmv_sellwin_kernel", "spmv_sellwin_kernel", "spmv_win8_kernel")


class _Env:
    """What every leg of the N > 1 record shares: the process group(s), the context, the transport choice."""
    pass


def cg_leg(E, args, n1, matrix, precond, shift, steps, warmup, valdict):
    """One partitioned problem, one stream format: generate this rank's z-slab of the n1^3 operator in HBM, build the
    distributed operator, check ||b - A x0|| against the closed form, time `steps` CG iterations between barriers.
    Returns rank 0's record (None elsewhere).  valdict = 0: the CRS value array is streamed (north_star's CRS SpMV),
    -1: the library's default stream format for the matrix."""
    import torch
    td, ctx, rank, world = E.td, E.ctx, E.rank, E.world
    from . import BisError, Dist, rccl_unique_id
    from .launcher import even_row_starts, negotiate_rccl_id, route_send_lists, torch_comm_ops

    def stage(name):  # a phase of the watchdog's board; inside the north-star leg every stage is a checkpoint of `512-leg`
        if E.leg_phase:
            E.wd.tick(name)
        else:
            E.wd.enter(name)

    stage("gen")
    N = n1 ** 3
    ctx.set_option("spmv_valdict", valdict)
    row_starts = even_row_starts(N, world, align=n1 * n1)
    row0, row1 = int(row_starts[rank]), int(row_starts[rank + 1])
    t_setup = time.perf_counter()
    if matrix == "hpcg":
        A = ctx.gen_hpcg(n1, row0=row0, row1=row1)
    else:
        A = ctx.gen_anderson(n1, shift=shift, row0=row0, row1=row1)
    nnz_local = A.nnz
    D = None
    if precond == "j":  # the Jacobi diagonal of this rank's rows, taken before the columns are renumbered
        D, Dinv = ctx.mat_diag(A, row0)
        Dinv.free()
    E.wd.tick("slab generated")
    d = Dist(ctx, A, rank, world, row_starts)
    setup_s = time.perf_counter() - t_setup
    E.wd.tick("halo plan made")
    route_send_lists(d, td, group=E.host_group)
    transport = "rccl (native ncclSend/ncclRecv + ncclAllReduce on the library's streams)"
    try:
        if E.rehearse:
            raise BisError("rehearsal: RCCL cannot put two ranks on one GPU")
        stage("rccl-id")
        uid, why = negotiate_rccl_id(td, rank, world, lambda: rccl_unique_id(ctx), E.host_group)
        if uid is None:
            raise BisError(f"native RCCL transport unavailable: {why}")
        stage("comm-init")
        d.use_rccl(uid)  # ncclCommInitRank: collective
        ok = 1
    except (BisError, OSError, RuntimeError) as ex:  # e.g. RCCL not loadable: fall back to torch.distributed's communicator
        print(f"rank {rank}: native RCCL transport unavailable ({ex}); using torch.distributed", flush=True)
        ok = 0
    flag = torch.tensor([ok], device="cuda")
    td.all_reduce(flag, op=td.ReduceOp.MIN)
    if int(flag.item()) == 0:
        d.set_comm(torch_comm_ops(td, torch, world, rank))
        transport = "torch.distributed through the C-ABI communicator callbacks"
    E.wd.tick("transport chosen")
    nl = d.n_local
    b, x = ctx.alloc(nl), ctx.alloc(nl)
    ctx.init_vector(b, 1.0)
    ctx.init_vector(x, 0.1)
    cg = d.cg(b, x, D)
    r0 = cg.init(0.0)
    # partition-independent check of the distributed operator: ||b - A x0||_2 from the row sums
    # (x0 = 0.1, b = 1); every rank evaluates the same closed form
    r0_exact = r0_closed_form(matrix, shift, n1)
    if not abs(r0 - r0_exact) <= 1e-10 * r0_exact:
        raise SystemExit(f"rank {rank}: distributed residual {r0!r} != {r0_exact!r}: partitioned operator is wrong")
    stage("warmup")
    cg.iterate(warmup)
    ctx.sync()
    torch.cuda.synchronize()
    E.wd.tick("warm-up iterations done")
    td.barrier()
    torch.cuda.synchronize()

    stage("timed")
    ctx.profile(True)
    t0 = time.perf_counter()
    cg.iterate(steps)
    torch.cuda.synchronize()
    td.barrier()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    stage("report")
    ctx.profile(False)
    launches, spmv_ms = ctx.profile_read()
    comm = d.profile_read()
    iters, conv, hist = cg.status(hist_cap=warmup + steps + 1)
    col_b, val_b, n_dict, form = d.spmv_stream_info()
    kernel = "spmv_sellmask_kernel" if form >= 4 and col_b == 0 else KERNELS[form]
    streamed = d.spmv_streamed_bytes()
    crs_local = 12 * nnz_local + 20 * nl  # this rank's share of SURVEY 8d's algorithmic bytes
    spmv_s = max(spmv_ms / max(steps, 1), 1e-9) * 1e-3
    mine = dict(rank=rank, device=E.local_rank, rows=nl, nnz=nnz_local, setup_s=setup_s, **d.stats(),
                spmv_ms_per_iter=spmv_ms / max(steps, 1),
                exchange_ms_per_iter=comm["exchange_ms"] / max(steps, 1),
                allreduce_ms_per_iter=comm["allreduce_ms"] / max(steps, 1),
                exchanges=comm["exchanges"], allreduces=comm["allreduces"],
                spmv_streamed_bytes=streamed,
                # this rank's share: its three launches per SpMV against ITS HBM
                spmv_streamed_GBs=streamed / spmv_s / 1e9,
                spmv_algorithmic_GBs=crs_local / spmv_s / 1e9,
                spmv_frac_of_peak=(crs_local if val_b == 8 else streamed) / spmv_s / 1e9 / HBM_PEAK_GBS,
                spmv_stream=dict(col_bytes=col_b, val_bytes=val_b, dictionary_values=n_dict, kernel=kernel, form=form,
                                 per_row_diagonal=form in (3, 5)))
    per_rank = [None] * world
    td.all_gather_object(per_rank, mine, group=E.host_group)
    el = torch.tensor([t1 - t0], dtype=torch.float64, device="cuda")
    td.all_reduce(el, op=td.ReduceOp.MAX)
    secs = float(el.item())
    tot = torch.tensor([float(nnz_local), float(spmv_ms), float(iters), float(streamed)], dtype=torch.float64, device="cuda")
    mx = tot.clone()
    td.all_reduce(tot, op=td.ReduceOp.SUM)
    td.all_reduce(mx, op=td.ReduceOp.MAX)
    nnz = int(tot[0].item())
    rec = None
    seen = [p["rccl_ranks_seen"] for p in per_rank]
    if transport.startswith("rccl") and any(v != world for v in seen):
        # the communicator every rank built must span the whole world: anything else is a partition that exchanged with nobody
        raise SystemExit(f"rank {rank}: RCCL communicators span {seen} ranks, expected {world} on every rank")
    if rank == 0:
        if iters != warmup + steps:
            raise SystemExit(f"timed region invalid: {iters} iterations executed")
        its = steps / secs
        # the interior + two boundary launches together are one distributed SpMV
        spmv_avg_s = float(mx[1].item()) * 1e-3 / steps
        # CRS value stream: priced on SURVEY 8d's algorithmic bytes; a compressed format on the bytes the ranks' kernels move
        # (bis_dist_spmv_streamed_bytes) -- both against the aggregate HBM peak, on the slowest rank's SpMV time
        spmv_bytes = 12 * nnz + 20 * N
        moved = float(tot[3].item())
        crs_stream = val_b == 8
        achieved = (spmv_bytes if crs_stream else moved) / spmv_avg_s / 1e9
        name = f"HPCG {n1}^3 27-point" if matrix == "hpcg" else \
            f"Anderson {n1}^3 7-point periodic W=5 shift={shift:g}"
        rec = {
            "value": its, "unit": "CG iterations/s", "n_gpus": world, "steps": steps,
            "warmup": warmup, "ms_per_step": 1e3 * secs / steps,
            "config": {"workload": f"{name}, -cg" + (" -p j" if precond == "j" else "") +
                                   ", b=1 x0=0.1, fused device schedule, " +
                                   ("CRS values streamed (8 B per non-zero)" if crs_stream else "default (compressed) stream format"),
                       "rows": N, "nnz": nnz,
                       "partition": f"1-D row blocks (z-slabs) over {world} GPUs, RCCL send/recv halo + "
                                    "2 all-reduces per iteration"},
            "spmv_gflops": 2.0 * nnz / spmv_avg_s / 1e9,
            "residual_r0": r0, "residual_r0_closed_form": r0_exact, "residual_last": float(hist[-1]),
            "transport": transport, "rccl_ranks_seen": min(seen), "per_rank": per_rank,
            "roofline": {"bound": "hbm", "kernel": kernel + " (interior + boundary launches; rank 0's interior rows)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS * world, "unit": "GB/s",
                         "frac": achieved / (HBM_PEAK_GBS * world), "traffic": None,
                         "priced_on": ("algorithmic bytes of the CRS SpMV (12 nnz + 20 N, SURVEY 8d)" if crs_stream else
                                       "bytes of the ranks' stream formats") + " (slowest rank's SpMV time, aggregate peak)",
                         "algorithmic_bytes_per_launch": spmv_bytes,
                         "moved_bytes_per_launch": moved, "moved_GBs": moved / spmv_avg_s / 1e9,
                         "avg_launch_ms": spmv_avg_s * 1e3, "launches": launches},
        }
        if not crs_stream:
            rec["roofline"]["crs_equivalent_GBs"] = spmv_bytes / spmv_avg_s / 1e9
    cg.free()
    td.barrier()
    b.free(); x.free()
    if D is not None:
        D.free()
    d.free()
    ctx.set_option("spmv_valdict", -1)
    return rec


def run_distributed(args, rank, world, local_rank):
    import torch
    import torch.distributed as td

    from . import Context

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    # stdout carries ONE JSON line (bench.py's contract): the banners RCCL and gloo print from C at start-up ("RCCL version :
    # ...", "[Gloo] Rank 0 is connected ...") go to stderr -- file descriptor 1 points at stderr until the record is printed
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    # BIS_BENCH_REHEARSE=1 (tests on a one-GPU box): every rank uses cuda:0, gloo process group,
    # torch.distributed transport through the C-ABI callbacks -- everything of this function except RCCL.
    from .watchdog import RankWatchdog
    E = _Env()
    # from here on a rank that outlives a phase limit ends the run with ONE diagnostic JSON line on the bench's stdout
    E.wd = RankWatchdog(rank, world, out_fd=saved_stdout)
    E.leg_phase = None
    E.rehearse = os.environ.get("BIS_BENCH_REHEARSE") == "1"
    if E.rehearse:
        local_rank = 0
        torch.cuda.set_device(0)
        td.init_process_group("gloo")
        E.host_group = None
    else:
        td.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        E.host_group = td.new_group(backend="gloo")  # host-side object routing
    E.td, E.rank, E.world, E.local_rank = td, rank, world, local_rank
    E.ctx = ctx = Context(local_rank)
    if getattr(args, "tune_placement", 0) > 0:  # applied inside bis_dist_create, before the row views
        ctx.lib.bis_set_option(b"tune_placement", int(args.tune_placement))
    headline_valdict = -1 if getattr(args, "headline", "crs") == "default" else 0
    n1 = args.size
    out = cg_leg(E, args, n1, args.matrix, args.precond, args.shift, args.steps, args.warmup, headline_valdict)
    extra = {}
    if headline_valdict == 0:
        leg = cg_leg(E, args, n1, args.matrix, args.precond, args.shift, min(args.steps, 50), min(args.warmup, 5), -1)
        if rank == 0 and leg["per_rank"][0]["spmv_stream"]["val_bytes"] < 8:
            leg["note"] = "the library's default stream format for this matrix (lossless re-encoding of the CRS arrays, bit-identical y)"
            extra["compressed_stream"] = leg
    # the north-star sentence is quoted on HPCG 512^3: the same partitioned solve on that problem, both stream formats (each
    # rank generates its own z-slab: at N >= 2 a slab has fewer than 2^31 non-zeros and takes the 32-bit row pointers)
    t512 = getattr(args, "target_size", 0)
    if E.rehearse and t512 == 512:
        t512 = 2 * n1  # a one-GPU rehearsal of the leg (all ranks share cuda:0): same code, a problem that fits beside itself
    if world > 1 and args.matrix == "hpcg" and t512 > 0 and not getattr(args, "no_target_512", False):
        info = ctx.device_info()
        slab_bytes = 12.0 * 27 * (t512 ** 3) / world * 1.35 + 8.0 * 12 * (t512 ** 3) / world
        if info["hbm_bytes"] * (0.5 if E.rehearse else 0.85) >= slab_bytes * (world if E.rehearse else 1):
            E.wd.enter("512-leg")
            E.leg_phase = "512-leg"
            t = cg_leg(E, args, t512, "hpcg", "none", 0.0, args.target_steps, 3, 0)
            E.wd.enter("512-leg")  # (a fresh fuse for the second format's leg)
            tc = cg_leg(E, args, t512, "hpcg", "none", 0.0, args.target_steps, 3, -1)
            E.leg_phase = None
            if rank == 0:
                if tc["per_rank"][0]["spmv_stream"]["val_bytes"] < 8:
                    t["compressed_stream"] = tc
                extra["target_%d" % t512] = t
    E.wd.enter("report")
    if rank == 0:
        rec = {"metric": "CG iterations/sec + SpMV GFLOP/s (% HBM roofline), HPCG 256^3 at 1/2/4/8 GPUs",
               "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic"}
        rec.update(out)
        rec.update(extra)
        rec["options"] = ctx.options()
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        print(json.dumps(rec), flush=True)
        os.dup2(2, 1)
    td.barrier()
    E.wd.enter("done")
    E.wd.stop()
    ctx.close()
    td.destroy_process_group()
