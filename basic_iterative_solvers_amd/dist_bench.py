"""bench.py, N > 1: strong scaling of the fused CG on the row-partitioned HPCG
operator (z-slabs), one process per GPU, RCCL over xGMI."""
import json
import os
import time

import numpy as np

HBM_PEAK_GBS = 8000.0


def run_distributed(args, rank, world, local_rank):
    import torch
    import torch.distributed as td

    from . import Context, Dist
    from . import BisError
    from .launcher import even_row_starts, route_send_lists, setup_rccl, torch_comm_ops

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    # BIS_BENCH_REHEARSE=1 (tests on a one-GPU box): every rank uses cuda:0, gloo process group,
    # torch.distributed transport through the C-ABI callbacks -- everything of this function except RCCL.
    rehearse = os.environ.get("BIS_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
        torch.cuda.set_device(0)
        td.init_process_group("gloo")
        host_group = None
    else:
        td.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        host_group = td.new_group(backend="gloo")  # host-side object routing

    ctx = Context(local_rank)
    n1 = args.size
    N = n1 ** 3
    row_starts = even_row_starts(N, world, align=n1 * n1)
    row0, row1 = int(row_starts[rank]), int(row_starts[rank + 1])
    if getattr(args, "tune_placement", 0) > 0:  # applied inside bis_dist_create, before the row views
        ctx.lib.bis_set_option(b"tune_placement", int(args.tune_placement))
    A = ctx.gen_hpcg(n1, row0=row0, row1=row1)
    nnz_local = A.nnz
    d = Dist(ctx, A, rank, world, row_starts)
    route_send_lists(d, td, group=host_group)
    transport = "rccl (native ncclSend/ncclRecv + ncclAllReduce on the library's streams)"
    try:
        if rehearse:
            raise BisError("rehearsal: RCCL cannot put two ranks on one GPU")
        setup_rccl(ctx, d, td, group=host_group)
        ok = 1
    except (BisError, OSError, RuntimeError) as ex:  # e.g. RCCL not loadable: fall back to torch.distributed's communicator
        print(f"rank {rank}: native RCCL transport unavailable ({ex}); using torch.distributed", flush=True)
        ok = 0
    flag = torch.tensor([ok], device="cuda")
    td.all_reduce(flag, op=td.ReduceOp.MIN)
    if int(flag.item()) == 0:
        d.set_comm(torch_comm_ops(td, torch, world, rank))
        transport = "torch.distributed through the C-ABI communicator callbacks"
    nl = d.n_local
    b, x = ctx.alloc(nl), ctx.alloc(nl)
    ctx.init_vector(b, 1.0)
    ctx.init_vector(x, 0.1)
    D = None
    if args.precond == "j":
        D = ctx.alloc(nl)
        ctx.init_vector(D, 26.0)
    cg = d.cg(b, x, D)
    r0 = cg.init(0.0)
    # partition-independent check of the distributed operator: ||b - A x0||_2 in closed form
    # (row sums of the HPCG operator: 27 - cx*cy*cz neighbours), x0 = 0.1, b = 1
    c = np.full(n1, 3.0); c[0] = c[-1] = 2.0
    rowsum = 27.0 - c[:, None, None] * c[None, :, None] * c[None, None, :]
    r0_exact = float(np.sqrt(np.sum((1.0 - 0.1 * rowsum) ** 2)))
    if not abs(r0 - r0_exact) <= 1e-10 * r0_exact:
        raise SystemExit(f"rank {rank}: distributed residual {r0!r} != {r0_exact!r}: partitioned operator is wrong")
    cg.iterate(args.warmup)
    ctx.sync()
    torch.cuda.synchronize()
    td.barrier()
    torch.cuda.synchronize()

    ctx.profile(True)
    t0 = time.perf_counter()
    cg.iterate(args.steps)
    torch.cuda.synchronize()
    td.barrier()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    ctx.profile(False)
    launches, spmv_ms = ctx.profile_read()
    iters, conv, hist = cg.status(hist_cap=args.warmup + args.steps + 1)
    el = torch.tensor([t1 - t0], dtype=torch.float64, device="cuda")
    td.all_reduce(el, op=td.ReduceOp.MAX)
    secs = float(el.item())
    tot = torch.tensor([float(nnz_local), float(spmv_ms), float(iters)], dtype=torch.float64, device="cuda")
    mx = tot.clone()
    td.all_reduce(tot, op=td.ReduceOp.SUM)
    td.all_reduce(mx, op=td.ReduceOp.MAX)
    nnz = int(tot[0].item())
    if rank == 0:
        if iters != args.warmup + args.steps:
            raise SystemExit(f"timed region invalid: {iters} iterations executed")
        its = args.steps / secs
        # the interior + two boundary launches together are one distributed SpMV
        spmv_avg_s = float(mx[1].item()) * 1e-3 / args.steps
        spmv_bytes = 12 * nnz + 20 * N
        achieved = spmv_bytes / spmv_avg_s / 1e9
        out = {
            "metric": "CG iterations/sec + SpMV GFLOP/s (% HBM roofline), HPCG 256^3 at 1/2/4/8 GPUs",
            "value": its, "unit": "CG iterations/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * secs / args.steps,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"HPCG {n1}^3 27-point, -cg" +
                                   (" -p j" if args.precond == "j" else "") +
                                   ", b=1 x0=0.1, fused device schedule", "rows": N, "nnz": nnz,
                       "partition": f"1-D row blocks (z-slabs) over {world} GPUs, RCCL send/recv halo + "
                                    "2 all-reduces per iteration"},
            "spmv_gflops": 2.0 * nnz / spmv_avg_s / 1e9,
            "residual_r0": r0, "residual_last": float(hist[-1]), "transport": transport,
            "roofline": {"bound": "hbm", "kernel": "spmv_rowblock_kernel (interior + boundary launches)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS * world, "unit": "GB/s",
                         "frac": achieved / (HBM_PEAK_GBS * world), "traffic": None,
                         "algorithmic_bytes_per_launch": spmv_bytes,
                         "avg_launch_ms": spmv_avg_s * 1e3, "launches": launches},
        }
        print(json.dumps(out), flush=True)
    cg.free()
    td.barrier()
    d.free()
    ctx.close()
    td.destroy_process_group()
