"""Worker for the multi-process tests of the row-partitioned path (launched by
torch.distributed.run from test_dist_*.py).  Modes:

  cpu <kind> <size>   gloo, no GPU: bis_halo_plan (C-ABI, host-only) + the
                      launcher's routing protocol, with the ORACLE standing in
                      for the device kernels (test infrastructure only), against
                      the single-process oracle.
  negotiate . .       the collective RCCL-or-fallback decision of the N > 1 bench with
                      injected per-rank binding failures (no GPU, no RCCL).
  gpu <kind> <size>   gloo transport through the C-ABI communicator callbacks,
                      HIP kernels on cuda:0 (ranks share the one GPU of the
                      box), against the single-process oracle.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("OMP_NUM_THREADS", "1")

import numpy as np  # noqa: E402


def gen(orc, kind, size, row0=0, row1=None):
    if kind == "hpcg":
        return orc.gen_hpcg(size, row0=row0, row1=row1)
    return orc.gen_anderson(size, shift=9.0, row0=row0, row1=row1)


def main():
    mode, kind, size = sys.argv[1], sys.argv[2], int(sys.argv[3])
    import torch
    import torch.distributed as td
    from oracle.pyoracle import CRS, Oracle
    from basic_iterative_solvers_amd import halo_plan
    from basic_iterative_solvers_amd.launcher import even_row_starts, route_need_lists

    td.init_process_group("gloo")
    rank, world = td.get_rank(), td.get_world_size()
    if mode == "negotiate":
        # the N > 1 bench's transport decision is collective: a rank that cannot bind RCCL (rank 0
        # included) makes EVERY rank fall back at the same point -- nobody waits in a broadcast
        from basic_iterative_solvers_amd.launcher import negotiate_rccl_id

        def failing(on):
            def make():
                if rank == on:
                    raise OSError("librccl.so.1: cannot open shared object file")
                return bytes([rank]) * 128
            return make
        for bad in range(world):
            uid, why = negotiate_rccl_id(td, rank, world, failing(bad))
            assert uid is None and f"rank {bad}:" in why and "librccl" in why, (uid, why)
        uid, why = negotiate_rccl_id(td, rank, world, failing(-1))
        assert why is None and uid == bytes([0]) * 128  # everybody holds rank 0's id
        print(f"rank {rank}/{world} negotiate OK", flush=True)
        td.barrier()
        td.destroy_process_group()
        return
    orc = Oracle()
    N = size ** 3
    # uneven split on purpose when world == 3; plane-aligned for hpcg
    row_starts = even_row_starts(N, world, align=size * size if kind == "hpcg" else 1)
    row0, row1 = int(row_starts[rank]), int(row_starts[rank + 1])
    A_glob = gen(orc, kind, size)
    A_loc = gen(orc, kind, size, row0, row1)
    nl = row1 - row0
    rng = np.random.default_rng(99)
    x_glob = rng.uniform(-1, 1, N)
    y_ref = orc.spmv(A_glob, x_glob)[row0:row1]
    ref_cg = orc.solve(A_glob, "cg", "j")

    if mode == "cpu":
        halo, recv, interior = halo_plan(nl, A_loc.row_ptr, A_loc.col, world, rank, row_starts)
        # plan invariants
        assert np.all(np.diff(halo) > 0)
        assert np.all((halo < row0) | (halo >= row1))
        owners = np.searchsorted(row_starts, halo, side="right") - 1
        assert np.array_equal(np.bincount(owners, minlength=world), recv)
        a, b = int(interior[0]), int(interior[1])
        for r in range(a, b):  # interior rows reference only owned columns
            c = A_loc.col[A_loc.row_ptr[r]:A_loc.row_ptr[r + 1]]
            assert np.all((c >= row0) & (c < row1))
        send_counts, send_cols = route_need_lists(halo, recv, rank, world, td)
        assert np.all((send_cols >= row0) & (send_cols < row1))
        # renumber exactly as bis_dist_create does: owned -> c-row0, remote -> nl + index in halo
        col = A_loc.col.astype(np.int64)
        own = (col >= row0) & (col < row1)
        lcol = np.where(own, col - row0, nl + np.searchsorted(halo, col)).astype(np.int32)
        A_ren = CRS(nl, A_loc.row_ptr, lcol, A_loc.val, n_cols=nl + len(halo))

        def exchange(x_local):
            sendbuf = x_local[send_cols - row0]
            x_ext = np.concatenate([x_local, np.zeros(len(halo))])
            reqs, so, ro = [], 0, 0
            bufs = []
            for p in range(world):
                sc, rc = int(send_counts[p]), int(recv[p])
                if sc:
                    reqs.append(td.isend(torch.from_numpy(sendbuf[so:so + sc].copy()), p))
                if rc:
                    t = torch.zeros(rc, dtype=torch.float64)
                    bufs.append((ro, rc, t))
                    reqs.append(td.irecv(t, p))
                so += sc
                ro += rc
            for q in reqs:
                q.wait()
            for ro, rc, t in bufs:
                x_ext[nl + ro:nl + ro + rc] = t.numpy()
            return x_ext

        def allsum(v):
            t = torch.tensor([v], dtype=torch.float64)
            td.all_reduce(t)
            return float(t[0])

        y = orc.spmv(A_ren, exchange(x_glob[row0:row1].copy()))
        assert np.max(np.abs(y - y_ref)) <= 1e-13 * np.max(np.abs(y_ref)), "distributed SpMV"
        # distributed Jacobi-CG, cg.hpp schedule, reductions via all-reduce
        Dg = np.array([A_glob.val[A_glob.row_ptr[r]:A_glob.row_ptr[r + 1]][
            A_glob.col[A_glob.row_ptr[r]:A_glob.row_ptr[r + 1]] == r][0] for r in range(row0, row1)])
        bvec, x = np.full(nl, 1.0), np.full(nl, 0.1)
        r = bvec - orc.spmv(A_ren, exchange(x))
        z = r / Dg
        p = z.copy()
        hist = [np.sqrt(allsum(float(r @ r)))]
        stop = 1e-14 * hist[0]
        for it in range(1000):
            tmp = orc.spmv(A_ren, exchange(p))
            rz = allsum(float(r @ z))
            alpha = rz / allsum(float(tmp @ p))
            x = x + alpha * p
            r = r - alpha * tmp
            z = r / Dg
            beta = allsum(float(r @ z)) / rz
            p = z + beta * p
            hist.append(np.sqrt(allsum(float(r @ r))))
            if hist[-1] < stop:
                break
        hist = np.array(hist)
        m = min(len(hist), len(ref_cg["hist"]))
        dev = np.max(np.abs(hist[:m] - ref_cg["hist"][:m])) / ref_cg["hist"][0]
        assert dev <= 1e-10 and abs(len(hist) - len(ref_cg["hist"])) <= 1, (dev, len(hist))
        assert np.max(np.abs(x - ref_cg["x"][row0:row1])) <= 1e-9
        print(f"rank {rank}/{world} cpu {kind}-{size}: halo {len(halo)} interior [{a},{b}) "
              f"cg {len(hist) - 1} iters dev {dev:.1e} OK", flush=True)
    else:
        from basic_iterative_solvers_amd import Context, Dist
        from basic_iterative_solvers_amd.launcher import route_send_lists, setup_rccl, torch_comm_ops
        torch.cuda.set_device(0)
        ctx = Context(0)
        dA = ctx.matrix(A_loc)
        d = Dist(ctx, dA, rank, world, row_starts)
        # the device-side halo plan (only remote entries and boundary rows leave HBM) equals the
        # host planner on the full structure: same halo columns, same per-owner counts, same interior run
        h_halo, h_recv, h_int = halo_plan(nl, A_loc.row_ptr, A_loc.col, world, rank, row_starts)
        d_halo, d_recv = d.halo_info()
        assert np.array_equal(d_halo, h_halo) and np.array_equal(d_recv, h_recv)
        assert d.stats()["interior_rows"] == int(h_int[1] - h_int[0])
        # Jacobi diagonal of the local rows from the device (bis_mat_diag) == the matrix' own
        dA2 = ctx.matrix(A_loc)
        dD, dDinv = ctx.mat_diag(dA2, row0)
        route_send_lists(d, td)
        if (world == 1 and kind == "hpcg") or os.environ.get("BIS_TEST_FORCE_RCCL") == "1":
            setup_rccl(ctx, d, td)  # exercises the RCCL binding (self all-reduce)
        else:
            d.set_comm(torch_comm_ops(td, torch, world, rank))
        x_ext = ctx.alloc(d.n_ext)
        xe = np.zeros(d.n_ext)
        xe[:nl] = x_glob[row0:row1]
        x_ext.set(xe)
        y = ctx.alloc(nl)
        d.spmv(x_ext, y)
        yh = y.to_host()
        assert np.max(np.abs(yh - y_ref)) <= 1e-13 * np.max(np.abs(y_ref)), "distributed SpMV (HIP)"
        a, b2 = ctx.upload(x_glob[row0:row1]), ctx.upload(y_ref)
        gd = d.dot(a, b2)
        exact = float(x_glob @ orc.spmv(A_glob, x_glob))
        assert abs(gd - exact) <= 1e-12 * abs(exact) + 1e-12 * N
        Dg = np.array([A_glob.val[A_glob.row_ptr[r]:A_glob.row_ptr[r + 1]][
            A_glob.col[A_glob.row_ptr[r]:A_glob.row_ptr[r + 1]] == r][0] for r in range(row0, row1)])
        assert np.array_equal(dD.to_host(), Dg) and np.array_equal(dDinv.to_host(), 1.0 / Dg)
        bv, xv, Dv = ctx.upload(np.full(nl, 1.0)), ctx.upload(np.full(nl, 0.1)), dD
        cg = d.cg(bv, xv, Dv)
        r0 = cg.init(1e-14)
        cg.iterate(200)
        iters, conv, hist = cg.status()
        m = min(len(hist), len(ref_cg["hist"]))
        dev = np.max(np.abs(hist[:m] - ref_cg["hist"][:m])) / ref_cg["hist"][0]
        assert conv and dev <= 1e-10 and abs(iters - ref_cg["iters"]) <= 1, (conv, dev, iters)
        assert np.max(np.abs(xv.to_host() - ref_cg["x"][row0:row1])) <= 1e-9
        print(f"rank {rank}/{world} gpu {kind}-{size}: n_ext {d.n_ext} cg {iters} iters dev {dev:.1e} OK",
              flush=True)
        cg.free()
        td.barrier()
        d.free()
        ctx.close()
    td.barrier()
    td.destroy_process_group()


if __name__ == "__main__":
    main()
