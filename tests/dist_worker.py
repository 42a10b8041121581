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


def closed_form_partition(kind, size, world, rank, row_starts):
    """What a plane-aligned 1-D partition of the n^3 stencils must cost: every z-neighbour
    rank contributes one n^2 plane of halo entries -- HPCG (open boundaries): one neighbour at the two ends, two in
    between; Anderson (periodic): ranks 0 and P-1 are neighbours too, two neighbours everywhere once P > 2.  The
    interior rows are the planes that touch no remote plane.  None when the partition is not plane-aligned."""
    n2 = size * size
    if any(int(r) % n2 for r in row_starts):
        return None
    planes = (int(row_starts[rank + 1]) - int(row_starts[rank])) // n2
    if planes == 0 or world == 1:
        return None
    if kind == "hpcg":
        nb = [q for q in (rank - 1, rank + 1) if 0 <= q < world]
    else:
        nb = sorted({(rank - 1) % world, (rank + 1) % world} - {rank})
    # with two ranks of a periodic chain both faces go to the same neighbour: 2 n^2 entries from it
    faces = 2 if kind != "hpcg" else len(nb)
    halo = faces * n2
    recv = {q: (halo // len(nb)) for q in nb}
    lower = kind != "hpcg" or rank > 0          # is there a remote plane below / above this slab?
    upper = kind != "hpcg" or rank < world - 1
    interior = max(planes - int(lower) - int(upper), 0) * n2 if planes > 1 or not (lower or upper) else 0
    return dict(halo=halo, neighbours=len(nb), recv=recv, interior_rows=interior, send=halo)


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
        cf = closed_form_partition(kind, size, world, rank, row_starts)
        if cf:  # closed forms of the plane-aligned partition: halo entries, neighbours, per-owner counts, interior rows, sends
            assert len(halo) == cf["halo"] and int(np.count_nonzero(recv)) == cf["neighbours"] <= 2, (len(halo), recv, cf)
            assert all(int(recv[q]) == c for q, c in cf["recv"].items()), (recv, cf)
            assert b - a == cf["interior_rows"], (a, b, cf)
            assert len(send_cols) == cf["send"] and int(np.count_nonzero(send_counts)) == cf["neighbours"]
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
        cf = closed_form_partition(kind, size, world, rank, row_starts)
        if cf:
            stt = d.stats()
            assert stt["halo_entries"] == cf["halo"] and stt["interior_rows"] == cf["interior_rows"], (stt, cf)
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
        cg.free()
        stt = d.stats()
        if cf:
            assert stt["send_entries"] == cf["send"] and stt["neighbours"] == cf["neighbours"] <= 2, (stt, cf)
        print(f"rank {rank}: distributed SpMV / dot / Jacobi-CG done", flush=True)
        if True:
            # `world` processes share ONE GPU here and all run persistent sweep grids at the same time: option
            # "device_share" keeps every residency-bound grid (the wave-per-row sweep deals its rows statically) to
            # 1 / world of the device, so that the grids of all ranks fit together.  Without it a starved grid ends in
            # BIS_ERR_SYNC after about a second (every wait polls the fault word), never in a hang.
            ctx.set_option("device_share", world)
            # ---- block-Jacobi of the sweeps: every rank preconditions with SGS / ILU(0) of ITS diagonal block
            # (bis_mat_diag_block -> bis_mat_split_strict / bis_mat_ilu0 -> bis_cg_set_preconditioner).  Reference: the
            # same PCG with the block-diagonal preconditioner assembled from oracle sweeps on the blocks.
            blocks = []
            for q in range(world):
                a, bq = int(row_starts[q]), int(row_starts[q + 1])
                Ab = gen(orc, kind, size, a, bq)
                keep = (Ab.col >= a) & (Ab.col < bq)
                rows = np.repeat(np.arange(bq - a), np.diff(Ab.row_ptr))
                rpb = np.concatenate([[0], np.cumsum(np.bincount(rows[keep], minlength=bq - a))])
                blocks.append(CRS(bq - a, rpb, (Ab.col[keep] - a).astype(np.int32), Ab.val[keep]))
            for pcn in ("sgs", "ilu0"):
                facs = []
                for Bq in blocks:
                    if pcn == "ilu0":
                        Ls_, L_D_, Us_, U_D_ = orc.factor_ilu0(Bq)
                        facs.append((Ls_, Us_, None, None, L_D_, U_D_))
                    else:
                        L_, Ls_, U_, Us_ = orc.split_LU(Bq)
                        D_, Dinv_, _ = orc.peel_diag(L_)
                        facs.append((Ls_, Us_, D_, Dinv_, None, None))

                def Minv(v):
                    out = np.empty(N)
                    for q, f in enumerate(facs):
                        a, bq = int(row_starts[q]), int(row_starts[q + 1])
                        out[a:bq] = orc.apply_preconditioner(pcn, f[0], f[1], f[2], f[3], f[4], f[5], v[a:bq])
                    return out
                xg = np.full(N, 0.1)
                rg = np.full(N, 1.0) - orc.spmv(A_glob, xg)
                zg = Minv(rg)
                pg = zg.copy()
                ref_hist = [np.sqrt(float(rg @ rg))]
                for _ in range(300):
                    tg = orc.spmv(A_glob, pg)
                    rz = float(rg @ zg)
                    al = rz / float(tg @ pg)
                    xg = xg + al * pg
                    rg = rg - al * tg
                    zg = Minv(rg)
                    be = float(rg @ zg) / rz
                    pg = zg + be * pg
                    ref_hist.append(np.sqrt(float(rg @ rg)))
                    if ref_hist[-1] < 1e-14 * ref_hist[0]:
                        break
                ref_hist = np.array(ref_hist)
                print(f"rank {rank}: {pcn} reference PCG {len(ref_hist) - 1} iterations", flush=True)
                dAb = ctx.diag_block(dA2, row0)
                assert dAb.n_rows == nl and dAb.nnz == blocks[rank].nnz
                ones = ctx.upload(np.ones(nl))
                if pcn == "ilu0":
                    fLs, fL_D, fUs, fU_D = ctx.ilu0(dAb)
                    pargs = dict(Ls=fLs, Us=fUs, A_D=ones, A_D_inv=ones, L_D=fL_D, U_D=fU_D)
                else:
                    fLs, fUs, fD, fDinv = ctx.split_strict(dAb)
                    pargs = dict(Ls=fLs, Us=fUs, A_D=fD, A_D_inv=fDinv, L_D=ones, U_D=ones)
                print(f"rank {rank}: {pcn} factors of the diagonal block on the device", flush=True)
                bv2, xv2 = ctx.upload(np.full(nl, 1.0)), ctx.upload(np.full(nl, 0.1))
                cg2 = d.cg(bv2, xv2)
                cg2.set_preconditioner(pcn, **pargs)
                cg2.init(1e-14)
                print(f"rank {rank}: {pcn} init done", flush=True)
                cg2.iterate(300)
                it2, conv2, hist2 = cg2.status()
                m2 = min(len(hist2), len(ref_hist))
                dev2 = np.max(np.abs(hist2[:m2] - ref_hist[:m2])) / ref_hist[0]
                assert conv2 and dev2 <= 1e-10 and abs(it2 - (len(ref_hist) - 1)) <= 1, (pcn, conv2, dev2, it2, len(ref_hist))
                assert np.max(np.abs(xv2.to_host() - xg[row0:row1])) <= 1e-9
                cg2.free()
        print(f"rank {rank}/{world} gpu {kind}-{size}: n_ext {d.n_ext} cg {iters} iters dev {dev:.1e}; block-Jacobi SGS / ILU(0) PCG OK",
              flush=True)
        td.barrier()
        d.free()
        ctx.close()
    td.barrier()
    td.destroy_process_group()


if __name__ == "__main__":
    main()
