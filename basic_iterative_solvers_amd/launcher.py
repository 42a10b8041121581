"""Launcher-side plumbing for the row-partitioned path: torch.distributed is
used ONLY to (a) route the halo need-lists between ranks once at setup,
(b) broadcast the RCCL unique id, (c) provide the barrier / max-over-ranks of
the bench contract, and (d) as a stand-in transport in tests (gloo), where the
C-ABI's communicator callbacks are served by torch.distributed calls."""
import ctypes as C

import numpy as np

from . import CommOps, Dist, rccl_unique_id


def even_row_starts(n_rows, n_ranks, align=1):
    """Contiguous split; `align` keeps boundaries on multiples (e.g. nx*ny for z-slabs)."""
    units = n_rows // align
    starts = [(units * r // n_ranks) * align for r in range(n_ranks)] + [n_rows]
    return np.array(starts, dtype=np.int64)


def route_need_lists(halo, recv, rank, n_ranks, td, group=None):
    """Every rank tells each owner which of its columns it needs.  `halo` is
    the sorted remote-column list, `recv[p]` how many of them rank p owns.
    Returns (send_counts, send_cols): what the peers need from this rank."""
    need, off = {}, 0
    for p in range(n_ranks):
        need[p] = np.asarray(halo[off:off + int(recv[p])], dtype=np.int32).copy()
        off += int(recv[p])
    gathered = [None] * n_ranks
    td.all_gather_object(gathered, need, group=group)
    send_counts = np.array([len(gathered[q][rank]) for q in range(n_ranks)], dtype=np.int64)
    parts = [np.asarray(gathered[q][rank], dtype=np.int32) for q in range(n_ranks)]
    send_cols = np.concatenate(parts) if parts else np.zeros(0, np.int32)
    return send_counts, send_cols


def route_send_lists(dist_op, td, group=None):
    halo, recv = dist_op.halo_info()
    send_counts, send_cols = route_need_lists(halo, recv, dist_op.rank, dist_op.n_ranks, td, group)
    dist_op.set_send_lists(send_counts, send_cols)
    return send_counts


def negotiate_rccl_id(td, rank, n_ranks, make_id, group=None):
    """Collective decision whether the native RCCL transport can be used.  EVERY rank calls
    `make_id` (bis_rccl_unique_id: loads and binds librccl, ncclGetUniqueId) and the outcomes are
    all-gathered over the host-side group, so all ranks reach the same answer at the same point:
    (rank 0's id, None) if every rank could bind RCCL, else (None, reason).  A failure on one
    rank -- rank 0 included -- can therefore never leave the others waiting in a broadcast."""
    try:
        mine, err = make_id(), None
    except Exception as ex:  # BisError, OSError ... anything: report it, do not desert the collective
        mine, err = None, f"rank {rank}: {ex}"
    gathered = [None] * n_ranks
    td.all_gather_object(gathered, (mine, err), group=group)
    errors = [e for (_, e) in gathered if e]
    if errors or gathered[0][0] is None:
        return None, "; ".join(errors) or "rank 0 produced no id"
    return gathered[0][0], None


def setup_rccl(ctx, dist_op, td, group=None):
    """Native RCCL transport on every rank, or BisError on every rank (collective, see above)."""
    from . import BisError
    uid, why = negotiate_rccl_id(td, dist_op.rank, dist_op.n_ranks, lambda: rccl_unique_id(ctx), group)
    if uid is None:
        raise BisError(f"native RCCL transport unavailable: {why}")
    dist_op.use_rccl(uid)


class _DevArray:
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = dict(data=(int(ptr), False), shape=(int(n),), typestr="<f8",
                                             version=2, strides=None)


def torch_comm_ops(td, torch, n_ranks, rank, group=None):
    """bis_comm_ops served by torch.distributed (tests: gloo on GPU tensors)."""

    def wrap(ptr, n):
        return torch.as_tensor(_DevArray(ptr, n), device="cuda")

    import os
    trace = os.environ.get("BIS_DIST_TRACE") == "1"

    # gloo reduces host memory: stage the (tiny) device buffers through CPU tensors explicitly -- wait for the
    # library's stream, copy out, reduce, copy back -- instead of handing gloo device tensors
    host_staged = td.get_backend(group) == "gloo"

    def allreduce(user, stream, buf, count):
        if trace:
            print(f"rank {rank}: allreduce({count}) enter", flush=True)
        try:
            with torch.cuda.stream(torch.cuda.ExternalStream(stream)):
                t = wrap(buf, count)
                if host_staged:
                    h = t.cpu()
                    if trace:
                        print(f"rank {rank}: allreduce({count}) stream drained", flush=True)
                    td.all_reduce(h, group=group)
                    t.copy_(h)
                else:
                    td.all_reduce(t, group=group)
            if trace:
                print(f"rank {rank}: allreduce({count}) done", flush=True)
            return 0
        except Exception as ex:  # noqa
            print("allreduce callback failed:", ex, flush=True)
            return 1

    def exchange(user, stream, sendbuf, send_counts, recvbuf, recv_counts, n):
        if trace:
            print(f"rank {rank}: exchange enter", flush=True)
        try:
            with torch.cuda.stream(torch.cuda.ExternalStream(stream)):
                torch.cuda.current_stream().synchronize()
                ops, so, ro = [], 0, 0
                back = []
                for p in range(n):
                    sc, rc = int(send_counts[p]), int(recv_counts[p])
                    if sc:
                        st = wrap(sendbuf + 8 * so, sc)
                        ops.append(td.P2POp(td.isend, st.cpu() if host_staged else st, p, group))
                    if rc:
                        rt = wrap(recvbuf + 8 * ro, rc)
                        if host_staged:
                            hr = torch.empty(rc, dtype=torch.float64)
                            back.append((rt, hr))
                            rt = hr
                        ops.append(td.P2POp(td.irecv, rt, p, group))
                    so += sc
                    ro += rc
                if ops:  # batched: safe for both gloo and nccl process groups
                    for r in td.batch_isend_irecv(ops):
                        r.wait()
                for rt, hr in back:
                    rt.copy_(hr)
                torch.cuda.current_stream().synchronize()
            if trace:
                print(f"rank {rank}: exchange done", flush=True)
            return 0
        except Exception as ex:  # noqa
            print("exchange callback failed:", ex, flush=True)
            return 1

    ops = CommOps()
    ops.user = None
    ops.allreduce_sum = type(ops.allreduce_sum)(allreduce)
    ops.exchange = type(ops.exchange)(exchange)
    return ops
