"""In-loop against stand-alone SpMV (win8), over time: per-chunk SpMV time of a long CG run (a throttling GPU slows down as
the run goes on; a placement effect is there from the first chunk), and the stand-alone kernel right after the loop."""
import json
import subprocess
import sys

sys.path.insert(0, ".")
from basic_iterative_solvers_amd import Context  # noqa: E402

n1 = int(sys.argv[1]) if len(sys.argv) > 1 else 256
chunks = int(sys.argv[2]) if len(sys.argv) > 2 else 12
ctx = Context()
ctx.set_option("spmv_valdict", 0)
A = ctx.gen_hpcg(n1)
N = A.n_rows
xs, ys = ctx.alloc(N), ctx.alloc(N)
ctx.init_vector(xs, 0.5)


def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp"], capture_output=True, text=True, timeout=20).stdout
        keep = [l.strip() for l in out.splitlines() if any(k in l for k in ("sclk", "mclk", "fclk", "Power", "junction", "edge"))]
        return keep[:12]
    except Exception as ex:  # noqa: BLE001
        return [repr(ex)]


def alone(x, y, reps=6):
    ctx.sync(); ctx.profile(True)
    for _ in range(reps):
        ctx.spmv(A, x, y)
    ctx.sync()
    k, ms = ctx.profile_read(); ctx.profile(False)
    return ms / max(k, 1)


ctx.spmv(A, xs, ys)
print(json.dumps({"tuning": A.win8_tuning(), "alone_cold_ms": alone(xs, ys), "smi": smi()}), flush=True)
for trial in range(3):
    b, x = ctx.alloc(N), ctx.alloc(N)
    ctx.init_vector(b, 1.0); ctx.init_vector(x, 0.1)
    cg = ctx.cg(A, b, x)
    cg.init(0.0)
    per = []
    for c in range(chunks):
        ctx.sync(); ctx.profile(True)
        cg.iterate(25)
        ctx.sync()
        k, ms = ctx.profile_read(); ctx.profile(False)
        per.append(round(ms / max(k, 1), 4))
    hot = alone(xs, ys)
    print(json.dumps({"trial": trial, "in_loop_ms_per_25": per, "alone_right_after_ms": hot, "alone_on_b_x_ms": alone(b, x), "smi": smi() if trial == 0 else None}), flush=True)
    cg.free(); b.free(); x.free()
    keep = ctx.alloc((1 + trial) * (1 << 26))  # 0.5, 1 GiB kept: the next CG's vectors land elsewhere
