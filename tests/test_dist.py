"""Row-partitioned path, multi-process.  CPU: gloo, world_size 2 and 3 -- the
host-side plan (bis_halo_plan), the routing protocol and the distributed CG
schedule against the single-process oracle.  GPU: the same with the HIP
kernels and the C-ABI communicator callbacks (ranks share the one GPU)."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch(world, mode, kind, size, timeout=600):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(ROOT, "tests", "dist_worker.py"), mode, kind, str(size)]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert out.stdout.count(" OK") == world, out.stdout[-2000:]


@pytest.mark.parametrize("world,kind,size", [(2, "hpcg", 8), (3, "anderson", 6), (2, "anderson", 5),
                                             # the 8-way partition the metric is quoted on: plane-aligned slabs of HPCG
                                             # (2 planes per rank), periodic Anderson with one plane per rank (ranks 0 and
                                             # 7 exchange); closed-form halo / neighbour / interior counts, P = 8 history
                                             # equal to the P = 1 history to 1e-10 r0 (tests/dist_worker.py)
                                             (8, "hpcg", 16), (8, "anderson", 8), (4, "anderson", 8)])
def test_partitioned_cg_gloo_cpu(world, kind, size):
    launch(world, "cpu", kind, size, timeout=900)


@pytest.mark.parametrize("world", [2, 3])
def test_rccl_negotiation_is_collective(world):
    """A rank that cannot load / bind RCCL -- rank 0 included -- must not leave the others
    waiting: every rank learns the outcome in the same all-gather (launcher.negotiate_rccl_id)."""
    launch(world, "negotiate", "-", 0, timeout=300)


@pytest.mark.gpu
@pytest.mark.parametrize("world,kind,size", [(2, "hpcg", 12), (3, "anderson", 7), (1, "hpcg", 8),
                                             # four ranks on the one GPU (the box allows six GPU processes, this runner and the
                                             # launcher included): plane-aligned HPCG slabs and the periodic Anderson chain with
                                             # its wrap-around neighbour, closed-form halo / interior counts (tests/dist_worker.py)
                                             (4, "hpcg", 8), (4, "anderson", 8)])
def test_partitioned_cg_hip(world, kind, size):
    launch(world, "gpu", kind, size)


@pytest.mark.gpu
@pytest.mark.parametrize("world,precond,matrix", [(2, "none", "hpcg"), (3, "j", "hpcg"), (2, "j", "anderson"), (3, "j", "anderson0")])
def test_bench_multi_gpu_path_rehearsal(world, precond, matrix):
    """bench.py --gpus N exactly as the driver launches it (torch.distributed.run, one process
    per rank), on ONE GPU: BIS_BENCH_REHEARSE=1 puts every rank on cuda:0 with the gloo
    transport, so everything of the N > 1 bench except RCCL itself runs -- partition,
    closed-form residual check, timed loop, max-over-ranks, the JSON line."""
    import json
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--size", "48", "--steps", "6", "--warmup", "2",
           "--precond", precond, "--target-size", "0"]
    if matrix.startswith("anderson"):  # BASELINE config 3's runner: the conditioned twin and the config as named
        cmd += ["--matrix", "anderson", "--shift", "9" if matrix == "anderson" else "0"]
    env = dict(os.environ, OMP_NUM_THREADS="1", BIS_BENCH_REHEARSE="1")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == world and j["steps"] == 6 and j["warmup"] == 2 and j["scaling"] == "strong"
    assert j["value"] > 0 and j["unit"] == "CG iterations/s" and j["config"]["rows"] == 48 ** 3
    assert j["roofline"]["achieved"] > 0 and j["roofline"]["peak"] == 8000.0 * world
    if matrix != "anderson0":  # the raw Anderson operator is indefinite: CG need not reduce the residual
        assert 0 < j["residual_last"] < j["residual_r0"] * 10
    assert abs(j["residual_r0"] - j["residual_r0_closed_form"]) <= 1e-10 * j["residual_r0"]
    pr = j["per_rank"]
    assert [p["rank"] for p in pr] == list(range(world)) and sum(p["rows"] for p in pr) == 48 ** 3
    for p in pr:  # every rank of a z-slab partition exchanges with its neighbours on every SpMV
        assert p["halo_entries"] > 0 and p["send_entries"] > 0 and p["exchanges"] == 6 and p["allreduces"] == 12
        assert p["neighbours"] >= 1 and p["interior_rows"] > 0 and p["rccl_ranks_seen"] == 0  # rehearsal: no RCCL
    # the headline streams the CRS value array (north_star's CRS SpMV), priced on 12 nnz + 20 N ...
    assert pr[0]["spmv_stream"]["val_bytes"] == 8 and j["roofline"]["algorithmic_bytes_per_launch"] == 12 * j["config"]["nnz"] + 20 * 48 ** 3
    # ... and the library's default (compressed) format for these constant-coefficient / random-diagonal operators is a leg beside it
    assert j["compressed_stream"]["per_rank"][0]["spmv_stream"]["val_bytes"] < 8 and j["compressed_stream"]["value"] > 0


@pytest.mark.gpu
def test_bench_launches_its_own_ranks_and_runs_the_north_star_leg():
    """`python bench.py --gpus 3` invoked plainly (no torch.distributed.run around it, no WORLD_SIZE): bench.py starts its
    ranks as a child process before anything touches the GPU and relays rank 0's ONE JSON line; the record carries the
    strong-scaling leg on the north-star problem (`target_<size>`: HPCG 512^3 on a multi-GPU node; a rehearsal on one
    GPU takes twice the main grid edge) with both stream formats and the per-rank SpMV / exchange / all-reduce times."""
    import json
    env = dict(os.environ, BIS_BENCH_REHEARSE="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--size", "24", "--steps", "4", "--warmup", "1",
                          "--target-steps", "3"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 3 and j["config"]["rows"] == 24 ** 3 and j["metric"].startswith("CG iterations/sec")
    t = j["target_48"]
    assert t["config"]["rows"] == 48 ** 3 and t["n_gpus"] == 3 and t["steps"] == 3 and t["value"] > 0
    assert abs(t["residual_r0"] - t["residual_r0_closed_form"]) <= 1e-10 * t["residual_r0"]
    assert t["per_rank"][0]["spmv_stream"]["val_bytes"] == 8 and t["compressed_stream"]["per_rank"][0]["spmv_stream"]["val_bytes"] < 8
    for leg in (t, t["compressed_stream"]):
        for p in leg["per_rank"]:
            assert p["spmv_ms_per_iter"] > 0 and p["exchanges"] == 3 and p["allreduces"] == 6 and 0 < p["spmv_frac_of_peak"] < 1
            assert "rccl_ranks_seen" in p and p["exchange_ms_per_iter"] >= 0 and p["allreduce_ms_per_iter"] >= 0


def test_bench_refuses_a_world_that_does_not_match_gpus():
    """Under a launcher, --gpus must equal WORLD_SIZE (the driver's contract); a mismatch is an error, not a silent 1-GPU run."""
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert out.returncode != 0 and "--gpus 4" in (out.stderr + out.stdout)


@pytest.mark.gpu
@pytest.mark.parametrize("how", ["own-launcher", "torch.distributed.run"])
def test_bench_stuck_rank_is_reported(how):
    """A rank that gets stuck must not leave a run that is killed at the driver's limit with nothing written (round-4 verdict,
    item 4): with a rank put to sleep at the start of its `gen` phase (BIS_BENCH_STUCK) and the phase limit at 20 s, bench.py
    --gpus 2 -- started plainly (its own launcher + the parent watchdog) and exactly as the driver starts it -- ends with a
    non-zero status and ONE JSON line that names the rank and the phase, within the limit plus start-up."""
    import json
    import time
    args = ["--gpus", "2", "--size", "32", "--steps", "4", "--warmup", "1", "--target-size", "0"]
    env = dict(os.environ, OMP_NUM_THREADS="1", BIS_BENCH_REHEARSE="1", BIS_BENCH_STUCK="1:gen:900", BIS_PHASE_LIMIT_S="20")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "BIS_PHASE_DIR"):
        env.pop(k, None)
    if how == "own-launcher":
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
               "--master-port", str(free_port()), os.path.join(ROOT, "bench.py")] + args
    t0 = time.time()
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    secs = time.time() - t0
    assert out.returncode != 0, out.stdout[-2000:]
    lines = [json.loads(ln) for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:] + out.stderr[-3000:]
    d = lines[0]
    assert d["rank"] == 1 and d["phase"] == "gen" and "error" in d and d["n_gpus"] == 2, d
    assert d["phases"]["0"].startswith("gen") and "halo plan made" in d["phases"]["0"]  # rank 0 waits in the routing collective
    assert "[bis-phase] rank=1 phase=gen" in out.stderr
    assert secs < 20 + 150, secs  # the limit + interpreter / torch start-up on a fresh box, not the 900 s of the sleeper


@pytest.mark.gpu
def test_bench_distributed_code_path_at_one_rank_agrees_with_the_plain_one():
    """`--gpus 1` under BIS_FORCE_DIST=1 runs the partitioned code path (a 1-rank communicator, RCCL where it loads: halo plan,
    interior / boundary launches, device all-reduces) on the metric's problem; its rate must agree with the plain path's
    within 3 % -- what the N > 1 records inherit from N = 1 is then the same kernel at the same speed."""
    import json

    def run(extra_env):
        env = dict(os.environ, OMP_NUM_THREADS="1", **extra_env)
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "BIS_PHASE_DIR", "BIS_BENCH_REHEARSE"):
            env.pop(k, None)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "60", "--warmup", "10",
                              "--no-cpu-baseline", "--no-target-512", "--no-sweeps", "--no-configs"],
                             capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
        lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1
        return json.loads(lines[0])

    plain = run({})
    dist = run({"BIS_FORCE_DIST": "1"})
    assert dist["n_gpus"] == 1 and "transport" in dist and dist["config"]["rows"] == 256 ** 3
    if dist["transport"].startswith("rccl"):
        assert dist["rccl_ranks_seen"] == 1
    assert plain["roofline"]["spmv_stream"]["val_bytes"] == 8 and dist["per_rank"][0]["spmv_stream"]["val_bytes"] == 8
    # The SpMV kernel itself has two placement levels 13 % apart (where its stream lies in HBM: DESIGN.md section 4; the library
    # searches, but two processes need not end on the same level), so the comparison is made on what the partitioned path ADDS:
    # its iteration time beyond its own SpMV time must not exceed the plain path's by more than 3 % of an iteration ...
    over_plain = plain["ms_per_step"] - plain["roofline"]["avg_launch_ms"]
    over_dist = dist["ms_per_step"] - dist["per_rank"][0]["spmv_ms_per_iter"]
    assert over_dist <= over_plain + 0.03 * plain["ms_per_step"], (plain["ms_per_step"], plain["roofline"]["avg_launch_ms"], dist["ms_per_step"], dist["per_rank"][0]["spmv_ms_per_iter"])
    # ... its SpMV (three launches over row views) must be within the placement spread of the plain one, and the rates within it too
    assert dist["per_rank"][0]["spmv_ms_per_iter"] <= 1.16 * plain["roofline"]["avg_launch_ms"]
    assert 0.86 <= dist["value"] / plain["value"] <= 1.16, (plain["value"], dist["value"])
