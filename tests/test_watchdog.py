"""CPU: the multi-rank bench's watchdog (basic_iterative_solvers_amd/watchdog.py) with stand-in ranks: a stuck rank ends the
run with ONE diagnostic JSON line naming the rank and its phase, within the limit, with a non-zero status -- reported by a
rank's own watchdog thread, or by the launcher when the interpreters themselves hang.  (The same through bench.py's real
ranks on a GPU: tests/test_dist.py::test_bench_stuck_rank_is_reported.)"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "wd_worker.py")
DRIVER = ("import sys; sys.path.insert(0, %r); from basic_iterative_solvers_amd.watchdog import supervise; "
          "raise SystemExit(supervise([sys.executable, %r, '--spawn', sys.argv[1], '--mode', sys.argv[2]], int(sys.argv[1])))" % (ROOT, WORKER))


def run(world, mode, **env):
    e = dict(os.environ, **{k: str(v) for k, v in env.items()})
    e.pop("BIS_PHASE_DIR", None)
    t0 = time.time()
    out = subprocess.run([sys.executable, "-c", DRIVER, str(world), mode], capture_output=True, text=True, timeout=120, env=e)
    return out, time.time() - t0


def json_lines(out):
    return [json.loads(ln) for ln in out.stdout.splitlines() if ln.startswith("{")]


def test_healthy_ranks_pass_through():
    out, secs = run(3, "ok", BIS_PHASE_LIMIT_S=20)
    assert out.returncode == 0, out.stderr[-2000:]
    assert json_lines(out) == [{"value": 1.0}]
    assert "[bis-phase] rank=2 phase=timed" in out.stderr


def test_stuck_rank_is_named_by_a_rank_watchdog():
    out, secs = run(3, "ok", BIS_PHASE_LIMIT_S=3, BIS_BENCH_STUCK="1:warmup:300")
    assert out.returncode != 0
    lines = json_lines(out)
    assert len(lines) == 1, out.stdout + out.stderr[-2000:]
    d = lines[0]
    assert d["rank"] == 1 and d["phase"] == "warmup" and "error" in d and d["n_gpus"] == 3
    assert d["reported_by"] == "rank 0" and d["phases"]["1"].startswith("warmup")
    assert secs < 30  # limit 3 s + process start-up, not the 300 s the rank would sleep


def test_hung_interpreter_is_reported_by_the_launcher():
    """rank 1 never reaches its watchdog (an import that hangs): the other ranks' own fuses name it (it is still at `start`)
    -- and where they cannot, the launcher's longer fuse does."""
    out, secs = run(2, "hang-import", BIS_PHASE_LIMIT_S=3)
    assert out.returncode != 0
    lines = json_lines(out)
    assert len(lines) == 1, out.stdout + out.stderr[-2000:]
    assert lines[0]["rank"] == 1 and lines[0]["phase"] == "start"
    assert secs < 40


def test_launcher_reports_when_no_rank_can(tmp_path):
    """every interpreter hangs before its watchdog exists: only the launcher is left to say so"""
    script = tmp_path / "sleepers.py"
    script.write_text("import time; time.sleep(600)\n")
    drv = ("import sys; sys.path.insert(0, %r); from basic_iterative_solvers_amd.watchdog import supervise; "
           "raise SystemExit(supervise([sys.executable, %r], 2))" % (ROOT, str(script)))
    t0 = time.time()
    out = subprocess.run([sys.executable, "-c", drv], capture_output=True, text=True, timeout=120,
                         env=dict(os.environ, BIS_PHASE_LIMIT_S="2"))
    assert out.returncode == 3
    lines = json_lines(out)
    assert len(lines) == 1 and lines[0]["phase"] == "start" and lines[0]["reported_by"] == "launcher"
    assert time.time() - t0 < 60
