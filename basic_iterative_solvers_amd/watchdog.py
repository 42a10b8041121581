"""Watchdog of the multi-rank bench: a stuck rank must end in ONE diagnostic JSON line within a limit, never in a
run that is killed at the driver's limit with nothing written.

Three pieces, all plain files and threads (no GPU, no torch):

* a PHASE BOARD: a directory in which every rank keeps one file with its current phase
  (`init`, `rccl-id`, `comm-init`, `gen`, `warmup`, `timed`, `512-leg`, `report`, `done`) and a CHECKPOINT NUMBER that grows
  with every phase change and every `tick` inside a phase.  All ranks run the same sequence of checkpoints, and a rank that
  waits for another one in a collective has passed every checkpoint before that collective: the rank with the LOWEST
  number is the one the others wait for.  A phase change is also a line on stderr (`[bis-phase] rank=R phase=NAME +S.Ss`),
  so a log that was cut off still says where every rank was;
* a RANK WATCHDOG (a daemon thread in every rank, `RankWatchdog`): when the rank's current phase outlives its limit it reads
  the board, names the rank that is furthest behind (the one the others wait for), prints ONE JSON line
  `{"error": ..., "phase": ..., "rank": ..., "phases": {...}}` to the bench's stdout and ends the process with status 3 --
  torch.distributed.run then stops the other ranks.  The limits are staggered by rank (rank 0 first), so exactly one
  rank reports.  This works under ANY launcher (the driver starts the ranks itself for N > 1);
* a PARENT WATCHDOG (`supervise`): `python bench.py --gpus N` starts its ranks as a child process group and watches the
  same board with a longer fuse; it covers ranks whose interpreter itself hangs (import, a C call that never returns):
  the children are killed by the process group that was started here -- never by pattern --, the JSON line is printed,
  the exit status is non-zero.  A restart, if anyone wants one, is a fresh child: nothing here re-executes a process that
  has touched the GPU.

Environment: BIS_PHASE_DIR (the board; made by `supervise`, else /tmp/bis_phases_<MASTER_PORT>_<launcher pid>), BIS_PHASE_LIMIT_S (one
limit for every phase: tests), BIS_BENCH_STUCK="rank:phase:seconds" (test hook: that rank sleeps when it enters the phase).
"""
import json
import os
import signal
import subprocess
import sys
import threading
import time

PHASES = ("start", "init", "rccl-id", "comm-init", "gen", "warmup", "timed", "512-leg", "report", "done")
# seconds a rank may stay in a phase.  `start` / `init` cover the first `import torch` on a fresh box (1-2 minutes while the
# image pages in) and the process-group rendezvous; `gen` the generation of a slab and the distributed plan; `512-leg` a
# whole leg on the north-star problem.
LIMITS = {"start": 420.0, "init": 300.0, "rccl-id": 90.0, "comm-init": 180.0, "gen": 240.0, "warmup": 180.0, "timed": 300.0,
          "512-leg": 420.0, "report": 120.0, "done": 60.0}
RANK_STAGGER_S = 5.0     # rank r's fuse is r * this longer than rank 0's: one reporter
PARENT_EXTRA_S = 60.0    # the parent's fuse on top of the slowest rank's


def limit_of(phase):
    if os.environ.get("BIS_PHASE_LIMIT_S"):
        return float(os.environ["BIS_PHASE_LIMIT_S"])
    return LIMITS.get(phase, 300.0)


def board_dir():
    d = os.environ.get("BIS_PHASE_DIR")
    if not d:
        # the ranks of one launch are children of one launcher process: its pid (and the rendezvous port) name a board that no
        # earlier launch has written to
        d = os.path.join("/tmp", "bis_phases_%s_%d" % (os.environ.get("MASTER_PORT", "0"), os.getppid()))
    os.makedirs(d, exist_ok=True)
    return d


def read_board(d, world=None):
    """{rank: (phase, seconds since the rank's last checkpoint, checkpoint number, label)}; ranks that have not written yet
    are in `start` at checkpoint 0."""
    out, now = {}, time.time()
    try:
        names = os.listdir(d)
    except OSError:
        names = []
    for f in names:
        if not f.startswith("rank_"):
            continue
        try:
            rank = int(f[5:])
            if world and rank >= world:
                continue  # (a file an earlier, larger run left on a board that is found by the rendezvous port)
            phase, seq, t, label = (open(os.path.join(d, f)).read().split(None, 3) + [""])[:4]
            out[rank] = (phase, now - float(t), int(seq), label.strip())
        except Exception:
            pass
    if world:
        t0 = now
        try:
            t0 = float(open(os.path.join(d, "t0")).read())
        except Exception:
            pass
        for r in range(world):
            out.setdefault(r, ("start", now - t0, 0, ""))
    return out


def furthest_behind(board):
    """(rank, phase, seconds since its last checkpoint) of the rank the others wait for: the lowest checkpoint number, the
    longest stay."""
    r, v = min(board.items(), key=lambda item: (item[1][2], -item[1][1]))
    return r, v[0], v[1]


def diagnostic(board, world, why):
    r, ph, dt = furthest_behind(board) if board else (None, "start", 0.0)
    return {"error": f"{why}: rank {r} has been in phase '{ph}' for {dt:.0f} s (limit {limit_of(ph):.0f} s); "
                     "the other ranks wait for it in a collective",
            "phase": ph, "rank": r, "seconds_in_phase": round(dt, 1), "n_gpus": world,
            "checkpoint": (board[r][3] if board and r is not None else ""),
            "phases": {str(k): f"{v[0]} #{v[2]} {v[3]}".strip() for k, v in sorted(board.items())}}


class RankWatchdog:
    """One per rank.  enter(phase) moves this rank on the board; a daemon thread ends the process when the rank's current
    phase has outlived its limit (+ the rank's stagger)."""

    def __init__(self, rank, world, out_fd=1):
        self.rank, self.world, self.out_fd = rank, world, out_fd
        self.dir = board_dir()
        self.t_start = time.time()
        self.phase, self.t_phase, self.seq = "start", self.t_start, 0
        self._lock = threading.Lock()
        self._stop = False
        self.enter("init")
        self._thr = threading.Thread(target=self._run, daemon=True)
        self._thr.start()

    def _post(self, phase, label, now):
        tmp = os.path.join(self.dir, f".rank_{self.rank}.tmp")
        with open(tmp, "w") as f:
            f.write(f"{phase} {self.seq} {now!r} {label}")
        os.replace(tmp, os.path.join(self.dir, f"rank_{self.rank}"))

    def tick(self, label=""):
        """a checkpoint inside the current phase (between two collectives): moves this rank's number, not its phase timer"""
        now = time.time()
        with self._lock:
            self.seq += 1
        self._post(self.phase, label, now)

    def enter(self, phase):
        now = time.time()
        with self._lock:
            self.phase, self.t_phase = phase, now
            self.seq += 1
        self._post(phase, "", now)
        print(f"[bis-phase] rank={self.rank} phase={phase} +{now - self.t_start:.1f}s", file=sys.stderr, flush=True)
        hook = os.environ.get("BIS_BENCH_STUCK")  # test hook: "rank:phase:seconds"
        if hook:
            try:
                r, ph, secs = hook.split(":")
                if int(r) == self.rank and ph == phase:
                    time.sleep(float(secs))
            except ValueError:
                pass

    def stop(self):
        self._stop = True

    def _run(self):
        while not self._stop:
            time.sleep(0.5)
            with self._lock:
                phase, t_phase = self.phase, self.t_phase
            if phase == "done":
                return
            if time.time() - t_phase > limit_of(phase) + RANK_STAGGER_S * self.rank:
                d = diagnostic(read_board(self.dir, self.world), self.world, "a rank is stuck")
                d["reported_by"] = f"rank {self.rank}"
                line = json.dumps(d)
                print("[bis-watchdog] " + line, file=sys.stderr, flush=True)
                try:
                    os.write(self.out_fd, (line + "\n").encode())
                except OSError:
                    pass
                os._exit(3)  # torch.distributed.run stops the other ranks when one ends with an error


def supervise(cmd, world, env=None):
    """Run `cmd` (the launcher and its ranks) as a child PROCESS GROUP, relay its stdout / stderr, and watch the phase board:
    returns the child's exit status, or -- when a rank outlives its phase limit and no rank reports it -- kills the group,
    prints the diagnostic JSON line and returns 3."""
    import tempfile
    env = dict(os.environ if env is None else env)
    d = tempfile.mkdtemp(prefix="bis_phases_")
    env["BIS_PHASE_DIR"] = d
    with open(os.path.join(d, "t0"), "w") as f:
        f.write(repr(time.time()))
    child = subprocess.Popen(cmd, env=env, start_new_session=True)  # its own process group: what we kill is what we started
    try:
        while True:
            try:
                return child.wait(timeout=1.0)
            except subprocess.TimeoutExpired:
                pass
            board = read_board(d, world)
            r, ph, dt = furthest_behind(board)
            if ph != "done" and dt > limit_of(ph) + RANK_STAGGER_S * world + PARENT_EXTRA_S * (0.1 if os.environ.get("BIS_PHASE_LIMIT_S") else 1.0):
                diag = diagnostic(board, world, "a rank is stuck and did not report it itself")
                diag["reported_by"] = "launcher"
                try:
                    os.killpg(child.pid, signal.SIGTERM)
                    try:
                        child.wait(timeout=10)
                    except subprocess.TimeoutExpired:
                        os.killpg(child.pid, signal.SIGKILL)
                        child.wait(timeout=10)
                except (ProcessLookupError, subprocess.TimeoutExpired):
                    pass
                print(json.dumps(diag), flush=True)
                return 3
    finally:
        if child.poll() is None:
            try:
                os.killpg(child.pid, signal.SIGKILL)
            except ProcessLookupError:
                pass
        try:
            for f in os.listdir(d):
                os.remove(os.path.join(d, f))
            os.rmdir(d)
        except OSError:
            pass
