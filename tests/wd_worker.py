"""Stand-in for `torch.distributed.run` + bench ranks in the watchdog tests (no torch, no GPU).

    wd_worker.py --spawn N [--mode ok|hang-import]    the launcher: starts N ranks, stops the rest when one fails
    wd_worker.py --rank R --world N [--mode ...]      a rank: the phases of dist_bench.py with file-based "collectives"
"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def arg(name, default=None):
    return sys.argv[sys.argv.index(name) + 1] if name in sys.argv else default


def launcher(n, mode):
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--rank", str(r), "--world", str(n), "--mode", mode])
             for r in range(n)]
    rc = 0
    while procs:
        time.sleep(0.1)
        for p in list(procs):
            c = p.poll()
            if c is None:
                continue
            procs.remove(p)
            if c != 0 and rc == 0:  # like torch.distributed.run: one failed worker stops the others
                rc = c
                for q in procs:
                    q.terminate()
    return rc


def rank_main(rank, world, mode):
    if mode == "hang-import" and rank == 1:
        time.sleep(600)  # an interpreter that never gets as far as its watchdog
    from basic_iterative_solvers_amd.watchdog import RankWatchdog, read_board
    wd = RankWatchdog(rank, world, out_fd=1)

    def collective(min_seq):  # returns once every rank has passed checkpoint `min_seq`
        while True:
            b = read_board(wd.dir, world)
            if all(v[2] >= min_seq for v in b.values()):
                return
            time.sleep(0.05)

    wd.enter("gen")
    wd.tick("slab generated")
    collective(wd.seq)
    wd.enter("warmup")
    collective(wd.seq)
    wd.enter("timed")
    collective(wd.seq)
    wd.enter("report")
    if rank == 0:
        print('{"value": 1.0}', flush=True)
    wd.enter("done")
    collective(wd.seq)
    wd.stop()


if __name__ == "__main__":
    mode = arg("--mode", "ok")
    if "--spawn" in sys.argv:
        raise SystemExit(launcher(int(arg("--spawn")), mode))
    rank_main(int(arg("--rank")), int(arg("--world")), mode)
