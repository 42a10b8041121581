"""Build the gfx950 C-ABI library (libbis_hip.so) in-tree with hipcc.

    python -m basic_iterative_solvers_amd.build

hipcc cross-compiles for gfx950 without a GPU.  Objects are cached under
basic_iterative_solvers_amd/build/ and rebuilt when a source or header is
newer.  The .so stays in-tree (git-ignored) so it travels with gpurun.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
OBJ = os.path.join(PKG, "build")
LIB = os.path.join(PKG, "lib", "libbis_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"

FLAGS = ["-std=c++17", "-O3", f"--offload-arch={ARCH}", "-fPIC", "-fvisibility=hidden",
         "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result", "-I", os.path.join(ROOT, "include"), "-I", CSRC]
# e.g. BIS_EXTRA_HIPCC_FLAGS=-DBIS_TILED_EXP: the tiled sweep's timing-experiment / stamped builds (tools/trsv_ab.py, tools/trsv_tile_debug.py)
FLAGS += os.environ.get("BIS_EXTRA_HIPCC_FLAGS", "").split()


def _newer(src, dst):
    return (not os.path.exists(dst)) or os.path.getmtime(src) > os.path.getmtime(dst)


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


def build(verbose=False, force=False):
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    headers.append(os.path.join(ROOT, "include", "bis_hip.h"))
    hdr_time = max(os.path.getmtime(h) for h in headers)
    jobs = []
    objs = []
    for f in sources():
        src = os.path.join(CSRC, f)
        obj = os.path.join(OBJ, f + ".o")
        objs.append(obj)
        if force or _newer(src, obj) or os.path.getmtime(obj) < hdr_time:
            jobs.append([HIPCC, *FLAGS, "-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if jobs or not os.path.exists(LIB):
        run([HIPCC, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB, *objs, "-ldl"])
    return LIB


if __name__ == "__main__":
    print(build(verbose=True, force="--force" in sys.argv))
