#!/usr/bin/env python3
"""The tiled sweep's speed is the latency of ONE wave's instruction stream, which makes it sensitive to code generation:
a load nobody consumed once left the register allocator free to reuse its destination, and the hazard logic answered with
`s_waitcnt vmcnt(0)` -- a wait for the previous step's global stores -- at the top of every step (Anderson-256 0.85 -> 1.09 ms,
DESIGN.md section 4, third pass).  This compiles bis_trsv_tiled.hip to gfx950 assembly and reports, for the production
kernels, the `vmcnt(0)` waits inside the compute wave's code.
   python tools/check_tiled_isa.py            (exit code 1 if a step loop waits for its stores)"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from basic_iterative_solvers_amd import build


def kernels(asm):
    """name -> list of lines, for the production (DBG = false, EXP = false) instantiations of trsv_tiled_kernel"""
    out, lines = {}, asm.split("\n")
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\S*trsv_tiled_kernel\S*Lb0ELb0E\S*):", l)
        if m:
            j = i
            while not lines[j].startswith(".Lfunc_end"): j += 1
            out[m.group(1)] = lines[i:j]
    return out


def check(body):
    """the `vmcnt(0)` waits in the compute wave's code.  Three are expected: the tile's first descriptors, and the batch
    boundary of the general step loop (the wait before the next batch is taken, and the loop's entry); a step loop that
    waits for its stores shows up as more (the regression above: eight)."""
    start = next(i for i, l in enumerate(body) if "s_setprio 3" in l)
    # the compute wave's code ends where the next role's begins: the quad loader is the first code after it that loads
    # 16-byte values non-temporally
    end = next((i for i in range(start, len(body)) if "global_load_dwordx4" in body[i] and " nt" in body[i]), len(body))
    return [i for i in range(start, end) if re.search(r"s_waitcnt.*vmcnt\(0\)", body[i])]


kExpected = 4  # (one spare: the count has been 3 since the per-length instances exist)


def main():
    src = os.path.join(build.CSRC, "bis_trsv_tiled.hip")
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "tiled.s")
        flags = [f for f in build.FLAGS if f != "-fPIC"]
        subprocess.check_call([build.HIPCC, *flags, "-w", "-S", "--cuda-device-only", "-o", out, src])
        asm = open(out).read()
    ks = kernels(asm)
    assert ks, "no production instantiation of trsv_tiled_kernel found"
    rc = 0
    for name, body in ks.items():
        waits = check(body)
        print(f"{name[:70]}...: {len(waits)} vmcnt(0) waits in the compute wave's code (expected at most {kExpected})")
        if len(waits) > kExpected: rc = 1
    return rc


if __name__ == "__main__":
    sys.exit(main())
