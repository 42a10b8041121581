#!/usr/bin/env python3
"""The tiled sweep's speed is the latency of ONE wave's instruction stream, which makes it sensitive to code generation:
a load nobody consumed once left the register allocator free to reuse its destination, and the hazard logic answered with
`s_waitcnt vmcnt(0)` -- a wait for the previous step's global stores -- at the top of every step (Anderson-256 0.85 -> 1.09 ms,
DESIGN.md section 4, third pass).  This compiles bis_trsv_tiled.hip to gfx950 assembly and reports, for the production
kernels, the `vmcnt(0)` waits inside each of the compute wave's step loops.
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
    """per step loop of the compute wave (the depth-2 loops that contain the division): (has a descriptor prefetch, number
    of `vmcnt(0)` waits in the blocks that do a step's work).  The general loop fetches the next batch of step descriptors
    every 64 steps and waits for it in a block of its own; no block on a step's path may wait for memory."""
    loops = []
    for k, l in enumerate(body):
        if "v_div_fixup_f64" not in l: continue
        # the block's own label names the loop it belongs to
        h = next((re.search(r"Header=(\w+) Depth=2", body[m]) for m in range(k, 0, -1) if body[m].startswith(".LBB")), None)
        if not h: continue
        header = h.group(1)
        if any(x[0] == header for x in loops): continue
        # a wait counts against the loop when the block it sits in does a step's work (descriptor read-lanes, LDS reads, the
        # division): the batch boundary of the general loop has a block of its own (the descriptor load and the wait for it)
        waits = loads = 0
        inside = False
        blk_wait = blk_step = 0
        for l2 in body + [".LBB_end:"]:
            if l2.startswith(".LBB"):
                if inside and blk_step: waits += blk_wait
                blk_wait = blk_step = 0
                inside = (l2.split(":")[0] == "." + "L" + header) or (f"Header={header} " in l2)
            if inside:
                blk_wait += bool(re.search(r"s_waitcnt.*vmcnt\(0\)", l2))
                blk_step += bool(re.search(r"v_readlane_b32|ds_read_b|v_div_|v_fma_f64|global_store", l2))
                loads += "global_load_dwordx4" in l2
        loops.append((header, loads > 0, waits))
    return loops


def selftest():
    """the pattern this guards against (as it was in the build that had it), and the legitimate one"""
    bad = ['.LBB1_300:                            ;   Parent Loop BB1_8 Depth=1', '                                        ; =>  This Loop Header: Depth=2',
           '\tv_readlane_b32 s11, v7, s52', '.LBB1_301:                            ;   in Loop: Header=BB1_300 Depth=2',
           '\ts_waitcnt vmcnt(0) lgkmcnt(6)', '\tds_read_b128 v[26:29], v10 offset:16432',
           '.LBB1_303:                            ;   in Loop: Header=BB1_300 Depth=2', '\tv_div_fixup_f64 v[10:11], v[14:15], v[12:13], v[10:11]']
    good = bad[:4] + ['\tds_read_b128 v[26:29], v10 offset:16432'] + bad[6:] + [
           '.LBB1_341:                            ;   in Loop: Header=BB1_300 Depth=2', '\ts_waitcnt vmcnt(0)',
           '\tglobal_load_dwordx4 v[10:13], v[10:11], off']
    assert [w for _, _, w in check(bad)] == [1], check(bad)
    assert [w for _, _, w in check(good)] == [0], check(good)


def main():
    selftest()
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
        loops = check(body)
        # rows of 1, 2 and 4 quads and the general loop: fewer means the pattern match has gone stale (a compiler update changed the
        # mangled name or the loop comments), not that the kernel is fine
        if len(loops) < 4:
            print(f"{name[40:90]}...: only {len(loops)} step loops recognised (expected 4): the guard no longer sees the kernel's loops")
            rc = 1
        for header, general, waits in loops:
            ok = waits == 0
            print(f"{name[40:90]}... step loop {header} ({'general' if general else 'per row length'}): {waits} vmcnt(0) waits{'' if ok else '  <-- waits for its stores'}")
            if not ok: rc = 1
    return rc


if __name__ == "__main__":
    sys.exit(main())
