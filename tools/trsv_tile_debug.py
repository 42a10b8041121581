#!/usr/bin/env python3
"""Per-tile cycle stamps of the tiled sweep (BIS_TRSV_TILE_DEBUG): where a tile's time goes.
   python tools/trsv_tile_debug.py anderson 256 [rows]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
F = "/tmp/tile_dbg.bin"
os.environ["BIS_TRSV_TILE_DEBUG"] = F
import numpy as np
from basic_iterative_solvers_amd import Context
kind, n1 = sys.argv[1], int(sys.argv[2])
ctx = Context(0)
if len(sys.argv) > 3: ctx.set_option("trsv_tile_rows", int(sys.argv[3]))
if len(sys.argv) > 4: ctx.set_option("trsv_tile_edge", int(sys.argv[4]))
ctx.set_option("trsv_tiled", int(os.environ.get("TILED_MODE", "-1")))
A = ctx.gen_hpcg(n1) if kind == "hpcg" else ctx.gen_fem(n1) if kind == "fem" else ctx.gen_anderson(n1, shift=9.0)
Ls, Us, D, Dinv = ctx.split_strict(A)
N = A.n_rows
b, x = ctx.alloc(N), ctx.alloc(N)
ctx.init_vector(b, 1.0)
for _ in range(2):
    ctx.sptrsv(Ls, x, D, b); ctx.sync()
raw = np.fromfile(F, dtype=np.int64, count=3)
n_t, n_s, n_e = (int(v) for v in raw)
body = np.fromfile(F, dtype=np.int64, offset=24, count=16 * n_t + n_s + n_e)
d = body[:16 * n_t].reshape(-1, 16)
pub, dlv = body[16 * n_t:16 * n_t + n_s], body[16 * n_t + n_s:]
off = 24 + 8 * (16 * n_t + n_s + n_e)
src = np.fromfile(F, dtype=np.int32, offset=off, count=n_e)
ext0 = np.fromfile(F, dtype=np.int64, offset=off + 4 * n_e, count=n_t + 1)
slot0 = np.fromfile(F, dtype=np.int64, offset=off + 4 * n_e + 8 * (n_t + 1), count=n_t + 1)
t0 = d[:, 0].min()
f = 100.0  # s_memrealtime ticks per us (100 MHz); the wait counters are core cycles (~2400 per us)
start, end = (d[:, 0] - t0) / f, (d[:, 1] - t0) / f
print(f"tiles {len(d)}  sweep {end.max():.1f} us (first start -> last end)")
dur = end - start
steps, loop_cyc = d[:, 7] & 0xffff, d[:, 7] >> 16
print(f"tile duration us: mean {dur.mean():.1f} median {np.median(dur):.1f} max {dur.max():.1f}; steps/tile mean {steps.mean():.1f}")
work = loop_cyc - d[:, 2] - d[:, 3]
print(f"step loop: mean {loop_cyc.mean()/2400:.1f} us/tile, of it neither waiting for loaders nor for operands {work.mean()/2400:.1f} us = {work.sum()/steps.sum():.0f} core cycles per step")
print(f"compute wave waited for loaders: mean {d[:,2].mean()/2400:.1f} us/tile; for external operands: mean {d[:,3].mean()/2400:.1f} us/tile")
for nm, c in (("entry loader", 4), ("slot loader", 5), ("poller", 6)):
    print(f"{nm} finished after (from tile start) mean {((d[:,c]-d[:,0])/f).mean():.1f} us")
k = np.arange(len(d))
for i in (0, 1, 2, 3, len(d)//2, len(d)//2+1, len(d)-1):
    print(f"tile {i}: start {start[i]:.1f} end {end[i]:.1f} wait_load {d[i,2]/2400:.1f} wait_ext {d[i,3]/2400:.1f}")
conc = [(np.sum((start <= t) & (end > t))) for t in np.linspace(0, end.max(), 21)[1:-1]]
print("tiles in flight at 5%..95% of the sweep:", conc)

# hand-off through memory: delivery stamp of an external ordinal minus publish stamp of the slot it stands for (100 MHz ticks)
lat = (dlv - pub[src]) / f
ok = (dlv > 0) & (pub[src] > 0)
fresh = ok & (lat < 20.0)
q = np.percentile(lat[fresh], [1, 5, 25, 50, 75, 95])
print(f"publish -> delivery, ordinals delivered within 20 us of their publication ({fresh.sum()} of {ok.sum()}): "
      f"1% {q[0]:.2f}  5% {q[1]:.2f}  25% {q[2]:.2f}  median {q[3]:.2f}  75% {q[4]:.2f}  95% {q[5]:.2f} us; negative (stamp order) {np.sum(lat[ok] < 0)}")
rounds, rc = d[:, 8], d[:, 9]
m = rounds > 0
print(f"poller: {rounds[m].mean():.1f} rounds per tile, {rc[m].sum() / rounds[m].sum():.0f} core cycles per round")
# per tile: the last delivery before the tile's first publication -> that publication (wake-up of the compute wave + its first step)
first_pub = np.array([pub[slot0[t]:slot0[t + 1]].min() for t in range(n_t)])
gap = []
for t in range(0, n_t, max(1, n_t // 2000)):
    e = dlv[ext0[t]:ext0[t + 1]]
    e = e[(e > 0) & (e <= first_pub[t])]
    if len(e): gap.append((first_pub[t] - e.max()) / f)
gap = np.array(gap)
print(f"last delivery before a tile's first result -> that result: median {np.median(gap):.2f} us, 25% {np.percentile(gap, 25):.2f}, 75% {np.percentile(gap, 75):.2f}")
# the critical chain: for every tile the latest delivery at all vs the tile's last publication
last_pub = np.array([pub[slot0[t]:slot0[t + 1]].max() for t in range(n_t)])
print(f"tile: first result -> last result median {np.median((last_pub - first_pub) / f):.2f} us")

# the critical chain, walked back from the last tile: the operand a tile's first result waited for (the latest delivery
# before it) -> the tile that produced it -> ...; per hop: steps inside the producer (its first result -> the operand's
# publication), memory hand-off (publication -> delivery), wake-up + first step (delivery -> the consumer's first result)
tile_of = np.searchsorted(slot0, np.arange(n_s - 1), side="right") - 1
t = int(np.argmax(last_pub))
hops = []
while True:
    e = dlv[ext0[t]:ext0[t + 1]]
    ok_e = (e > 0) & (e <= first_pub[t])
    if not ok_e.any(): break
    k = int(np.argmax(np.where(ok_e, e, 0)))
    s_src = int(src[ext0[t] + k])
    tp = int(tile_of[s_src])
    hops.append(((pub[s_src] - first_pub[tp]) / f, (e[k] - pub[s_src]) / f, (first_pub[t] - e[k]) / f))
    t = tp
h = np.array(hops)
if len(h):
    print(f"critical chain: {len(h)} hops; per hop mean: in-producer first result -> operand published {h[:,0].mean():.2f} us, "
          f"published -> delivered {h[:,1].mean():.2f} us (median {np.median(h[:,1]):.2f}), delivered -> consumer's first result {h[:,2].mean():.2f} us; "
          f"sum {h.sum():.0f} us of the sweep")

# placement: which SIMD the four roles of a workgroup sit on, and how many compute waves share a SIMD of a CU at a time
hw = d[:, 10:14]
simd = (hw >> 4) & 3
print("SIMD of (compute, entry loader, slot loader, poller), share of tiles:",
      [np.round(np.bincount(simd[:, w], minlength=4) / len(d), 2).tolist() for w in range(4)])
slot = hw[:, 0] & 15
print("wave slot of the compute wave:", np.bincount(slot, minlength=10).tolist())
cu = ((hw[:, 0] >> 8) & 0xff) | ((hw[:, 0] >> 32) << 8) | (((hw[:, 0] >> 13) & 7) << 12)
print("distinct (XCD, SE, CU) of compute waves:", len(np.unique(cu)))
# does a step cost more when the chip is full?  step cycles (waits excluded) by the number of tiles in flight at the tile's start
order = np.argsort(start)
inflight = np.array([np.sum((start <= start[i]) & (end > start[i])) for i in order[:: max(1, len(order) // 400)]])
samp = order[:: max(1, len(order) // 400)]
for lo, hi in ((0, 64), (64, 256), (256, 768), (768, 2048)):
    m = (inflight >= lo) & (inflight < hi)
    if m.any(): print(f"tiles started with {lo}..{hi} tiles in flight: {work[samp][m].sum() / steps[samp][m].sum():.0f} core cycles per step ({m.sum()} sampled)")
# one tile in the middle of the sweep, in detail: when each of its steps published, and per block of 8 external ordinals
# (first-need order) when their producers published them and when the poller delivered them (us relative to the tile's first result)
for t in [int(v) for v in os.environ.get('TILES', str(n_t // 2)).split(',')]:
  sp = np.unique(pub[slot0[t]:slot0[t + 1]])
  print(f"tile {t}: {len(sp)} steps published at", np.round((sp - first_pub[t]) / f, 2).tolist())
  e_pub = (pub[src[ext0[t]:ext0[t + 1]]] - first_pub[t]) / f
  e_dlv = (dlv[ext0[t]:ext0[t + 1]] - first_pub[t]) / f
  print("ordinal: producer published / delivered (us rel. to the tile's first result)")
  for i in range(0, len(e_pub), 8):
      print(f"  {i:4d}: " + "  ".join(f"{a:6.2f}/{b:6.2f}" for a, b in zip(e_pub[i:i + 8], e_dlv[i:i + 8])))

# inside the steps of the middle tile (core cycles): loop top -> operands ready checked -> codes/values/row read -> operands read
# -> fma chain done -> division done -> stores issued; and from the stores to the next loop top
st = np.fromfile(F, dtype=np.int64, offset=off + 4 * n_e + 16 * (n_t + 1), count=512).reshape(64, 8)
st = st[st[:, 0] > 0]
names = ["checks", "codes+row", "operands", "fma", "division", "stores"]
for k in range(len(st)):
    seg = [int(st[k, i + 1] - st[k, i]) for i in range(6)]
    nxt = int(st[k + 1, 0] - st[k, 6]) if k + 1 < len(st) else 0
    print(f"  step {k:2d}: " + "  ".join(f"{n} {v:5d}" for n, v in zip(names, seg)) + f"  to next {nxt:5d}")

