#!/usr/bin/env python3
"""When is the machine busy?  Per-wave start / end times of one frame (diagnostic library: make STAMPS=1 SPILLS_OK=1, run with
MI355RT_DEBUG_COUNTERS=1): waves alive per microsecond, split by what they end up doing, and the phase shares.
usage: python tools/timeline.py [W H [flags [orbit pose 0..23]]]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["MI355RT_DEBUG_COUNTERS"] = "1"
os.environ["MI355RT_ALLOW_DIAGNOSTIC"] = "1"
import __graft_entry__ as graft  # noqa: E402

pkg = graft.load_package()
W = int(sys.argv[1]) if len(sys.argv) > 1 else 1920
H = int(sys.argv[2]) if len(sys.argv) > 2 else 1080
flags = int(sys.argv[3]) if len(sys.argv) > 3 else 0
sc = pkg.Scene.load_from_file(os.path.join(ROOT, "scenes", "20spheres.yml")).set_size(W, H)
r = pkg.Renderer(sc, device=0, flags=flags)
cam = None
if len(sys.argv) > 4:   # a pose of bench.py's orbit
    a = 2.0 * np.pi * int(sys.argv[4]) / 24
    pos = (5.0 + 14.0 * np.sin(a), 2.0 + 2.0 * np.sin(2 * a), 15.0 - 14.0 * np.cos(a))
    cam = pkg.camera_matrix(pos, float(np.degrees(np.arctan2(15.0 - pos[2], 5.0 - pos[0]))), float(-np.degrees(np.arctan2(pos[1] - 2.0, 14.0))))
    print(f"orbit pose {sys.argv[4]}")
for _ in range(4):
    ms = r.update(cam)
rows = r.stamp_rows()
row_index = np.nonzero(rows[:, 13] > 0)[0]
rows = rows[rows[:, 13] > 0]
t0 = rows[:, 12].min()
start = (rows[:, 12] - t0).astype(np.float64) / 100.0   # 100 MHz -> us
end = (rows[:, 13] - t0).astype(np.float64) / 100.0
traced = rows[:, 6] > 0          # waves that went through the shadow phase
other = ~traced
print(f"20spheres {W}x{H} flags={flags}: kernel {ms*1e3:.1f} us (stamped build); {len(rows)} waves wrote a row, {int(traced.sum())} of them traced shadow rays")
print(f"last wave ends at {end.max():.1f} us; tracing waves: start median {np.median(start[traced]):.1f} us (90 %: {np.percentile(start[traced], 90):.1f}), "
      f"lifetime median {np.median((end - start)[traced]):.1f} us, max {(end - start)[traced].max():.1f} us")
step = max(1.0, round(end.max() / 30.0))
print(f"{'t (us)':>8} {'tracing':>8} {'other':>8}   waves alive (5120 slots)")
for t in np.arange(0.0, end.max() + step, step):
    a = int(((start <= t) & (end > t) & traced).sum())
    b = int(((start <= t) & (end > t) & other).sum())
    print(f"{t:8.1f} {a:8d} {b:8d}   {'#' * (a // 100)}{'.' * (b // 100)}")
names = ["stage scene->LDS", "setup/primary dir", "A nearest", "A normal+compact", "barrier after A", "A' balls + barrier",
         "B shadow items", "barrier after B", "C shade", "D blend + barrier", "store", "classify / paint"]
if os.environ.get("TIMELINE_LEAN_NAMES"):   # the lean instantiation stamps its set-up in pieces (slots 3, 4, 7, 8)
    names[3], names[4], names[7], names[8] = "setup: args + launch-order decode", "setup: decode barrier", "setup: tile -> staging written", "setup: staging barrier"
    names[1] = "setup: camera tables + primary dir"
tot = rows[:, :12].sum()
for i, n in enumerate(names):
    print(f"  {n:22s} {int(rows[:, i].sum()):14d}  {100.0 * rows[:, i].sum() / max(tot, 1):5.1f} %")
# per wave of the tracing workgroups: where a wave's own time goes (us at 2.4 GHz), median / 90 % / max over the waves that traced shadow rays
for i, n in enumerate(names):
    v = rows[traced, i].astype(np.float64) / 2400.0
    if v.max() > 0:
        print(f"  per tracing wave  {n:22s} median {np.median(v):6.2f}  90 % {np.percentile(v, 90):6.2f}  max {v.max():6.2f} us")
# per workgroup: lifetime against the tile's hit count
info = rows[:, 14]
wg = {}
for i in np.nonzero(traced)[0]:
    key = int(info[i] >> np.uint64(32))
    e = wg.setdefault(key, dict(hits=int(info[i] & np.uint64(0xFFFFFFFF)), start=start[i], end=end[i], b=0, a=0, setup=0, c=0))
    e["start"], e["end"] = min(e["start"], start[i]), max(e["end"], end[i])
    e["b"] = max(e["b"], int(rows[i, 6])); e["a"] = max(e["a"], int(rows[i, 2])); e["setup"] = max(e["setup"], int(rows[i, 1])); e["c"] = max(e["c"], int(rows[i, 8]))
life = np.array([v["end"] - v["start"] for v in wg.values()])
hits = np.array([v["hits"] for v in wg.values()])
bcyc = np.array([v["b"] for v in wg.values()], dtype=np.float64)
print(f"{len(wg)} tracing workgroups; lifetime by round-0 hits:")
for lo, hi in ((1, 64), (65, 128), (129, 192), (193, 255), (256, 256)):
    m = (hits >= lo) & (hits <= hi)
    if m.any():
        print(f"  hits {lo:3d}-{hi:3d}: {int(m.sum()):5d} tiles, lifetime median {np.median(life[m]):5.1f} us, 90 % {np.percentile(life[m], 90):5.1f}, max {life[m].max():5.1f};"
              f" longest wave's shadow phase median {np.median(bcyc[m]) / 2400.0:5.1f} us (at 2.4 GHz), max {bcyc[m].max() / 2400.0:5.1f}")
order = np.argsort(-life)[:12]
keys = list(wg.keys())
print("longest tiles: tile, hits, lifetime us, setup / A / B / C of its slowest wave (us at 2.4 GHz)")
for j in order:
    v = wg[keys[j]]
    print(f"  tile {keys[j]:5d} hits {v['hits']:3d} life {life[j]:5.1f}  setup {v['setup']/2400.0:4.1f} A {v['a']/2400.0:4.1f} B {v['b']/2400.0:5.1f} C {v['c']/2400.0:4.1f}")
# per CU: how much shadow-phase work landed there and when its last tracing wave ended (HW_ID: cu [11:8], sh [12], se [15:13]; XCC_ID [3:0])
hw = rows[:, 15]
xcc = (hw >> np.uint64(32)) & np.uint64(0xF)
hwid = hw & np.uint64(0xFFFFFFFF)
cu = ((xcc << np.uint64(8)) | (((hwid >> np.uint64(13)) & np.uint64(7)) << np.uint64(5)) | (((hwid >> np.uint64(12)) & np.uint64(1)) << np.uint64(4)) | ((hwid >> np.uint64(8)) & np.uint64(0xF))).astype(np.int64)
cus = np.unique(cu[traced])
work = np.array([rows[traced & (cu == c), 6].sum() / 2400.0 for c in cus])      # wave-us of shadow phase
nw = np.array([int((traced & (cu == c)).sum()) for c in cus])
last = np.array([end[traced & (cu == c)].max() for c in cus])
print(f"{len(cus)} CUs ran tracing waves: waves per CU min {nw.min()} median {int(np.median(nw))} max {nw.max()}; shadow-phase wave-us per CU min {work.min():.0f} median {np.median(work):.0f} max {work.max():.0f};"
      f" last tracing wave ends: min {last.min():.1f} median {np.median(last):.1f} max {last.max():.1f} us; correlation(work, end) {np.corrcoef(work, last)[0, 1]:.2f}")
# How well could the shadow-phase work be spread?  Per-SIMD sums of the tracing waves' shadow-phase cycles as they ran (HW_ID simd_id [5:4]),
# against longest-first assignments of the same waves / of whole workgroups to 1024 SIMDs / 256 CUs (the measured cycles include the
# slow-down of a crowded SIMD, so this is indicative only).
blocks = row_index // 4
simd = ((hwid >> np.uint64(4)) & np.uint64(3)).astype(np.int64)
key = cu[traced] * 4 + simd[traced]
bw = rows[traced, 6].astype(np.float64) / 2400.0
loads = {}
for k, v in zip(key, bw):
    loads[int(k)] = loads.get(int(k), 0.0) + v
ld = np.array(list(loads.values()) + [0.0] * max(0, 1024 - len(loads)))
def lpt(items, bins):
    import heapq
    h = [0.0] * bins
    heapq.heapify(h)
    for v in sorted(items, reverse=True):
        heapq.heappush(h, heapq.heappop(h) + v)
    return max(h)
wg_b = {}
for b, v in zip(blocks[traced], bw):
    wg_b[int(b)] = wg_b.get(int(b), 0.0) + v
print(f"shadow-phase wave-us per SIMD as run: mean {bw.sum() / 1024:.1f} max {ld.max():.1f} (max / mean {ld.max() / (bw.sum() / 1024):.2f}); "
      f"longest-first by wave over 1024 SIMDs: max {lpt(bw, 1024):.1f}; by workgroup over 256 CUs (per SIMD): max {lpt(wg_b.values(), 256) / 4:.1f}; "
      f"largest single wave {bw.max():.1f}, largest workgroup / 4 {max(wg_b.values()) / 4:.1f}")
# which workgroups (in dispatch order) landed on which CU?
blocks = row_index // 4
for c in cus[:6]:
    b = np.unique(blocks[traced & (cu == c)])
    print(f"  CU {int(c):5d} (xcc {int(c) >> 8}): tracing workgroups (blockIdx) {b.tolist()}")
first = {}
for i in np.argsort(blocks):
    if traced[i]:
        first.setdefault(int(blocks[i]), int(cu[i]))
seq = [first[b] for b in sorted(first)][:48]
print("  CU of the first 48 tracing workgroups in blockIdx order:", [f"{c >> 8}.{c & 255}" for c in seq])
