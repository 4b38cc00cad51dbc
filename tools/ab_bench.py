#!/usr/bin/env python3
"""A/B timing of the kernel variants in one process (interleaved, same scene): evidence for DESIGN.md's choices.
usage: python tools/ab_bench.py [scene] [W] [H] [max_refl] [reps]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

pkg = graft.load_package()
name = sys.argv[1] if len(sys.argv) > 1 else "20spheres"
W = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
H = int(sys.argv[3]) if len(sys.argv) > 3 else 1080
mr = int(sys.argv[4]) if len(sys.argv) > 4 else -1
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 30
only = sys.argv[6].split(",") if len(sys.argv) > 6 else None
sc = pkg.Scene.load_from_file(os.path.join(ROOT, "scenes", name + ".yml")).set_size(W, H)
if mr >= 0:
    sc.set_max_reflections(mr)
variants = [("wavefront strict", 0), ("wavefront-nocull strict", pkg.RT_FLAG_NOCULL), ("simple strict", pkg.RT_FLAG_SIMPLE),
            ("wavefront fast", pkg.RT_FLAG_FAST), ("wavefront-nocull fast", pkg.RT_FLAG_FAST | pkg.RT_FLAG_NOCULL),
            ("simple fast", pkg.RT_FLAG_FAST | pkg.RT_FLAG_SIMPLE)]
if only:
    variants = [v for v in variants if v[0].split()[0] in only or v[0] in only]
rens = [(n, pkg.Renderer(sc, device=0, flags=f)) for n, f in variants]
times = {n: [] for n, _ in variants}
for n, r in rens:
    for _ in range(3):
        r.update()
for _ in range(reps):
    for n, r in rens:
        times[n].append(r.update())
ref = None
print(f"{name} {W}x{H}")
for n, r in rens:
    img = r.download()
    if ref is None:
        ref = img
    t = np.array(times[n])
    print(f"  {n:26s} median {np.median(t)*1e3:9.1f} us  min {t.min()*1e3:9.1f} us   identical to first: {np.array_equal(img, ref)}")
    r.cleanup_update()
for n, f in variants[:2]:
    r = pkg.Renderer(sc, device=0, flags=f | pkg.RT_FLAG_COUNT)
    r.update()
    print(f"  counters {n:24s}", r.counters())
    r.cleanup_update()
