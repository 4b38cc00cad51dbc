#!/usr/bin/env python3
"""Per-frame times of back-to-back frames right after a camera jump (orbit pose i-1 -> i): which frames are slow, and how slow.
usage: python tools/transient_probe.py [first pose [last pose]]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402
import torch  # noqa: E402

pkg = graft.load_package()
sc = pkg.Scene.load_from_file(os.path.join(ROOT, "scenes", "20spheres.yml")).set_size(1920, 1080)
r = pkg.Renderer(sc, device=0, flags=int(os.environ.get("PROBE_FLAGS", "0")))
stream = torch.cuda.current_stream()


def orbit(i, n=24):
    a = 2.0 * np.pi * i / n
    pos = (5.0 + 14.0 * np.sin(a), 2.0 + 2.0 * np.sin(2 * a), 15.0 - 14.0 * np.cos(a))
    return pkg.camera_matrix(pos, float(np.degrees(np.arctan2(15.0 - pos[2], 5.0 - pos[0]))), float(-np.degrees(np.arctan2(pos[1] - 2.0, 14.0))))


first = int(sys.argv[1]) if len(sys.argv) > 1 else 17
last = int(sys.argv[2]) if len(sys.argv) > 2 else 20
settle = int(os.environ.get("PROBE_SETTLE", "3"))
for i in range(first - 1, last + 1):
    cam = orbit(i)
    for _ in range(settle):
        r.update(cam)
    alone = [r.update(cam) for _ in range(5)]
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(25)]
    ev[0].record(stream)
    for k in range(24):
        r.update(cam, stream=stream.cuda_stream, timed=False)
        ev[k + 1].record(stream)
    torch.cuda.synchronize()
    per = [ev[k].elapsed_time(ev[k + 1]) * 1e3 for k in range(24)]
    print(f"pose {i:2d}: alone {np.median(alone)*1e3:6.1f} us; back to back: " + " ".join(f"{t:.0f}" for t in per), flush=True)
