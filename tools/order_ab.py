#!/usr/bin/env python3
"""A/B of the launch-order feedback (default) against index order (RT_FLAG_STATIC_ORDER): static poses and an orbit
where the camera moves every frame (the lists are then one frame stale)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

pkg = graft.load_package()
W, H = 1920, 1080
sc = pkg.Scene.load_from_file(os.path.join(ROOT, "scenes", "20spheres.yml")).set_size(W, H)
ra = pkg.Renderer(sc, device=0)
rb = pkg.Renderer(sc, device=0, flags=pkg.RT_FLAG_STATIC_ORDER)


def pose(i, n=24):
    a = 2.0 * np.pi * i / n
    pos = (5.0 + 14.0 * np.sin(a), 2.0 + 2.0 * np.sin(2 * a), 15.0 - 14.0 * np.cos(a))
    yaw = float(np.degrees(np.arctan2(15.0 - pos[2], 5.0 - pos[0])))
    pitch = float(-np.degrees(np.arctan2(pos[1] - 2.0, 14.0)))
    return pkg.camera_matrix(pos, yaw, pitch)


print("static poses (median of 20 frames each)")
for i in (0, 5, 12, 16, 20):
    cam = pose(i)
    ts = []
    for r in (ra, rb):
        r.update(cam); r.update(cam)
        ts.append(float(np.median([r.update(cam) for _ in range(20)])) * 1e3)
    same = np.array_equal(ra.download(), rb.download())
    print(f"  pose {i:2d}: feedback {ts[0]:6.1f} us   index order {ts[1]:6.1f} us   identical {same}")
cam = pkg.camera_matrix((5.0, 2.0, 1.0), -90.0, 0.0)   # looking away from the spheres: every tile is empty
ts = []
for r in (ra, rb):
    r.update(cam); r.update(cam)
    ts.append(float(np.median([r.update(cam) for _ in range(50)])) * 1e3)
print(f"  all-empty view: feedback {ts[0]:6.1f} us   index order {ts[1]:6.1f} us")
for n in (240, 2400):
    print(f"moving camera, {n} frames per orbit (every frame a new pose)")
    tot = []
    for r in (ra, rb):
        r.update(pose(0, n))
        tot.append(sum(r.update(pose(i, n)) for i in range(n)) / n * 1e3)
    print(f"  mean frame: feedback {tot[0]:6.1f} us   index order {tot[1]:6.1f} us")

print("start-up pose of the other BASELINE configs (median of 20 frames)")
for name, w, h in (("quadratic", 640, 480), ("reflection_test", 1920, 1080), ("clebsch", 3840, 2160), ("20spheres", 7680, 4320)):
    s2 = pkg.Scene.load_from_file(os.path.join(ROOT, "scenes", name + ".yml")).set_size(w, h)
    ts = []
    for fl in (0, pkg.RT_FLAG_STATIC_ORDER):
        r = pkg.Renderer(s2, device=0, flags=fl)
        r.update(None); r.update(None); r.update(None)
        ts.append(float(np.median([r.update(None) for _ in range(20)])) * 1e3)
        r.cleanup_update()
    print(f"  {name:16s} {w}x{h}: feedback {ts[0]:7.1f} us   index order {ts[1]:7.1f} us")
