#!/usr/bin/env python3
"""Moving-camera sequence: the reference host's camera flies an orbit around the 20spheres
scene; time per frame for each pose.  Culling and hit density change with the view, so this shows how stable
the frame time is away from the start-up pose that bench.py uses.  `us` = per frame of 10 frames of the pose issued back to back
(one HIP event pair, single launches), median of three such batches; `alone` = one frame into an idle GPU with a host synchronisation after it."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402
import torch  # noqa: E402

pkg = graft.load_package()
W, H = 1920, 1080
sc = pkg.Scene.load_from_file(os.path.join(ROOT, "scenes", "20spheres.yml")).set_size(W, H)
r = pkg.Renderer(sc, device=0)
rc = pkg.Renderer(sc, device=0, flags=pkg.RT_FLAG_COUNT) if not os.environ.get("FLY_NO_COUNT") else None
stream = torch.cuda.current_stream()
print(f"{'frame':>5} {'pos':>24} {'yaw':>6} {'pitch':>6} {'us':>8} {'alone':>7} {'hits':>8} {'rays':>9} {'Mrays/s':>9} {'exec':>9} {'solves':>8} {'culls':>8}")
ts = []
for i in range(24):
    a = 2.0 * np.pi * i / 24
    pos = (5.0 + 14.0 * np.sin(a), 2.0 + 2.0 * np.sin(2 * a), 15.0 - 14.0 * np.cos(a))   # orbit around (5, 2, 15)
    yaw = float(np.degrees(np.arctan2(15.0 - pos[2], 5.0 - pos[0])))
    pitch = float(-np.degrees(np.arctan2(pos[1] - 2.0, 14.0)))
    cam = pkg.camera_matrix(pos, yaw, pitch)
    for _ in range(3):   # the launch-order feedback settles on the new view
        r.update(cam)
    alone = np.median([r.update(cam) for _ in range(5)])
    batches = []
    for _ in range(3):   # median of three batches (a host hiccup while enqueuing would otherwise own the pose)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(10):
            r.update(cam, stream=stream.cuda_stream, timed=False)
        e1.record(stream)
        e1.synchronize()
        batches.append(e0.elapsed_time(e1) / 10.0)
    t = float(np.median(batches))
    if rc is not None:
        rc.update(cam)
    c = rc.counters() if rc is not None else dict(hits=0, rays_total=0, tests_executed=0, solves=0, cull_evals=0)
    ts.append(t)
    print(f"{i:5d} ({pos[0]:7.2f},{pos[1]:6.2f},{pos[2]:7.2f}) {yaw:6.1f} {pitch:6.1f} {t*1e3:8.1f} {alone*1e3:7.1f} {c['hits']:8d} {c['rays_total']:9d} {c['rays_total']/t/1e3:9.0f} {c['tests_executed']:9d} {c['solves']:8d} {c['cull_evals']:8d}")
ts = np.array(ts)
print(f"frame time over the orbit: min {ts.min()*1e3:.1f} us, median {np.median(ts)*1e3:.1f} us, max {ts.max()*1e3:.1f} us")
