#!/usr/bin/env python3
"""A/B of rt_config flag sets on the 20spheres scene at several sizes and poses, back-to-back launches timed with one HIP
event pair per batch (what bench.py measures), interleaved in one process.  Evidence for DESIGN.md's schedule choices.
usage: python tools/ab_flags.py [name=flags ...]     e.g.  scan=0 noscan=32 static=16 static_noscan=48"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402
import torch  # noqa: E402

pkg = graft.load_package()
variants = [a.split("=") for a in sys.argv[1:] if "=" in a] or [["scan", "0"], ["noscan", str(pkg.RT_FLAG_NOSCAN)]]
variants = [(n, int(f)) for n, f in variants]
sizes = [(1920, 1080), (3840, 2160), (7680, 4320)]
SCENE = os.environ.get("AB_SCENE", "20spheres")   # AB_SCENE=reflection_test AB_SIZES=1920x1080 AB_REFL=4 for the other configs
if os.environ.get("AB_SIZES"):
    sizes = [tuple(int(v) for v in t.split("x")) for t in os.environ["AB_SIZES"].split(",")]
away = pkg.camera_matrix((0.0, 0.0, 0.0), -90.0, 0.0)
poses = [("start", pkg.IDENTITY), ("orbit5", None), ("orbit16", None), ("empty", away), ("orbit6", None), ("orbit19", None)]


def orbit(i, n=24):
    a = 2.0 * np.pi * i / n
    pos = (5.0 + 14.0 * np.sin(a), 2.0 + 2.0 * np.sin(2 * a), 15.0 - 14.0 * np.cos(a))
    yaw = float(np.degrees(np.arctan2(15.0 - pos[2], 5.0 - pos[0])))
    pitch = float(-np.degrees(np.arctan2(pos[1] - 2.0, 14.0)))
    return pkg.camera_matrix(pos, yaw, pitch)


poses[1] = ("orbit5", orbit(5))
poses[2] = ("orbit16", orbit(16))
poses[4] = ("orbit6", orbit(6))
poses[5] = ("orbit19", orbit(19))
for extra in [int(v) for v in os.environ.get("AB_ORBIT", "").split(",") if v]:   # AB_ORBIT=7,20: further poses of the orbit
    poses.append((f"orbit{extra}", orbit(extra)))
stream = torch.cuda.current_stream()
for W, H in sizes:
    sc = pkg.Scene.load_from_file(os.path.join(ROOT, "scenes", SCENE + ".yml")).set_size(W, H)
    if os.environ.get("AB_REFL"):
        sc.set_max_reflections(int(os.environ["AB_REFL"]))
    fb = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
    rens = [(n, pkg.Renderer(sc, device=0, flags=f)) for n, f in variants]
    reps = 100 if W <= 1920 else (40 if W <= 3840 else 15)
    for pname, cam in poses:
        out = []
        ref = None
        for n, r in rens:
            for _ in range(5):
                r.update(cam, dev_fb=fb.data_ptr(), stream=stream.cuda_stream, timed=False)
            ts = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for _ in range(reps):
                    r.update(cam, dev_fb=fb.data_ptr(), stream=stream.cuda_stream, timed=False)
                e1.record(stream)
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / reps * 1e3)
            img = fb.cpu().numpy()
            if ref is None:
                ref = img
            out.append(f"{n} {np.median(ts):8.1f} us{'' if np.array_equal(img, ref) else ' (FRAME DIFFERS)'}" + (f" batches {' '.join(f'{t:.1f}' for t in ts)}" if os.environ.get("AB_BATCHES") else ""))
        print(f"{W}x{H} {pname:8s} " + "   ".join(out), flush=True)
    for _, r in rens:
        r.cleanup_update()
