#!/usr/bin/env python3
"""Timing experiment (library built with `make DEBUG_EXITS=1`): where does an empty tile's workgroup spend its time?
The all-background view of 20spheres; MI355RT_DEBUG_EXIT=1 index slots leave at once, 2 after their first loads, 4 after
the barrier (no paint).  Frames are wrong by construction; only the times matter."""
import os
import subprocess
import sys
os.environ["MI355RT_ALLOW_DIAGNOSTIC"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    import __graft_entry__ as graft
    pkg = graft.load_package()
    away = pkg.camera_matrix((0.0, 0.0, 0.0), -90.0, 0.0)
    for W, H in ((1920, 1080), (7680, 4320)):
        sc = pkg.Scene.load_from_file(os.path.join(ROOT, "scenes", "20spheres.yml")).set_size(W, H)
        fb = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
        stream = torch.cuda.current_stream()
        for name, fl in (("scan", 0), ("noscan", 32), ("static_noscan", 48), ("static_scan", 16)):
            r = pkg.Renderer(sc, device=0, flags=fl)
            for cam_name, cam in (("empty", away), ("start", pkg.IDENTITY)):
                for _ in range(5):
                    r.update(cam, dev_fb=fb.data_ptr(), stream=stream.cuda_stream, timed=False)
                reps = 50 if W < 4000 else 10
                ts = []
                for _ in range(5):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(stream)
                    for _ in range(reps):
                        r.update(cam, dev_fb=fb.data_ptr(), stream=stream.cuda_stream, timed=False)
                    e1.record(stream)
                    torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1) / reps * 1e3)
                print(f"  {W}x{H} {name:14s} {cam_name:6s} {np.median(ts):8.1f} us", flush=True)
            r.cleanup_update()
else:
    for mode in ("0", "1", "2", "4"):
        print(f"MI355RT_DEBUG_EXIT={mode}", flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, MI355RT_DEBUG_EXIT=mode))
