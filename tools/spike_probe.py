import os, sys, time
import numpy as np
ROOT = os.getcwd()
sys.path.insert(0, ROOT)
import __graft_entry__ as graft
import torch
pkg = graft.load_package()
sc = pkg.Scene.load_from_file(os.path.join(ROOT, "scenes", "20spheres.yml")).set_size(1920, 1080)
r = pkg.Renderer(sc, device=0)
stream = torch.cuda.current_stream()
def orbit(i, n=24):
    a = 2.0 * np.pi * i / n
    pos = (5.0 + 14.0 * np.sin(a), 2.0 + 2.0 * np.sin(2 * a), 15.0 - 14.0 * np.cos(a))
    return pkg.camera_matrix(pos, float(np.degrees(np.arctan2(15.0 - pos[2], 5.0 - pos[0]))), float(-np.degrees(np.arctan2(pos[1] - 2.0, 14.0))))
for rep in range(3):
  for i in range(24):
    cam = orbit(i)
    for _ in range(3): r.update(cam)
    out = []
    for b in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record(stream)
        for _ in range(10): r.update(cam, stream=stream.cuda_stream, timed=False)
        e1.record(stream)
        t1 = time.perf_counter()
        e1.synchronize()
        out.append((e0.elapsed_time(e1) / 10.0 * 1e3, (t1 - t0) * 1e6 / 10))
    if max(o[0] for o in out) > 100:
        print(f"rep {rep} pose {i}: gpu us/frame " + " ".join(f"{o[0]:.0f}" for o in out) + " | host enqueue us/frame " + " ".join(f"{o[1]:.0f}" for o in out), flush=True)
print("done")
