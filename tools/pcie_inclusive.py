import os, sys, time
import numpy as np
ROOT = os.getcwd(); sys.path.insert(0, ROOT)
import __graft_entry__ as graft
pkg = graft.load_package()
sc = pkg.Scene.load_from_file(os.path.join(ROOT, "scenes", "20spheres.yml")).set_size(1920, 1080)
for name, fmt in (("rgba32f", pkg.RT_FMT_RGBA32F), ("rgba8", pkg.RT_FMT_RGBA8)):
    r = pkg.Renderer(sc, device=0, fmt=fmt)
    for _ in range(5):
        r.update(); img = r.download()
    t0 = time.perf_counter()
    n = 50
    for _ in range(n):
        r.update(); img = r.download()
    dt = (time.perf_counter() - t0) / n
    print(f"{name}: update() + download to host memory: {dt*1e3:.3f} ms per frame = {1/dt:.0f} frames/s ({img.nbytes/1e6:.1f} MB per frame, {img.nbytes/dt/1e9:.1f} GB/s)")
    r.cleanup_update()
