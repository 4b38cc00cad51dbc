#!/usr/bin/env python3
"""Synthetic N-sphere x M-light scaling sweep: frame time of the product kernel vs the
simple kernel as the scene grows, at 1920x1080.  Scenes are seeded random sphere fields over a floor plane."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

pkg = graft.load_package()


def field(n_spheres, n_lights, w=1920, h=1080, seed=1):
    rng = np.random.default_rng(seed)
    s = pkg.Scene.new(w, h, 60.0, 2, (0.0, 0.1, 0.2))
    for _ in range(n_spheres):
        c = rng.uniform([-20, -8, 12], [20, 12, 60])
        s.add_object(pkg.surface_make("sphere", c, [float(rng.uniform(0.4, 1.6))]), rng.uniform(0.2, 1, 3))
    s.add_object(pkg.surface_make("plane", [0, -9, 0], [0, 1, 0]), (0.5, 0.5, 0.5))
    for i in range(n_lights):
        s.add_light("directional", rng.normal(size=3) * 0.4 + np.array([0.3, -1.0, 0.2]), (1, 1, 1), 1.5 / n_lights)
    return s


print(f"{'spheres':>8} {'lights':>6} {'wavefront us':>13} {'nocull us':>10} {'simple us':>10} {'identical':>9} {'Mrays/s':>9}")
for n_s, n_l in ((5, 4), (20, 19), (64, 19), (128, 19), (256, 19), (400, 8), (64, 64)):
    try:
        sc = field(n_s, n_l)
        out = {}
        ref = None
        same = True
        for name, fl in (("wavefront", 0), ("nocull", pkg.RT_FLAG_NOCULL), ("simple", pkg.RT_FLAG_SIMPLE)):
            r = pkg.Renderer(sc, device=0, flags=fl)
            r.update()
            t = np.median([r.update() for _ in range(5)])
            img = r.download()
            r.cleanup_update()
            out[name] = t * 1e3
            if ref is None:
                ref = img
            else:
                same = same and np.array_equal(ref, img)
        rc = pkg.Renderer(sc, device=0, flags=pkg.RT_FLAG_COUNT)
        rc.update()
        rays = rc.counters()["rays_total"]
        rc.cleanup_update()
        print(f"{n_s:8d} {n_l:6d} {out['wavefront']:13.1f} {out['nocull']:10.1f} {out['simple']:10.1f} {str(same):>9} {rays / out['wavefront']:9.0f}")
    except pkg.RtError as e:
        print(f"{n_s:8d} {n_l:6d}  {e}")
