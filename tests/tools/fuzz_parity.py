#!/usr/bin/env python3
"""Fuzz the bit-exactness claim: N random degree <= 2 scenes (spheres incl. huge / tiny / overlapping ones, random
quadrics, planes, both light kinds, mirrors, random cameras, odd image sizes), each rendered by the wavefront kernel,
the wavefront kernel without culling, the simple kernel and the CPU oracle; every frame must be identical.
usage: python tests/tools/fuzz_parity.py [n_scenes] [first_seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as graft  # noqa: E402

pkg, O = graft.load_package(), graft.load_oracle()
from test_gpu_parity import oracle_from, render_desc  # noqa: E402


def scene(seed):
    rng = np.random.default_rng(77000 + seed)
    w, h = int(rng.integers(1, 140)), int(rng.integers(1, 100))
    s = pkg.Scene.new(w, h, float(rng.uniform(10, 110)), int(rng.integers(0, 5)), rng.uniform(0, 1, 3))
    n_obj = int(rng.integers(0, 30))
    scale = float(10 ** rng.uniform(-1, 2))  # overall scene scale 0.1 .. 100
    for i in range(n_obj):
        kind = int(rng.integers(0, 6))
        refl = float(rng.uniform(0.05, 1.0)) if rng.random() < 0.25 else 0.0
        col = rng.uniform(0, 1, 3)
        if kind <= 2:
            c = rng.uniform([-12, -8, -5], [12, 8, 40]) * scale
            r = float(10 ** rng.uniform(-1.5, 1.0)) * scale
            s.add_object(pkg.surface_make("sphere", c, [r]), col, refl)
        elif kind == 3:
            q = np.zeros(20)
            q[10:13] = rng.uniform(-2, 2, 3)
            if rng.random() < 0.5:
                q[13:16] = rng.uniform(-1, 1, 3)
            c = rng.uniform([-6, -4, 4], [6, 4, 25]) * scale
            q[16:19] = -2.0 * q[10:13] * c
            q[19] = float(np.dot(q[10:13], c * c) - rng.uniform(0.2, 8.0) * scale * scale)
            s.add_object(q, col, refl)
        elif kind == 4:
            n = rng.normal(size=3)
            s.add_object(pkg.surface_make("plane", rng.uniform(-8, 8, 3) * scale, n), col, refl)
        else:  # unit-square class with an imaginary radius (never hit) or a degenerate one
            q = np.zeros(20)
            q[10:13] = 1.0
            q[16:19] = rng.uniform(-4, 4, 3)
            q[19] = float(rng.uniform(0, 50))
            s.add_object(q, col, refl)
    for i in range(int(rng.integers(0, 9))):
        if rng.random() < 0.5:
            s.add_light("directional", rng.normal(size=3), rng.uniform(0, 1, 3), float(rng.uniform(0, 2)))
        else:
            s.add_light("spherical", rng.uniform([-15, -10, -10], [15, 20, 40]) * scale, rng.uniform(0, 1, 3), float(rng.uniform(1, 900)) * scale * scale)
    cam = pkg.camera_matrix(pos=rng.uniform(-3, 3, 3) * scale, yaw_deg=float(rng.uniform(60, 120)), pitch_deg=float(rng.uniform(-25, 25)))
    return s, cam


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    bad = 0
    for seed in range(first, first + n):
        s, cam = scene(seed)
        a = render_desc(pkg, s, cam)
        b = render_desc(pkg, s, cam, flags=pkg.RT_FLAG_NOCULL)
        c = render_desc(pkg, s, cam, flags=pkg.RT_FLAG_SIMPLE)
        want = oracle_from(pkg, O, s).render(cam=cam, nthreads=4)
        ok = np.array_equal(a, b) and np.array_equal(a, c) and np.array_equal(a[..., :3], want, equal_nan=True)
        if not ok:
            bad += 1
            d = s.desc()
            print(f"seed {seed}: MISMATCH  cull-vs-nocull {np.array_equal(a, b)}  wavefront-vs-simple {np.array_equal(a, c)}  "
                  f"vs-oracle {np.array_equal(a[..., :3], want, equal_nan=True)}  ({d.width}x{d.height}, {d.n_objects} objects, {d.n_lights} lights)", flush=True)
        if (seed - first) % 50 == 49:
            print(f"... {seed - first + 1} scenes, {bad} mismatches", flush=True)
    print(f"fuzz: {n} scenes, {bad} mismatches")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
