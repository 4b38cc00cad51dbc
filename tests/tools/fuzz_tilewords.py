#!/usr/bin/env python3
"""Stress for the frame schedule of all-sphere scenes (tile words, paint workgroups, cost-class launch order): N random sphere
fields at image sizes with hundreds to thousands of tiles, a camera that drifts / jumps / looks away, RGBA32F and RGBA8, band
sharding; every frame of the product context must equal the frame of a context without tile words and launch order, and the
first frame the simple kernel's.  usage: python tests/tools/fuzz_tilewords.py [n_scenes] [first_seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

pkg = graft.load_package()


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    bad = frames = 0
    for seed in range(first, first + n):
        rng = np.random.default_rng(910000 + seed)
        w, h = int(rng.integers(100, 1300)), int(rng.integers(60, 800))
        s = pkg.Scene.new(w, h, float(rng.uniform(30, 90)), int(rng.integers(0, 4)), rng.uniform(0, 1, 3))
        for i in range(int(rng.integers(1, 90))):
            c = rng.uniform([-20, -12, 4], [20, 12, 60])
            s.add_object(pkg.surface_make("sphere", c, [float(rng.uniform(0.2, 4.0))]), rng.uniform(0, 1, 3), float(rng.uniform(0.2, 0.8)) if rng.random() < 0.15 else 0.0)
        for i in range(int(rng.integers(0, 24))):
            if rng.random() < 0.7:
                s.add_light("directional", rng.normal(size=3) + np.array([0, -1.0, 0]), rng.uniform(0, 1, 3), float(rng.uniform(0.2, 1.5)))
            else:
                s.add_light("spherical", rng.uniform([-25, -5, -5], [25, 25, 50]), rng.uniform(0, 1, 3), float(rng.uniform(50, 900)))
        fmt = pkg.RT_FMT_RGBA8 if seed % 4 == 3 else pkg.RT_FMT_RGBA32F
        world = [1, 1, 2, 3][seed % 4]
        rank = int(rng.integers(0, world))
        band = int(rng.choice([1, 5, 8, 16, 33]))
        a = pkg.Renderer(s, device=0, rank=rank, world=world, band_rows=band, fmt=fmt)
        b = pkg.Renderer(s, device=0, rank=rank, world=world, band_rows=band, fmt=fmt, flags=pkg.RT_FLAG_NOSCAN | pkg.RT_FLAG_STATIC_ORDER)
        c = pkg.Renderer(s, device=0, rank=rank, world=world, band_rows=band, fmt=fmt, flags=pkg.RT_FLAG_SIMPLE)
        pos, yaw, pitch = np.zeros(3), 90.0, 0.0
        ok = True
        for f in range(6):
            kind = int(rng.integers(0, 4))
            if kind == 0:
                pos = pos + rng.normal(scale=0.5, size=3); yaw += float(rng.normal(scale=3.0))
            elif kind == 1:
                pos = rng.uniform([-12, -5, -8], [12, 8, 10]); yaw = float(rng.uniform(30, 150)); pitch = float(rng.uniform(-20, 20))
            cam = pkg.camera_matrix(tuple(pos), -90.0 if kind == 2 else yaw, pitch)
            a.update(cam); b.update(cam)
            fa, fb = a.download(), b.download()
            ok = ok and np.array_equal(fa, fb, equal_nan=True)
            if f == 0:
                c.update(cam)
                ok = ok and np.array_equal(fa, c.download(), equal_nan=True)
            frames += 1
        for r in (a, b, c):
            r.cleanup_update()
        if not ok:
            bad += 1
            print(f"seed {seed}: MISMATCH ({w}x{h}, world {world} rank {rank} band {band}, fmt {fmt})", flush=True)
        if (seed - first) % 50 == 49:
            print(f"... {seed - first + 1} scenes, {frames} frames, {bad} mismatches", flush=True)
    print(f"fuzz_tilewords: {n} scenes, {frames} frames, {bad} mismatches")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
