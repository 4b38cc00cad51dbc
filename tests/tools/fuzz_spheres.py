#!/usr/bin/env python3
"""Fuzz the wave-per-block ("lean") instantiation of the wavefront kernel: N random scenes of spheres only, no mirrors --
the scenes it renders -- with everything its own-sphere rule and its per-block culling could trip over: tiny, huge,
overlapping and nested spheres, spheres far from the origin, cameras inside spheres, grazing and near-degenerate light
directions, point lights inside spheres, more than 64 spheres, odd image sizes.  Each scene is rendered by the lean
instantiation (three frames: index order, then launch-order feedback), by the general one (RT_FLAG_NOLEAN) and by the CPU
oracle; all frames must be identical, and so must the reference-equivalent counters of the two counting builds.
usage: python tests/tools/fuzz_spheres.py [n_scenes] [first_seed]"""
import os
import sys

import numpy as np

os.environ["MI355RT_LEAN"] = "always"   # (rt_create reads it: small frames would otherwise switch to the general instantiation after the first frame)

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as graft  # noqa: E402

pkg, O = graft.load_package(), graft.load_oracle()
from test_gpu_parity import oracle_from, render_desc  # noqa: E402


def scene(seed):
    rng = np.random.default_rng(515000 + seed)
    w, h = int(rng.integers(1, 200)), int(rng.integers(1, 140))
    s = pkg.Scene.new(w, h, float(rng.uniform(10, 110)), int(rng.integers(0, 5)), rng.uniform(0, 1, 3))
    style = int(rng.integers(0, 6))
    n_obj = int(rng.integers(4, 90)) if style != 5 else int(rng.integers(4, 12))
    scale = float(10 ** rng.uniform(-1, 2))                    # overall scene scale 0.1 .. 100
    shift = np.zeros(3)
    if style == 4:                                             # far from the origin: the own-sphere window must close by itself
        shift = rng.uniform(-1, 1, 3) * float(10 ** rng.uniform(3, 7))
    for i in range(n_obj):
        c = rng.uniform([-12, -8, 4], [12, 8, 40]) * scale + shift
        r = float(10 ** rng.uniform(-1.5, 1.0)) * scale
        if style == 1 and i % 4 == 0:                          # tiny
            r = float(10 ** rng.uniform(-4, -2)) * scale
        if style == 2 and i % 5 == 0:                          # huge (the camera is usually inside, or they cover everything)
            r = float(10 ** rng.uniform(1.5, 4)) * scale
        if style == 3 and i > 0 and i % 2 == 0:                # nested / overlapping: centred near the previous one
            c = prev_c + rng.normal(size=3) * 0.3 * prev_r
            r = prev_r * float(rng.uniform(0.5, 1.5))
        prev_c, prev_r = c, r
        s.add_object(pkg.surface_make("sphere", c, [r]), rng.uniform(0, 1, 3), 0.0)
    for i in range(int(rng.integers(0, 12))):
        if rng.random() < 0.65:
            d = rng.normal(size=3)
            if rng.random() < 0.2:
                d[int(rng.integers(0, 3))] *= 1e-9              # nearly axis-aligned
            s.add_light("directional", d, rng.uniform(0, 1, 3), float(rng.uniform(0, 2)))
        else:
            p = rng.uniform([-15, -10, -10], [15, 20, 40]) * scale + shift
            s.add_light("spherical", p, rng.uniform(0, 1, 3), float(rng.uniform(1, 900)) * scale * scale)
    pos = rng.uniform(-3, 3, 3) * scale + shift
    if style in (2, 3) and rng.random() < 0.5:                 # camera at a sphere's centre / just inside its surface
        pos = prev_c + rng.normal(size=3) * 0.5 * prev_r
    cam = pkg.camera_matrix(pos=pos, yaw_deg=float(rng.uniform(60, 120)), pitch_deg=float(rng.uniform(-25, 25)))
    return s, cam


def counters(s, cam, flags):
    r = pkg.Renderer(s, device=0, flags=flags | pkg.RT_FLAG_COUNT)
    r.update(cam)
    c = r.counters()
    img = r.download().copy()
    r.cleanup_update()
    return c, img


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    bad = 0
    for seed in range(first, first + n):
        s, cam = scene(seed)
        a = render_desc(pkg, s, cam)
        b = render_desc(pkg, s, cam, flags=pkg.RT_FLAG_NOLEAN)
        f = render_desc(pkg, s, cam, flags=pkg.RT_FLAG_FAST)
        g = render_desc(pkg, s, cam, flags=pkg.RT_FLAG_FAST | pkg.RT_FLAG_NOLEAN)
        want = oracle_from(pkg, O, s).render(cam=cam, nthreads=4)
        ok = np.array_equal(a, b) and np.array_equal(a[..., :3], want, equal_nan=True) and np.array_equal(f, g, equal_nan=True)
        ca, ia = counters(s, cam, 0)
        cb, ib = counters(s, cam, pkg.RT_FLAG_NOLEAN)
        okc = all(ca[k] == cb[k] for k in ("primary_rays", "shadow_rays", "reflect_rays", "tests", "hits")) and np.array_equal(ia, a) and np.array_equal(ib, a)
        if not (ok and okc):
            bad += 1
            d = s.desc()
            print(f"seed {seed}: MISMATCH  lean-vs-general {np.array_equal(a, b)}  vs-oracle {np.array_equal(a[..., :3], want, equal_nan=True)}  fast lean-vs-general {np.array_equal(f, g, equal_nan=True)}  "
                  f"counters {okc}  ({d.width}x{d.height}, {d.n_objects} objects, {d.n_lights} lights)", flush=True)
        if (seed - first) % 50 == 49:
            print(f"... {seed - first + 1} scenes, {bad} mismatches", flush=True)
    print(f"fuzz_spheres: {n} scenes, {bad} mismatches")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
