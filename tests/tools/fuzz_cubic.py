#!/usr/bin/env python3
"""Fuzz the degree-3 path: N random scenes with one or two random cubic surfaces (dense or sparse coefficients) next
to spheres / a plane, random lights and cameras.  Device cbrt / acos / cos differ from glibc's in the last ulp, so the
bar is the north star's 1e-5 relative per channel; pixels beyond it are solver flips at root discontinuities
(tangent rays, triple roots) and are counted.  The wavefront and the simple kernel share the device functions and must
agree bit for bit.  usage: python tests/tools/fuzz_cubic.py [n_scenes] [first_seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as graft  # noqa: E402

pkg = graft.load_package()
O = graft.load_oracle()
from conftest import compare  # noqa: E402
from test_gpu_parity import oracle_from, render_desc  # noqa: E402


def scene(seed):
    rng = np.random.default_rng(88000 + seed)
    w, h = int(rng.integers(40, 200)), int(rng.integers(30, 150))
    s = pkg.Scene.new(w, h, float(rng.uniform(25, 80)), int(rng.integers(0, 4)), rng.uniform(0, 1, 3))
    for _ in range(int(rng.integers(1, 3))):
        q = np.zeros(20)
        q[:10] = rng.uniform(-1, 1, 10) * (rng.random(10) < rng.uniform(0.2, 1.0))
        q[10:16] = rng.uniform(-1, 1, 6) * (rng.random(6) < 0.8)
        q[16:19] = rng.uniform(-2, 2, 3)
        q[19] = rng.uniform(-4, 4)
        s.add_object(q, rng.uniform(0, 1, 3), float(rng.uniform(0.1, 0.8)) if rng.random() < 0.3 else 0.0)
    for _ in range(int(rng.integers(0, 6))):
        c = rng.uniform([-4, -3, -2], [4, 3, 8])
        s.add_object(pkg.surface_make("sphere", c, [float(rng.uniform(0.2, 1.5))]), rng.uniform(0, 1, 3), float(rng.uniform(0.1, 0.8)) if rng.random() < 0.2 else 0.0)
    if rng.random() < 0.5:
        s.add_object(pkg.surface_make("plane", [0, float(rng.uniform(-5, -2)), 0], [float(rng.normal(scale=0.1)), 1.0, float(rng.normal(scale=0.1))]), (0.5, 0.5, 0.5))
    for _ in range(int(rng.integers(1, 5))):
        if rng.random() < 0.5:
            s.add_light("directional", rng.normal(size=3) + np.array([0, -1.0, 0]), rng.uniform(0, 1, 3), float(rng.uniform(0.3, 1.5)))
        else:
            s.add_light("spherical", rng.uniform([-8, 2, -12], [8, 12, 6]), rng.uniform(0, 1, 3), float(rng.uniform(50, 500)))
    cam = pkg.camera_matrix(pos=(float(rng.uniform(-2, 2)), float(rng.uniform(-1, 2)), float(rng.uniform(-12, -6))), yaw_deg=float(rng.uniform(80, 100)),
                            pitch_deg=float(rng.uniform(-10, 10)))
    return s, cam, w * h


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    kernels_differ, over, worst, px_total, px_bad, px_diff, exact = 0, 0, 0.0, 0, 0, 0, 0
    for seed in range(first, first + n):
        s, cam, npx = scene(seed)
        a = render_desc(pkg, s, cam)
        if not np.array_equal(a, render_desc(pkg, s, cam, flags=pkg.RT_FLAG_SIMPLE)):
            kernels_differ += 1
            print(f"seed {seed}: wavefront and simple kernels differ")
        want = oracle_from(pkg, O, s).render(cam=cam, nthreads=4)
        c = compare(a[..., :3], want)
        nd = int(np.any(a[..., :3] != want, axis=-1).sum())
        px_diff += nd
        exact += nd == 0
        px_total += npx
        px_bad += c["n_bad_pixels"]
        frac = c["n_bad_pixels"] / npx
        worst = max(worst, frac)
        if c["n_bad_pixels"] > max(3, int(0.002 * npx)):
            over += 1
            print(f"seed {seed}: {c['n_bad_pixels']} of {npx} pixels beyond 1e-5 ({100 * frac:.2f} %)")
        if (seed - first + 1) % 50 == 0:
            print(f"... {seed - first + 1} scenes", flush=True)
    print(f"cubic fuzz: {n} scenes, {px_bad} of {px_total} pixels beyond 1e-5 relative ({100.0 * px_bad / max(px_total, 1):.4f} %), worst scene {100 * worst:.2f} %, "
          f"{over} scenes over the 0.2 % test bound, {kernels_differ} with kernels disagreeing; {exact} scenes bit-identical to the oracle, "
          f"{px_diff} pixels differ in some bit")
    return 1 if kernels_differ else 0


if __name__ == "__main__":
    sys.exit(main())
