"""Function-level checks of the oracle.  The reference has no unit vectors, so these pin the restated functions
through (a) mathematical properties that hold for the reference's formulas, (b) the behaviours SURVEY.md's quirk
list Q1-Q14 calls parity-critical, and (c) the committed golden frames."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import ROOT, scene_path

D3 = C.c_double * 3


def poly(c, p):
    x, y, z = p
    mono = [x**3, y**3, z**3, x*x*y, x*y*y, x*x*z, x*z*z, y*y*z, y*z*z, x*y*z, x*x, y*y, z*z, x*y, x*z, y*z, x, y, z, 1.0]
    return float(np.dot(c, mono))


def intersect(oracle, c, o, d):
    L = oracle.lib()
    tc = (C.c_double * 4)()
    br = C.c_int()
    t = L.orc_intersect_ray_ex((C.c_double * 20)(*c), D3(*o), D3(*d), tc, C.byref(br))
    return t, list(tc), br.value


def test_expansion_coefficients_are_the_ray_polynomial(oracle):
    """t3..t0 are the coefficients of F(o + t d) in t (include/surface_impl.h:44-103)."""
    rng = np.random.default_rng(1)
    for _ in range(200):
        c = rng.normal(size=20)
        o, d = rng.normal(size=3), rng.normal(size=3)
        _, tc, _ = intersect(oracle, c, o, d)
        for t in (-1.3, 0.0, 0.7, 2.1):
            want = poly(c, o + t * d)
            got = ((tc[3] * t + tc[2]) * t + tc[1]) * t + tc[0]
            assert abs(got - want) <= 1e-10 * (1 + abs(want)) * 50


def test_all_solver_branches_return_roots(oracle):
    rng = np.random.default_rng(2)
    seen = set()
    for i in range(4000):
        c = rng.normal(size=20)
        kind = i % 4
        if kind >= 1:
            c[:10] = 0  # quadric
        if kind >= 2:
            c[10:16] = 0  # plane
        if kind == 3:
            c[16:19] = 0  # constant
        o, d = rng.normal(size=3), rng.normal(size=3)
        d /= np.linalg.norm(d)
        t, tc, br = intersect(oracle, c, o, d)
        seen.add(br)
        if br in (1, 3, 4, 5) and np.isfinite(t):
            val = ((tc[3] * t + tc[2]) * t + tc[1]) * t + tc[0]
            scale = abs(tc[3] * t**3) + abs(tc[2] * t * t) + abs(tc[1] * t) + abs(tc[0]) + 1e-300
            assert abs(val) <= 1e-7 * scale, (br, t, val, scale)
        if br in (0, 2):
            assert t == -1.0
    assert seen == {0, 1, 2, 3, 4, 5}


def test_quadratic_quirk_negative_leading_coefficient(oracle):
    """Q5: the first candidate is (-t1 - sqrt(delta)) / (2 t2); with t2 < 0 that is the LARGER root and it is
    returned even though the smaller one is also acceptable (include/surface_impl.h:139-148)."""
    c = np.zeros(20)
    c[10:13] = -1.0  # -(x^2 + y^2 + z^2) + 1 = 0: unit sphere with negated polynomial
    c[19] = 1.0
    t, tc, br = intersect(oracle, c, [0, 0, -5], [0, 0, 1])
    assert br == 3 and tc[2] < 0 and t == 6.0  # far side, not 4.0
    c = -c
    assert intersect(oracle, c, [0, 0, -5], [0, 0, 1])[0] == 4.0


def test_degree_selection_uses_absolute_eps(oracle):
    """Q3: |t3| <= 1e-7 falls through to the quadratic even though t3 != 0 (include/surface_impl.h:106,138,150)."""
    c = np.zeros(20)
    c[2] = 5e-8   # z3: t3 = 5e-8 for d = (0,0,1)
    c[12] = 1.0   # z2
    c[19] = -4.0
    t, tc, br = intersect(oracle, c, [0, 0, 0], [0, 0, 1])
    assert tc[3] == 5e-8 and br == 3 and t == 2.0
    c[2] = 2e-7
    assert intersect(oracle, c, [0, 0, 0], [0, 0, 1])[2] in (4, 5)


def test_triple_root_is_nan(oracle):
    """Q4: delta == 0 and q == 0 gives acos(0/0) = NaN, which the callers' comparisons reject."""
    c = np.zeros(20)
    c[2] = 1.0  # z^3 = 0
    t, _, br = intersect(oracle, c, [0, 0, -1], [0, 0, 1])  # (t - 1)^3
    assert br == 5 and np.isnan(t)


def test_normal_is_the_normalised_gradient(oracle):
    rng = np.random.default_rng(3)
    L = oracle.lib()
    for _ in range(100):
        c, p = rng.normal(size=20), rng.normal(size=3)
        out = D3()
        L.orc_normal_vector((C.c_double * 20)(*c), D3(*p), out)
        h = 1e-6
        g = np.array([(poly(c, p + h * e) - poly(c, p - h * e)) / (2 * h) for e in np.eye(3)])
        assert np.allclose(np.array(out), g / np.linalg.norm(g), atol=1e-6)


def test_shadow_ray_goes_through_float32(oracle):
    """Q10: shadow_ray returns a float vector; max_t is 1 for point lights and 1e6 for directional ones."""
    L = oracle.lib()
    light = oracle.OrcLight()
    L.orc_light_spherical(2.0, D3(0.1, 4.0, 4.0), (C.c_float * 3)(1, 0.8, 0.4), C.byref(light))
    sp = np.array([0.3, -1.7, 2.2])
    out, mt = (C.c_float * 3)(), C.c_double()
    L.orc_shadow_ray(C.byref(light), D3(*sp), out, C.byref(mt))
    assert mt.value == 1.0
    assert np.array_equal(np.array(out, dtype=np.float32), (np.array([0.1, 4.0, 4.0]) - sp).astype(np.float32))
    L.orc_light_directional(3.0, D3(0.8, -0.3, 0.2), (C.c_float * 3)(1, 1, 1), C.byref(light))
    L.orc_shadow_ray(C.byref(light), D3(*sp), out, C.byref(mt))
    v = -np.array([0.8, -0.3, 0.2]) * (1.0 / np.sqrt(0.8 * 0.8 + 0.3 * 0.3 + 0.2 * 0.2))
    assert mt.value == 1e6 and np.allclose(np.array(light.p), v, rtol=1e-15)
    assert np.array_equal(np.array(out, dtype=np.float32), np.array(light.p).astype(np.float32))
    assert np.array_equal(np.array(light.color, dtype=np.float32), np.float32(3.0) * np.ones(3, np.float32))


def test_surface_color_formula(oracle):
    """Q11 in numpy float32: albedo/pi * light * max(0, n.l), point lights / (4 pi |d|^2)."""
    L = oracle.lib()
    f32 = np.float32
    pi = f32(3.14159274)
    n = np.array([0.0, 0.6, 0.8])
    p = np.array([1.0, 2.0, 3.0])
    alb = np.array([0.8, 0.5, 0.25], dtype=f32)
    light = oracle.OrcLight()
    L.orc_light_spherical(400.0, D3(1.0, 8.0, 3.0), (C.c_float * 3)(0, 1, 0.5), C.byref(light))
    out = (C.c_float * 3)()
    L.orc_surface_color(C.byref(light), D3(*p), D3(*n), (C.c_float * 3)(*alb), out)
    lc = f32(400.0) * np.array([0, 1, 0.5], dtype=f32)
    col = lc / (f32(4.0) * pi * f32(36.0))
    want = alb / pi * col * f32(0.6)
    assert np.array_equal(np.array(out, dtype=f32), want)
    # back-facing: max(0, .) clamps to exactly 0, never negative (normals are not flipped, Q8)
    L.orc_surface_color(C.byref(light), D3(*p), D3(0.0, -1.0, 0.0), (C.c_float * 3)(*alb), out)
    assert np.array_equal(np.array(out, dtype=f32), np.zeros(3, f32))


def test_primary_ray_convention(oracle):
    """Q1: pixel centres, row 0 at the bottom, +z forward for the identity camera."""
    s = oracle.load_scene(scene_path("quadratic")).with_size(640, 480)
    sc = s.c_scene()
    L = oracle.lib()
    cam = oracle.IDENTITY.ctypes.data_as(C.POINTER(C.c_double))
    out = D3()
    L.orc_primary_dir(C.byref(sc), cam, 0, 0, out)
    bl = np.array(out)
    L.orc_primary_dir(C.byref(sc), cam, 639, 479, out)
    tr = np.array(out)
    assert bl[0] < 0 and bl[1] < 0 and bl[2] > 0 and np.allclose(tr, [-bl[0], -bl[1], bl[2]], atol=1e-15)
    tanf = np.tan(np.radians(60) / 2)
    assert np.isclose(bl[1] / bl[2], -(1 - 1 / 480) * tanf) and np.isclose(bl[0] / bl[2], -(1 - 1 / 640) * tanf * 640 / 480)


def two_mirror_scene(oracle, max_reflections):
    """Two facing mirrors (planes z = 10 and z = -10): every ray keeps bouncing, so the depth limit is reached."""
    s = oracle.Scene(32, 24, 40.0, max_reflections, (0.2, 0.4, 0.6))
    L = oracle.lib()
    for z, nz in ((10.0, -1.0), (-10.0, 1.0)):
        out = (C.c_double * 20)()
        L.orc_surface_plane(D3(0, 0, z), D3(0, 0, nz), out)
        s.add_object(list(out), (0.9, 0.1, 0.1), 0.5)
    light = oracle.OrcLight()
    L.orc_light_directional(1.0, D3(0.2, -1.0, 0.3), (C.c_float * 3)(1, 1, 1), C.byref(light))
    s.lights.append(light)
    return s


@pytest.mark.parametrize("depth", [0, 1, 3, 5])
def test_reflection_depth_limit(oracle, depth):
    """Q13: exactly max_reflections bounces per pixel, then one blend with the background."""
    s = two_mirror_scene(oracle, depth)
    img, cnt = s.render(counters=True)
    assert cnt["reflect_rays"] == depth * 32 * 24
    assert cnt["primary_rays"] == 32 * 24 and cnt["normals"] == (depth + 1) * 32 * 24
    assert np.isfinite(img).all() and img.min() >= 0 and img.max() <= 1


def test_golden_frames(oracle):
    g = np.load(os.path.join(ROOT, "tests", "golden", "frames_96x72.npz"))
    for key in g.files:
        if key.startswith("cam_"):
            continue
        name, cam = key.split("__")
        s = oracle.load_scene(scene_path(name)).with_size(96, 72)
        assert np.array_equal(s.render(cam=g["cam_" + cam]), g[key]), key
