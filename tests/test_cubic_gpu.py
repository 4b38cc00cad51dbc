"""Degree-3 surfaces on the GPU: the guarded Taylor path (rt_math.hpp, cubic_guarded) as the kernels run it.
The CPU side of the same function is tests/test_cubic_guard.py; parity of whole frames is in test_gpu_parity.py / test_full_size.py."""
import numpy as np
import pytest

from conftest import compare, scene_path

pytestmark = pytest.mark.gpu


def _detail(pkg, name, w, h, cam=None, flags=0):
    sc = pkg.Scene.load_from_file(scene_path(name)).set_size(w, h)
    r = pkg.Renderer(sc, device=0, flags=flags | pkg.RT_FLAG_COUNT)
    r.update(cam)
    img = r.download().copy()
    d, c = r.counters_detail(), r.counters()
    r.cleanup_update()
    return img, c, d


def test_guard_refuses_little_on_clebsch_and_everything_degenerate_on_cayley(pkg, oracle):
    """clebsch: a per cent or two of the primary rays (their root is asked for and must be sharp), practically no shadow ray (asked
    for a decision only).  cayley seen from the origin: F(0) = 0, every primary ray has a double root at t = 0 -- all of them go to the
    reference's dense path; its shadow rays (axis-parallel lights: the degree-2 branch) are answered by the guard."""
    w, h = 400, 300
    _, c, d = _detail(pkg, "clebsch", w, h)
    assert d["executed_by_class"]["cubic"] > w * h and d["cubic_points"] == c["hits"]    # one Taylor record per hit (phase A'), none for the frame's origin
    assert 0 < d["cubic_refused"] < 0.03 * d["executed_by_class"]["cubic"]
    assert d["cubic_refused"] < 0.05 * w * h                                                # (all of them primary rays)
    _, c, d = _detail(pkg, "cayley", w, h)
    assert w * h <= d["cubic_refused"] < w * h + 0.001 * d["executed_by_class"]["cubic"]
    cam = oracle.camera_matrix(pos=(0.3, 0.2, -4.0), yaw_deg=90.0, pitch_deg=0.0)          # ... and seen from elsewhere it is an ordinary cubic
    _, c, d = _detail(pkg, "cayley", w, h, cam=cam)
    assert d["cubic_refused"] < 0.1 * d["executed_by_class"]["cubic"]


@pytest.mark.parametrize("name", ["clebsch", "cayley", "dingdong", "monkey_saddle", "cubic"])
def test_counting_render_equals_product_render_and_branch_counts_are_the_oracles(pkg, oracle, name):
    """The counting instantiation inlines the guarded path, the product calls it out of line: same frame.  The solver-branch counters
    (classified on the dense coefficients) stay the oracle's whether the guard answered or not."""
    w, h = 240, 180
    sc = pkg.Scene.load_from_file(scene_path(name)).set_size(w, h)
    r = pkg.Renderer(sc, device=0)
    r.update()
    plain = r.download().copy()
    r.cleanup_update()
    counted, c, d = _detail(pkg, name, w, h)
    assert np.array_equal(plain, counted)
    _, ocnt = oracle.load_scene(scene_path(name)).with_size(w, h).render(counters=True, nthreads=8)
    assert c["primary_rays"] == ocnt["primary_rays"] and c["shadow_rays"] == ocnt["shadow_rays"]


def test_fast_build_uses_the_same_guard(pkg, oracle):
    """RT_FLAG_FAST: the guarded path is shared (its own arithmetic uses explicit fma either way); only the dense fall-back is contracted."""
    w, h = 320, 240
    sc = pkg.Scene.load_from_file(scene_path("clebsch")).set_size(w, h)
    r = pkg.Renderer(sc, device=0, flags=pkg.RT_FLAG_FAST)
    r.update()
    got = r.download().copy()
    r.cleanup_update()
    want = oracle.load_scene(scene_path("clebsch")).with_size(w, h).render(nthreads=8)
    c = compare(got[..., :3], want)
    assert c["n_bad_pixels"] <= max(2, int(0.0004 * w * h)), c


def test_many_cubic_objects_and_mirrors(pkg, oracle):
    """More degree-3 objects than the frame arguments carry data for (RT_CUB_AT_MAX = 4: objects 5 and 6 form their CubicAbs in the lane's
    working record), next to spheres and a mirror plane -- bounce rays form their Taylor records per lane.  Wavefront == simple kernel bit for
    bit (they share the guarded path), both within 1e-5 of the oracle."""
    from test_gpu_parity import oracle_from, render_desc
    rng = np.random.default_rng(4242)
    w, h = 200, 150
    s = pkg.Scene.new(w, h, 50.0, 3, (0.1, 0.2, 0.3))
    for k in range(6):
        q = np.zeros(20)
        q[:10] = rng.uniform(-0.3, 0.3, 10) * (rng.random(10) < 0.6)
        q[10:13] = rng.uniform(0.5, 1.5, 3)           # mostly an ellipsoid, bent by the cubic terms
        c = np.array([-5.0 + 2.0 * k, float(rng.uniform(-1, 1)), 6.0 + 0.8 * k])
        # shift the surface to c:  F(x - c)  expanded only approximately is fine for a test -- use the linear and constant terms
        q[16:19] = -2.0 * q[10:13] * c
        q[19] = float(np.sum(q[10:13] * c * c) - 1.0)
        s.add_object(q, rng.uniform(0.2, 1, 3), 0.4 if k % 2 else 0.0)
    s.add_object(pkg.surface_make("sphere", [0.0, 1.5, 9.0], [1.0]), (0.9, 0.2, 0.2), 0.5)
    e = np.zeros(20)   # an ellipsoid (general quadric class): 2 (x + 3)^2 + 0.5 (y - 2)^2 + (z - 7)^2 = 1 -- the instantiation with every feature
    e[10:13] = (2.0, 0.5, 1.0)
    e[16:19] = (12.0, -2.0, -14.0)
    e[19] = 18.0 + 2.0 + 49.0 - 1.0
    s.add_object(e, (0.2, 0.4, 0.9), 0.3)
    s.add_object(pkg.surface_make("plane", [0, -2.5, 0], [0.0, 1.0, 0.0]), (0.5, 0.5, 0.5), 0.3)
    s.add_light("directional", [0.3, -1.0, 0.4], (1, 1, 1), 1.0)
    s.add_light("spherical", [0.0, 8.0, 2.0], (1, 0.9, 0.8), 200.0)
    got = render_desc(pkg, s)
    assert np.array_equal(got, render_desc(pkg, s, flags=pkg.RT_FLAG_SIMPLE))
    want = oracle_from(pkg, oracle, s).render(nthreads=8)
    c = compare(got[..., :3], want)
    assert c["n_bad_pixels"] <= max(3, int(0.002 * w * h)), c
