"""cubic_guarded (cuda-ray-tracer_amd/csrc/rt_math.hpp) on the CPU: the very header the kernels are compiled from, built for the host
(tests/tools/cubic_guard_lab.cpp) and run against the oracle's intersect_ray on every primary ray and every shadow ray of the
repository's degree-3 scenes and of random ones.  Where the guard answers itself, the decisions the kernels make with the result
(t >= EPS for nearest hits, EPS < t < max_t for shadow rays) must be the oracle's and an accepted nearest-hit root must agree to
1e-7 relative (the guard promises 1e-8); where it does not, the kernels run the reference's dense expansion and solver as before.
The host stand-ins for the device's reciprocal / reciprocal-square-root estimates are made 2^-13 wrong on purpose."""
import os
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))


@pytest.fixture(scope="module")
def lab():
    import cubic_guard_lab as L
    return L, L.build()


@pytest.mark.parametrize("name,max_refused", [("clebsch", 0.03), ("dingdong", 0.4), ("monkey_saddle", 0.02), ("cayley", 1.0), ("cubic", 1.0)])
def test_guard_agrees_with_the_oracle_on_the_repository_scenes(lab, name, max_refused):
    L, lib = lab
    osc = L.O.load_scene(os.path.join(ROOT, "scenes", name + ".yml")).with_size(200, 150)
    st = L.run(lib, osc, None, 0)
    assert st.tests[0] == 200 * 150 and st.tests[1] > 0
    assert st.decision_diff[0] == 0 and st.decision_diff[1] == 0
    assert st.value_diff[0] == 0 and st.worst_rel[0] < 1e-9
    assert st.fallback[0] <= max_refused * st.tests[0]
    if name in ("clebsch", "monkey_saddle"):   # shadow rays are asked for a decision only: the guard hardly ever refuses
        assert st.fallback[1] <= 0.002 * st.tests[1]


def test_cayley_from_the_origin_is_refused_wholesale(lab):
    """F(0) = 0 for scenes/cayley.yml and the camera sits at the origin: every primary ray has a double root at t = 0 and the
    reference's Cardano-or-trigonometric choice hangs on the last bit of its own coefficients -- the guard must not answer."""
    L, lib = lab
    osc = L.O.load_scene(os.path.join(ROOT, "scenes", "cayley.yml")).with_size(96, 64)
    st = L.run(lib, osc, None, 0)
    assert st.fallback[0] == st.tests[0]


def test_guard_agrees_with_the_oracle_on_random_scenes(lab):
    L, lib = lab
    tests = refused = 0
    for seed in range(12):
        osc, cam = L.fuzz_scene(seed)
        st = L.run(lib, osc, cam, 0)
        assert st.decision_diff[0] == 0 and st.decision_diff[1] == 0, seed
        assert st.value_diff[0] == 0 and st.worst_rel[0] < 1e-9, seed
        tests += st.tests[0] + st.tests[1]
        refused += st.fallback[0] + st.fallback[1]
    assert tests > 100000 and refused < 0.1 * tests


def _guard(lib, t3, t2, t1, t0, max_t=1e6, decide=False, m=1e-15):
    import ctypes as C
    lib.lab_guard.restype = C.c_int
    lib.lab_guard.argtypes = [C.c_double] * 4 + [C.POINTER(C.c_double), C.c_double, C.c_int, C.POINTER(C.c_double)]
    lib.lab_reference.restype = C.c_double
    lib.lab_reference.argtypes = [C.c_double] * 4
    mm = (C.c_double * 4)(*([m] * 4 if not isinstance(m, (list, tuple)) else m))
    t = C.c_double(0.0)
    ok = lib.lab_guard(t3, t2, t1, t0, mm, max_t, int(decide), C.byref(t))
    return bool(ok), t.value, lib.lab_reference(t3, t2, t1, t0)


def test_guard_on_hand_made_polynomials(lab):
    """cubic_guarded by itself: what it answers, what it refuses."""
    import math
    L, lib = lab
    # (t - 1)(t - 2)(t - 3) = t^3 - 6 t^2 + 11 t - 6: three roots, the reference returns the smallest one >= EPS
    ok, t, ref = _guard(lib, 1.0, -6.0, 11.0, -6.0)
    assert ok and abs(t - 1.0) < 1e-12 and abs(ref - 1.0) < 1e-12
    # (t + 1)(t - 2)(t - 3): the smallest root is negative, the middle one is taken
    ok, t, ref = _guard(lib, 1.0, -4.0, 1.0, 6.0)
    assert ok and abs(t - 2.0) < 1e-12 and abs(ref - 2.0) < 1e-12
    # one real root (Cardano): (t - 2)(t^2 + t + 5)
    ok, t, ref = _guard(lib, 1.0, -1.0, 3.0, -10.0)
    assert ok and abs(t - 2.0) < 1e-12 and abs(ref - 2.0) < 1e-12
    # ... returned unfiltered when it is negative: (t + 2)(t^2 - t + 5)
    ok, t, ref = _guard(lib, 1.0, 1.0, 3.0, 10.0)
    assert ok and abs(t + 2.0) < 1e-12 and abs(ref + 2.0) < 1e-12
    # a shadow ray asks for the decision only: with the largest root inside (EPS, max_t) the answer is "blocked" whatever root comes back
    ok, t, _ = _guard(lib, 1.0, -6.0, 11.0, -6.0, max_t=1e6, decide=True)
    assert ok and 1e-7 < t < 1e6
    ok, t, _ = _guard(lib, 1.0, -6.0, 11.0, -6.0, max_t=1.5, decide=True)   # the largest root is beyond max_t: the smaller ones decide (1 is inside)
    assert ok and abs(t - 1.0) < 1e-12
    ok, t, _ = _guard(lib, 1.0, 6.0, 11.0, 6.0, decide=True)                # roots -1, -2, -3: nothing in front
    assert ok and t < 0.0
    # double root (discriminant 0): (t - 1)^2 (t - 3) -- Cardano or trigonometric hangs on rounding: refused
    ok, _, _ = _guard(lib, 1.0, -5.0, 7.0, -3.0)
    assert not ok
    # a root sitting on EPS: refused; clearly beside it: answered
    eps = 1e-7
    ok, _, _ = _guard(lib, 1.0, -(eps + 5.0), 6.0 + 5.0 * eps, -6.0 * eps)   # roots eps, 2, 3
    assert not ok
    ok, t, ref = _guard(lib, 1.0, -(0.5 + 5.0), 6.0 + 2.5, -3.0)             # roots 0.5, 2, 3
    assert ok and abs(t - 0.5) < 1e-12 and abs(ref - 0.5) < 1e-12
    # leading coefficient near EPS: refused; exactly 0: the quadratic branch answers
    ok, _, _ = _guard(lib, 1.5e-7, 1.0, -3.0, 2.0)
    assert not ok
    ok, t, ref = _guard(lib, 0.0, 1.0, -3.0, 2.0)                            # t^2 - 3 t + 2: the first candidate (3 - 1) / 2 = 1 >= EPS
    assert ok and abs(t - 1.0) < 1e-12 and abs(ref - 1.0) < 1e-12
    ok, t, ref = _guard(lib, 0.0, 1.0, 0.0, 1.0)                             # no real root: -1
    assert ok and t == -1.0 and ref == -1.0
    ok, t, ref = _guard(lib, 0.0, 0.0, 2.0, -1.0)                            # linear
    assert ok and abs(t - 0.5) < 1e-12 and abs(ref - 0.5) < 1e-15   # (the lab build makes the reciprocal estimate 2^-21 wrong on purpose: one Newton step leaves 2^-42)
    ok, t, ref = _guard(lib, 0.0, 0.0, 0.0, 1.0)                             # constant: -1
    assert ok and t == -1.0 and ref == -1.0
    # coefficients that are not numbers, or hopelessly uncertain ones: refused
    for bad in (float("nan"), float("inf")):
        assert not _guard(lib, bad, 1.0, 1.0, 1.0)[0]
        assert not _guard(lib, 1.0, bad, 1.0, 1.0)[0]
        assert not _guard(lib, 1.0, 1.0, 1.0, bad)[0]
    assert not _guard(lib, 1.0, -6.0, 11.0, -6.0, m=1e-6)[0]                  # the root would be uncertain by far more than 1e-8
    assert _guard(lib, 1.0, -6.0, 11.0, -6.0, m=1e-6, decide=True)[0]         # ... which a decision far from its thresholds does not mind
    assert not _guard(lib, 1.0, -6.0, 11.0, -6.0, m=1e-3, decide=True)[0]     # (uncertain enough and even the discriminant's sign is)
    # random cubics with well separated roots: answered, and equal to the reference's solver to 1e-9
    rng = __import__("numpy").random.default_rng(7)
    answered = 0
    for _ in range(2000):
        r = sorted(rng.uniform(-5, 5, 3))
        if min(r[1] - r[0], r[2] - r[1]) < 0.2 or min(abs(x - 1e-7) for x in r) < 0.05:
            continue
        a = float(rng.uniform(0.5, 2.0)) * (1 if rng.random() < 0.5 else -1)
        t3, t2, t1, t0 = a, -a * sum(r), a * (r[0] * r[1] + r[0] * r[2] + r[1] * r[2]), -a * r[0] * r[1] * r[2]
        ok, t, ref = _guard(lib, t3, t2, t1, t0)
        if ok:
            answered += 1
            assert abs(t - ref) <= 1e-9 * max(1.0, abs(ref)), (r, t, ref)
    assert answered > 1000
