"""GPU tests of the wave-per-block ("lean") instantiation of the wavefront kernel: scenes of unit spheres without mirrors,
dense output (rt_wavefront.hip, "the lean path").  It must produce the frames of the general instantiation (RT_FLAG_NOLEAN)
and of the oracle bit for bit; its own-sphere rule (a shadow ray that leaves a sphere towards a light in front of the
surface is not tested against that sphere) is exercised where its window could matter."""
import os

import numpy as np
import pytest

from conftest import scene_path
from test_gpu_parity import oracle_from, render_desc, random_scene

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _lean_always(monkeypatch):
    """rt_create reads MI355RT_LEAN: frames as small as these would otherwise switch to the general instantiation after the first one
    (few tiles with hits: render_impl in rt_capi.cpp); the adaptive choice itself is covered by test_adaptive_choice_of_the_instantiation."""
    monkeypatch.setenv("MI355RT_LEAN", "always")


def _three_way(pkg, oracle, sc, cam=None):
    a = render_desc(pkg, sc, cam)
    assert np.array_equal(a, render_desc(pkg, sc, cam, flags=pkg.RT_FLAG_NOLEAN)), "lean and general instantiation disagree"
    want = oracle_from(pkg, oracle, sc).render(cam=cam, nthreads=8)
    assert np.array_equal(a[..., :3], want, equal_nan=True), "lean instantiation and oracle disagree"
    return a


@pytest.mark.parametrize("seed", range(10))
def test_random_sphere_fields(pkg, oracle, seed):
    n_s = [4, 8, 20, 40, 70, 130, 12, 5, 33, 64][seed]
    sc = random_scene(pkg, 424200 + seed, n_s, 1 + seed % 9, w=150 + 13 * seed, h=100 + 7 * seed, with_plane=False)
    cam = oracle.camera_matrix(pos=(seed * 0.3 - 1.0, 0.5, -4.0), yaw_deg=90.0 + 2 * seed, pitch_deg=-3.0 + seed) if seed % 3 else None
    _three_way(pkg, oracle, sc, cam)


@pytest.mark.parametrize("fmt", ["rgba32f", "rgba8"])
def test_reference_scene_both_formats(pkg, fmt):
    sc = pkg.Scene.load_from_file(scene_path("20spheres")).set_size(500, 300)
    f = pkg.RT_FMT_RGBA8 if fmt == "rgba8" else pkg.RT_FMT_RGBA32F
    a = render_desc(pkg, sc, fmt=f)
    assert np.array_equal(a, render_desc(pkg, sc, flags=pkg.RT_FLAG_NOLEAN, fmt=f))
    assert np.array_equal(a, render_desc(pkg, sc, flags=pkg.RT_FLAG_SIMPLE, fmt=f))


def test_own_sphere_rule_corner_cases(pkg, oracle):
    """Camera inside a sphere (hits from the inside: the un-flipped normal points away from the viewer), tiny and huge radii,
    spheres far from the origin (the window of the rule closes: lanes take the test), overlapping spheres (a hit point inside
    another sphere), lights grazing the surface."""
    def base(w=160, h=120):
        return pkg.Scene.new(w, h, 60.0, 2, (0.2, 0.3, 0.4))
    # camera inside a big sphere that contains four small ones
    s = base()
    s.add_object(pkg.surface_make("sphere", [0, 0, 0], [50.0]), (0.9, 0.8, 0.7))
    for k in range(4):
        s.add_object(pkg.surface_make("sphere", [3 * k - 4.5, 0.5 * k, 12], [1.0 + 0.2 * k]), (0.3, 0.9, 0.5))
    s.add_light("directional", [0.2, -1, 0.3])
    s.add_light("directional", [-1, -0.1, 0.0])
    s.add_light("spherical", [0, 5, 5], (1, 1, 1), 400.0)
    _three_way(pkg, oracle, s)
    # radii from 1e-3 to 1e4
    s = base()
    for k, r in enumerate([1e-3, 1e-2, 0.1, 1.0, 10.0]):
        s.add_object(pkg.surface_make("sphere", [2.5 * k - 5, 0, 6 + 30 * r], [r * 20 if r < 1 else r]), (0.8, 0.4, 0.2))
    s.add_object(pkg.surface_make("sphere", [0, -1e4 - 3, 0], [1e4]), (0.5, 0.5, 0.5))
    for d in ([0.3, -1, 0.2], [1, -0.02, 0], [0, -1, 0], [0, -1e-9, 1]):
        s.add_light("directional", d)
    _three_way(pkg, oracle, s)
    # the same scene a million units away from the origin
    for off in (1e4, 1e6, 3e7):
        s = base(96, 64)
        o = np.array([off, -off, 0.5 * off])
        for k in range(6):
            s.add_object(pkg.surface_make("sphere", o + [2.2 * k - 5.5, 0.3 * k, 14], [1.3]), (0.8, 0.4 + 0.1 * k, 0.2))
        s.add_light("directional", [0.3, -1, 0.2])
        s.add_light("directional", [-0.5, -0.2, 1])
        s.add_light("spherical", o + [0, 8, 6], (1, 1, 1), 500.0)
        cam = pkg.camera_matrix(tuple(o), 90.0, 0.0)
        _three_way(pkg, oracle, s, cam)
    # overlapping and nested spheres
    s = base()
    for k in range(8):
        s.add_object(pkg.surface_make("sphere", [0.9 * k - 3, 0.2 * (k % 3), 10 + 0.5 * (k % 2)], [1.0 + 0.15 * k]), (0.2 + 0.1 * k, 0.5, 0.9 - 0.1 * k))
    s.add_object(pkg.surface_make("sphere", [0, 0, 10], [0.3]), (1, 1, 1))
    for d in ([0.3, -1, 0.2], [-1, -0.3, 0.5], [0.1, 0.1, 1.0], [0, 1, 0]):
        s.add_light("directional", d)
    _three_way(pkg, oracle, s)


def test_counters_of_the_two_instantiations(pkg):
    """Reference-equivalent counts (rays, tests, hits) are the same; the lean instantiation executes fewer sphere tests
    (the own-sphere rule) -- what bench.py's flop accounting reads."""
    sc = pkg.Scene.load_from_file(scene_path("20spheres")).set_size(480, 270)
    out = []
    for fl in (0, pkg.RT_FLAG_NOLEAN):
        r = pkg.Renderer(sc, device=0, flags=fl | pkg.RT_FLAG_COUNT)
        r.update()
        out.append((r.counters(), r.counters_detail()))
        r.cleanup_update()
    for k in ("primary_rays", "shadow_rays", "reflect_rays", "tests", "hits"):
        assert out[0][0][k] == out[1][0][k], (k, out)
    assert out[0][0]["tests_executed"] < out[1][0]["tests_executed"]
    assert out[0][1]["shadow_rays_traced"] == out[1][1]["shadow_rays_traced"]


def test_moving_camera_and_cuts(pkg):
    """Launch-order feedback and tile words under the lean instantiation: an orbit with cuts, every frame equal to a fresh
    general-instantiation render."""
    sc = pkg.Scene.load_from_file(scene_path("20spheres")).set_size(320, 200)
    r = pkg.Renderer(sc, device=0)
    ref = pkg.Renderer(sc, device=0, flags=pkg.RT_FLAG_NOLEAN | pkg.RT_FLAG_STATIC_ORDER | pkg.RT_FLAG_NOSCAN)
    poses = [(0, 0, 0, 90, 0), (1, 0.5, -2, 85, 3), (20, 2, 15, 180, 0), (20, 2, 15, 0, 0), (5, 2, 29, -90, 0), (0, 0, 0, 90, 0), (0, 0, 0, -90, 0), (0, 0, 0, 90, 0)]
    for i, (x, y, z, yaw, pitch) in enumerate(poses * 2):
        cam = pkg.camera_matrix((x, y, z), yaw, pitch)
        r.update(cam)
        ref.update(cam)
        assert np.array_equal(r.download(), ref.download()), f"frame {i}"
    r.cleanup_update()
    ref.cleanup_update()


@pytest.mark.parametrize("flags", [0, 256])
def test_counters_after_a_camera_cut(pkg, oracle, flags):
    """A counting renderer keeps its launch-order lists across frames: after a cut from a view full of hits to an empty one the
    tiles that had hits are still traced by list slots while their index slots find them EMPTY -- each ray must be booked once."""
    from test_gpu_parity import render_cpu
    w, h = 320, 200
    sc = pkg.Scene.load_from_file(scene_path("20spheres")).set_size(w, h)
    r = pkg.Renderer(sc, device=0, flags=flags | pkg.RT_FLAG_COUNT)
    away = pkg.camera_matrix((0.0, 0.0, 0.0), -90.0, 0.0)
    half = pkg.camera_matrix((0.0, 0.0, 0.0), 140.0, 0.0)
    for cam in (None, None, away, away, None, half, away, half, None):
        r.update(cam)
        got = r.counters()
        _, want = render_cpu(oracle, "20spheres", w, h, cam=cam, counters=True)
        for k in ("primary_rays", "shadow_rays", "reflect_rays", "tests"):
            assert got[k] == want[k], (k, got, want)
    r.cleanup_update()


def test_adaptive_choice_of_the_instantiation(pkg, monkeypatch):
    """Without MI355RT_LEAN the host picks the instantiation per frame from the previous frames' count of tiles with hits (lean while the GPU
    is full, the general one -- which splits costly tiles -- while few tiles have hits).  The frames must not notice the switches: a camera
    that moves between a view full of spheres, a sparse one and an empty one, every frame equal to a fresh general-instantiation render."""
    monkeypatch.delenv("MI355RT_LEAN", raising=False)
    sc = pkg.Scene.load_from_file(scene_path("20spheres")).set_size(1920, 1080)
    r = pkg.Renderer(sc, device=0)
    ref = pkg.Renderer(sc, device=0, flags=pkg.RT_FLAG_NOLEAN | pkg.RT_FLAG_STATIC_ORDER)
    poses = [(0, 0, 0, 90, 0)] * 6 + [(19, 2, 15, 180, 0)] * 8 + [(0, 0, 0, -90, 0)] * 4 + [(0, 0, 0, 90, 0)] * 8
    for i, (x, y, z, yaw, pitch) in enumerate(poses):
        cam = pkg.camera_matrix((x, y, z), yaw, pitch)
        r.update(cam)
        ref.update(cam)
        assert np.array_equal(r.download(), ref.download()), f"frame {i}"
    r.cleanup_update()
    ref.cleanup_update()
