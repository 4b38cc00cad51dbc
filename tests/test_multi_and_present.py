"""Several GPUs behind the boundary (rt_create_multi / rt_render_multi, and update() with MI355RT_DEVICES) and the
presentation row of SURVEY.md 8(f): presenter hook, RGBA8 through update.h, PPM sink.

On the one-GPU box the device list repeats device 0 (same choreography with device copies instead of RCCL) and the
RCCL calls are exercised with one rank sending its rows to itself (RT_MULTI_SELF_EXCHANGE); the 8-GPU run is the
driver's.  Every frame must equal the single-context frame bit for bit."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, scene_path

EXE = os.path.join(ROOT, "tests", "host_driver", "update_driver")


def test_band_and_slot_arithmetic_of_the_multi_layout(pkg):
    """CPU: context q = part * n + r is rank q of a world of n * parts; its rows and its slot in the rank-major gather
    are those of sharding.py (what rt_assemble inverts)."""
    for h, band, n, parts in ((1080, 16, 8, 2), (4320, 16, 8, 1), (250, 8, 3, 2), (37, 5, 2, 3)):
        world = n * parts
        idx = pkg.assemble_index(h, band, world)
        mx = pkg.max_local_rows(h, band, world)
        seen = np.zeros(h, dtype=bool)
        for p in range(parts):
            for r in range(n):
                q = p * n + r
                rows = pkg.band_rows_of_rank(h, band, world, q)
                assert np.array_equal(idx[rows], q * mx + np.arange(len(rows)))
                seen[rows] = True
        assert seen.all()
        per_dev = [sum(len(pkg.band_rows_of_rank(h, band, world, p * n + r)) for p in range(parts)) for r in range(n)]
        assert max(per_dev) - min(per_dev) <= band * parts


@pytest.mark.gpu
def test_one_device_one_part_equals_rt_render(pkg):
    sc = pkg.Scene.load_from_file(scene_path("reflection_test")).set_size(333, 197).set_max_reflections(4)
    ref = pkg.Renderer(sc, device=0)
    ref.update()
    want = ref.download()
    m = pkg.MultiRenderer(sc, [0], parts=1)
    assert m.transport == "in place" and m.n_contexts == 1
    assert m.update() > 0.0
    assert np.array_equal(m.download(), want)
    m.cleanup_update()


@pytest.mark.gpu
@pytest.mark.parametrize("devices,parts,band,fmt", [([0, 0, 0], 1, 8, 0), ([0, 0], 2, 16, 0), ([0, 0, 0, 0], 2, 16, 1), ([0], 3, 16, 0)])
def test_repeated_device_choreography(pkg, devices, parts, band, fmt):
    """n "devices" (the same GPU) x parts contexts, streams, events, copies into rank-major slots, rt_assemble: three frames
    with a moving camera, each equal to the single-context frame."""
    w, h = 400, 277
    sc = pkg.Scene.load_from_file(scene_path("20spheres")).set_size(w, h)
    ref = pkg.Renderer(sc, device=0, fmt=fmt)
    m = pkg.MultiRenderer(sc, devices, band_rows=band, parts=parts, fmt=fmt)
    assert m.n_contexts == len(devices) * parts
    assert m.transport == ("device copies" if len(devices) > 1 else "in place")
    for k in range(3):
        cam = pkg.camera_matrix((0.5 * k, 0.2 * k, -1.0 * k), 90.0 + 4.0 * k, 1.0 * k)
        ref.update(cam)
        m.update(cam, timed=(k != 1))   # the middle frame enqueue-only
        assert np.array_equal(m.download(), ref.download()), k
    m.cleanup_update()


@pytest.mark.gpu
@pytest.mark.parametrize("devices,parts,band,fmt,size", [([0, 0, 0], 1, 8, 0, (400, 277)), ([0, 0, 0, 0], 2, 16, 0, (400, 277)), ([0, 0, 0, 0], 2, 16, 1, (333, 256)),
                                                         ([0], 3, 16, 0, (400, 277)), ([0, 0], 4, 4, 0, (129, 31))])
def test_bandwise_transport_matches_single_context(pkg, devices, parts, band, fmt, size):
    """RT_MULTI_BANDWISE: every context's rows go band by band (one strided copy per context on this one-GPU box) straight to their place in
    the full frame -- no rank-major slots, no rt_assemble.  Heights that are and are not multiples of the band (a shorter last band), more
    contexts than bands on some ranks, both formats; three frames with a moving camera, each equal to the single-context frame, and equal
    to what the classic transport delivers."""
    w, h = size
    sc = pkg.Scene.load_from_file(scene_path("20spheres")).set_size(w, h)
    ref = pkg.Renderer(sc, device=0, fmt=fmt)
    m = pkg.MultiRenderer(sc, devices, band_rows=band, parts=parts, fmt=fmt, flags=pkg.RT_MULTI_BANDWISE)
    classic = pkg.MultiRenderer(sc, devices, band_rows=band, parts=parts, fmt=fmt)
    assert m.n_contexts == len(devices) * parts
    for k in range(3):
        cam = pkg.camera_matrix((0.5 * k, 0.2 * k, -1.0 * k), 90.0 + 4.0 * k, 1.0 * k)
        ref.update(cam)
        m.update(cam, timed=(k != 1))   # the middle frame enqueue-only
        classic.update(cam)
        got = m.download()
        assert np.array_equal(got, ref.download()), k
        assert np.array_equal(got, classic.download()), k
    m.cleanup_update()
    classic.cleanup_update()


@pytest.mark.gpu
def test_bandwise_rccl_self_exchange_in_a_plain_host_process(pkg, oracle, tmp_path):
    """The same through RCCL: one rank sends every band of its rows to itself with ncclSend / ncclRecv pairs in one group per part, the
    receives pointing into the final frame (MI355RT_MULTI_SELF=1 MI355RT_MULTI_BANDWISE=1; a process without PyTorch)."""
    assert os.path.exists(EXE)
    out = str(tmp_path / "f.f32")
    w, h = 320, 203
    env = dict(os.environ, MI355RT_DEVICES="0", MI355RT_MULTI_SELF="1", MI355RT_MULTI_BANDWISE="1", MI355RT_PARTS="2")
    p = subprocess.run([EXE, scene_path("20spheres"), str(w), str(h), "-1", out, "--frames", "3"], capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    got = np.fromfile(out, dtype=np.float32).reshape(h, w, 4)
    want = oracle.load_scene(scene_path("20spheres")).with_size(w, h).render(nthreads=8)
    assert np.array_equal(got[..., :3], want)


@pytest.mark.gpu
def test_rccl_self_exchange_in_a_plain_host_process(pkg, oracle, tmp_path):
    """update() over `MI355RT_DEVICES=0` + MI355RT_MULTI_SELF=1: libmi355rt_multi.so creates a one-rank RCCL communicator and the
    rows travel through ncclSend / ncclRecv before rt_assemble (a process without PyTorch: the system's librccl)."""
    assert os.path.exists(EXE)
    out = str(tmp_path / "f.f32")
    w, h = 320, 200
    env = dict(os.environ, MI355RT_DEVICES="0", MI355RT_MULTI_SELF="1", MI355RT_PARTS="2")
    p = subprocess.run([EXE, scene_path("20spheres"), str(w), str(h), "-1", out, "--frames", "3"], capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    got = np.fromfile(out, dtype=np.float32).reshape(h, w, 4)
    want = oracle.load_scene(scene_path("20spheres")).with_size(w, h).render(nthreads=8)
    assert np.array_equal(got[..., :3], want)


@pytest.mark.gpu
def test_update_h_over_a_device_list(pkg, oracle, tmp_path):
    out = str(tmp_path / "f.f32")
    w, h = 256, 192
    env = dict(os.environ, MI355RT_DEVICES="0,0,0", MI355RT_PARTS="2", MI355RT_BAND_ROWS="8")
    p = subprocess.run([EXE, scene_path("reflection_test"), str(w), str(h), "4", out, "--frames", "2"], capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    got = np.fromfile(out, dtype=np.float32).reshape(h, w, 4)
    want = oracle.load_scene(scene_path("reflection_test")).with_size(w, h, 4).render(nthreads=8)
    assert np.array_equal(got[..., :3], want)


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", ["rgba32f", "rgba8"])
def test_presenter_and_ppm_sink(pkg, oracle, tmp_path, fmt):
    """mi355rt_set_presenter[_rgba8]: the callback receives exactly the frame rt_download returns (texture name and size passed
    through); --ppm writes the quantised image top row first; MI355RT_FORMAT=rgba8 is honoured by init_update."""
    out, ppm = str(tmp_path / "f.bin"), str(tmp_path / "f.ppm")
    w, h = 200, 150
    env = dict(os.environ, MI355RT_FORMAT=fmt)
    p = subprocess.run([EXE, scene_path("20spheres"), str(w), str(h), "-1", out, "--present", "--ppm", ppm, "--frames", "2"], capture_output=True, text=True,
                       env=env, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    raw, presented = open(out, "rb").read(), open(out + ".present", "rb").read()
    assert raw == presented and len(raw) == w * h * (4 if fmt == "rgba8" else 16)
    want = oracle.load_scene(scene_path("20spheres")).with_size(w, h).render(nthreads=8)
    q = np.floor(want * 255.0 + 0.5).astype(np.int32)
    if fmt == "rgba8":
        got = np.frombuffer(raw, dtype=np.uint8).reshape(h, w, 4)
        assert np.all(got[..., 3] == 255) and np.abs(got[..., :3].astype(np.int32) - q).max() <= 1
    else:
        got = np.frombuffer(raw, dtype=np.float32).reshape(h, w, 4)
        assert np.array_equal(got[..., :3], want)
    data = open(ppm, "rb").read()
    header = f"P6\n{w} {h}\n255\n".encode()
    assert data.startswith(header) and len(data) == len(header) + w * h * 3
    img = np.frombuffer(data[len(header):], dtype=np.uint8).reshape(h, w, 3)[::-1]   # back to bottom row first
    assert np.abs(img.astype(np.int32) - q).max() <= 1
