"""The oracle against the only reference-derived numbers that exist: SURVEY.md's survey-time anchors
(tests/golden/survey_anchors.json).  Formal status: "parity unpinned" -- see oracle/rt_oracle.h."""
import json
import os

import numpy as np
import pytest

from conftest import scene_path

A = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "survey_anchors.json")))


@pytest.mark.parametrize("name", sorted(A["native_checksums"]))
def test_native_resolution_checksums(oracle, name):
    w, h, want = A["native_checksums"][name]
    s = oracle.load_scene(scene_path(name))
    assert (s.width, s.height) == (w, h)
    img = s.render(nthreads=8)
    # SURVEY prints 6 decimals
    assert abs(oracle.checksum(img) - want) < 1e-6 * max(1.0, abs(want)) * 1e-3 + 2e-6


@pytest.mark.parametrize("cfg", A["configs"], ids=lambda c: f"config{c['id']}")
def test_baseline_configs_checksum_and_work_counts(oracle, cfg):
    s = oracle.load_scene(scene_path(cfg["scene"])).with_size(cfg["w"], cfg["h"], cfg["max_reflections"])
    img, cnt = s.render(counters=True, nthreads=8)
    assert abs(oracle.checksum(img) - cfg["checksum"]) < 6e-3  # anchors are printed with 2-3 decimals
    assert cnt["primary_rays"] == cfg["primary"]
    assert cnt["shadow_rays"] == cfg["shadow"]
    assert cnt["reflect_rays"] == cfg["reflect"]
    assert cnt["rays_total"] == cfg["rays"]
    assert cnt["tests"] == cfg["tests"]
    for key, want in cfg["branches"].items():
        assert sum(cnt[k] for k in key.split("+")) == want, key
    for px in A["sample_pixels"]:
        if px["config"] == cfg["id"]:
            got = img[px["y"], px["x"]]
            assert np.allclose(got, np.array(px["rgb"], dtype=np.float32), rtol=3e-8, atol=1e-9), (px, got)


def test_single_thread_equals_multi_thread(oracle):
    s = oracle.load_scene(scene_path("dingdong")).with_size(160, 90)
    assert np.array_equal(s.render(nthreads=1), s.render(nthreads=5))


def test_start_up_camera_is_identity(oracle):
    """src/ray-tracer.cpp:25-32,54-58: position 0, yaw 90, pitch 0 -> identity to ~6e-17."""
    cam = oracle.camera_matrix()
    assert np.abs(cam - np.eye(4).reshape(16)).max() < 1e-15
