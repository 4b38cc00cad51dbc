"""bench.py's launch contract: `--gpus N` really runs N ranks (or fails loudly), and the RCCL code path (process group,
async dist.gather, stream-side wait, double buffers, device reassembly; dense rows and sparse tiles) executes on the
1-GPU box with one rank."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env=None, timeout=900):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, env=e, timeout=timeout)


def test_world_size_mismatch_is_refused():
    p = _run(["--gpus", "2"], env={"WORLD_SIZE": "4"})
    assert p.returncode != 0 and "WORLD_SIZE=4" in p.stderr and p.stdout.strip() == ""
    p = _run(["--gpus", "1"], env={"WORLD_SIZE": "2"})
    assert p.returncode != 0 and p.stdout.strip() == ""


def test_gpus_n_spawns_n_ranks_and_propagates_failure():
    """Without a GPU (this container) the N rank processes each refuse to run: the supervisor must report failure and
    print no result line.  (On the GPU box the same holds when N exceeds the GPUs present: see the gpu test below.)"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("CPU-only check")
    p = _run(["--gpus", "3", "--steps", "1", "--warmup", "0"])
    assert p.returncode != 0
    assert p.stdout.strip() == ""
    assert p.stderr.count("bench.py needs a GPU") >= 1 and "stopping the other ranks" in p.stderr


@pytest.mark.gpu
def test_more_ranks_than_gpus_fails_loudly():
    import torch
    n = torch.cuda.device_count() + 1
    p = _run(["--gpus", str(n), "--steps", "2", "--warmup", "1", "--no-cpu-baseline"])
    assert p.returncode != 0, p.stdout
    assert "n_gpus" not in p.stdout
    assert "refusing to share a GPU" in p.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("gather", ["dense", "sparse"])
def test_rccl_path_with_one_rank(gather):
    """backend nccl (= RCCL) at world size 1: the very step() / assemble() / flush() code of the N > 1 bench, one rank."""
    p = _run(["--gpus", "1", "--force-dist", "--steps", "6", "--warmup", "3", "--gather", gather, "--no-cpu-baseline"])
    assert p.returncode == 0, p.stderr[-3000:]
    r = json.loads(p.stdout.strip().splitlines()[-1])
    assert r["n_gpus"] == 1 and r["rccl_ranks"] == 1
    # the distributed path measures BASELINE config 5 (the 8K frame) and carries its own 1-GPU denominator and the link bound
    assert "config5" in r["config"]["workload"] and "7680x4320" in r["config"]["workload"] and r["scaling"] == "strong"
    assert r["one_gpu_same_frame"]["rgba32f_ms"] > 0 and r["one_gpu_same_frame"]["rgba8_ms"] > 0
    sp = r["speedup_vs_1gpu_config5"]
    assert 0.2 < sp["rgba32f_dense"] < 1.5 and sp["rgba8_sparse" if gather == "sparse" else "rgba8"] > 0   # one rank: about 1, less the transport
    assert r["links_into_rank0"]["gather_ingest_gbs"] > 0 and r["gather_bound_ms"]["rgba32f_dense"] == 0.0   # (world 1: nothing arrives from other ranks)
    assert set(r["predicted_rgba32f_dense"]["table"]) >= {"n2_link50GBs", "n8_link75GBs"}
    assert r["config"]["framebuffer_format"] == "rgba32f"
    assert r["config"]["gathered_frame_identical_to_single_gpu_frame"] is True
    assert r["config"]["gather"]["kind"] == "dense rows"
    alt = r["config"]["alt"]
    assert alt["framebuffer_format"] == "rgba8" and alt["gathered_frame_identical_to_single_gpu_frame"] is True
    assert alt["gather"]["kind"] == ("sparse tiles" if gather == "sparse" else "dense rows")
    assert r["roofline"]["kernel_ms"] > 0 and alt["kernel_ms"] > 0


@pytest.mark.gpu
def test_sparse_overflow_is_resent_densely():
    """A message capacity that is too small: the overflowed frames are sent again as dense rows and counted, and the frame
    rank 0 ends up with is still the single-context frame."""
    p = _run(["--gpus", "1", "--force-dist", "--steps", "4", "--warmup", "2", "--sparse-capacity", "100", "--no-cpu-baseline"])
    assert p.returncode == 0, p.stderr[-3000:]
    r = json.loads(p.stdout.strip().splitlines()[-1])
    alt = r["config"]["alt"]
    assert alt["gather"]["frames_resent_densely"] >= 1
    assert alt["gathered_frame_identical_to_single_gpu_frame"] is True


@pytest.mark.gpu
@pytest.mark.parametrize("graph", [True, False])
def test_timed_region_as_one_graph_or_as_plain_launches(graph):
    """N = 1: the K timed frames are launched as one captured hipGraph (default) or one by one (--no-graph); either way the JSON
    line says which, and the last frame of the timed region equals a plain launch of the same frame."""
    p = _run(["--steps", "7", "--warmup", "3", "--no-orbit", "--no-cpu-baseline"] + ([] if graph else ["--no-graph"]))
    assert p.returncode == 0, p.stderr[-3000:]
    r = json.loads(p.stdout.strip().splitlines()[-1])
    roof = r["roofline"]
    assert roof["timed_region"].startswith("one hipGraph of the K = 7 frame launches" if graph else "K launches")
    assert roof["timed_region_last_frame_identical_to_plain_launch"] is True
    assert r["steps"] == 7 and r["n_gpus"] == 1 and roof["kernel_ms"] > 0
    # all five BASELINE configs on this GPU in the same line, and what update() returns for one synchronised frame
    assert [c["config"] for c in r["configs"]] == ["config1", "config2", "config3", "config4", "config5"]
    for c in r["configs"]:
        assert c["ms_per_step"] > 0 and c["single_frame_ms"] > 0 and c["value"] > 0 and 0 < c["roofline"]["frac"] < 1
    assert r["single_frame_ms"] == r["configs"][1]["single_frame_ms"]
    assert "config2" in r["config"]["workload"] and "1920x1080" in r["config"]["workload"]
