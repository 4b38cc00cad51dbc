#!/usr/bin/env python3
"""bench.py -- throughput of the per-pixel ray-tracing hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: this process only starts N rank processes)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one frame: every pixel's primary ray, nearest hit over all surfaces, shadow rays to all lights,
Lambert shading and the mirror-bounce loop, written to a framebuffer resident in HBM.
Metric (BASELINE.json): Mrays/s on scenes/20spheres.yml; a ray = one primary, shadow or reflection ray
(SURVEY.md 8(d)); rays per frame are counted by the kernel itself (RT_FLAG_COUNT pass before timing).

N = 1: BASELINE config 2, 20spheres @ 1920x1080, camera = identity (the reference host's start-up pose), RGBA32F; the line
also carries a `configs` array with all five BASELINE configs on this GPU (short regions) and `single_frame_ms`, what update()
returns for one synchronised frame (src/update-cuda.cu:178-189).
N > 1: BASELINE config 5, 20spheres @ 7680x4320 -- the configuration BASELINE.json names for several GPUs -- rows band-cyclic
over the N ranks (no data-path communication while rendering), one RCCL gather to rank 0 per frame + a device reassembly
kernel on rank 0, as BASELINE.json's north_star prescribes: total work is fixed ("scaling": "strong").  Rank 0 first renders
the whole 8K frame alone, so that the line carries its own denominator (`speedup_vs_1gpu_config5`), and the ranks measure the
link bandwidth into rank 0 (`gather_bound_ms`).  The headline `value` is measured on the SAME output as N = 1 (RGBA32F, dense
rows); the RGBA8 / sparse-tile transport is timed in a second region and reported as `config.alt`.  (--workload config2w keeps
the weak-scaled 1080p workload of rounds 1-2.)

Prints ONE JSON line (rank 0).  The oracle (oracle/) is used ONLY for the `cpu_baseline` legs.
"""
import argparse
import hashlib
import json
import math
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_VECTOR_PEAK_TFLOPS = 78.6   # MI355X: 256 CU x 4 SIMD x 16 FP64 lanes/clk x 2 flop x 2.4 GHz = half the FP32 vector
                                 # peak (157.3 TF, MI355X_MICROARCH.md chip table); checked by profiles/*fp64_peak*
HBM_PEAK_GBS = 8000.0
DIV_FLOPS_IN_ISA, SQRT_FLOPS_IN_ISA = 12.0, 14.0   # FP64 instruction-flops of hipcc's expansion of one division / square root (see roofline.table_in_pmc_terms)
PMC_SUMMARY = os.path.join("profiles", "r03_pmc_summary.txt")
KERNEL_STATS = "profiles/r03_kernel_stats.json"   # tools/kernel_stats_summarize.py: rocprofv3 --kernel-trace --stats of this command, stamped with the kernel sources' digest
CONFIGS = {"config1": ("quadratic", 640, 480, None), "config2": ("20spheres", 1920, 1080, None), "config3": ("reflection_test", 1920, 1080, 4),
           "config4": ("clebsch", 3840, 2160, None), "config5": ("20spheres", 7680, 4320, None)}
KERNEL_SOURCES = ["cuda-ray-tracer_amd/csrc/rt_wavefront.hip", "cuda-ray-tracer_amd/csrc/rt_math.hpp",
                  "cuda-ray-tracer_amd/csrc/rt_scene_dev.h", "cuda-ray-tracer_amd/csrc/rt_wavefront_math.hpp"]


# ----------------------------------------------------------------------------------------------------------------
# --gpus N > 1 without a launcher: start the N rank processes ourselves
# ----------------------------------------------------------------------------------------------------------------
def spawn_ranks(n):
    """Start N fresh copies of this script, one per GPU, BEFORE anything in this process imports torch or touches HIP
    (the parent stays a plain supervisor: no exec of a GPU-initialised process, no fork of one).  Rank 0's JSON line goes
    to our stdout; any rank failing makes the whole run fail."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    live = list(procs)
    while live:
        time.sleep(0.2)
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                print(f"bench.py: rank process {p.pid} exited with code {code}; stopping the other ranks", file=sys.stderr)
                for q in live:   # exactly the processes started above
                    q.terminate()
    return rc


# ----------------------------------------------------------------------------------------------------------------
# workloads
# ----------------------------------------------------------------------------------------------------------------
def workload_for(n_gpus, name, dist_on=False):
    """auto: BASELINE config 2 on one GPU, BASELINE config 5 (the 8K frame) as soon as the frame is distributed.  config2w: the
    weak-scaled 1080p frame (N x the pixels) of rounds 1-2."""
    if name == "auto":
        name = "config5" if (n_gpus > 1 or dist_on) else "config2"
    if name in CONFIGS:
        return (name,) + CONFIGS[name]
    if name != "config2w":
        raise SystemExit(f"bench.py: unknown --workload {name}")
    s = math.sqrt(n_gpus)
    w = int(round(1920 * s / 16.0)) * 16
    h = int(round(w * 9 / 16.0))
    return "config2w", "20spheres", w, h, None


def orbit_pose(pkg, i, n=24):
    """Pose i of n on an orbit around the 20spheres scene (tools/flythrough_bench.py): the reference host's camera
    (src/ray-tracer.cpp:44-58) looking at (5, 2, 15) from 14 units away."""
    import numpy as np
    a = 2.0 * np.pi * i / n
    pos = (5.0 + 14.0 * np.sin(a), 2.0 + 2.0 * np.sin(2 * a), 15.0 - 14.0 * np.cos(a))
    yaw = float(np.degrees(np.arctan2(15.0 - pos[2], 5.0 - pos[0])))
    pitch = float(-np.degrees(np.arctan2(pos[1] - 2.0, 14.0)))
    return pkg.camera_matrix(pos, yaw, pitch)


# ----------------------------------------------------------------------------------------------------------------
# flop accounting
# ----------------------------------------------------------------------------------------------------------------
def load_flop_table():
    """FP64 operations per unit of work of the KERNEL'S OWN algorithm.  The figures come from tools/count_flops.cpp, which
    runs the kernel's math (rt_math.hpp, rt_wavefront_math.hpp -- the very headers the kernel is compiled from) over an
    operation-counting scalar: the technique SURVEY.md 8(d) prescribes (add / sub / mul / div / sqrt = 1 each, strict build,
    so no FMA).  The reference's dense as-written count is 286 + solver per test; these are far below it because absent
    coefficient groups, shared per-ray monomials and culling remove work -- which is why `achieved` must not be computed
    from the dense figure, and is capped by it."""
    path = os.path.join(ROOT, "profiles", "flop_table.json")
    with open(path) as fh:
        return json.load(fh), os.path.relpath(path, ROOT)


def dense_reference_flops(cnt, table):
    """The reference's as-written FP64 count for the frame: 286 per intersect_ray expansion + its solver branch
    (linear 1, quadratic miss 4 / hit 8, Cardano 26, trig 39) + 79 per normal_vector.  Cubic tests by the branch the device
    counted; a non-cubic test is a quadratic miss unless the kernel found a root worth solving (then: hit)."""
    r = table["reference_dense"]
    cb = cnt["cubic_branches"]
    n_cubic = sum(cb.values())
    solved = cnt["solves_by_class"]["unitsq"] + cnt["solves_by_class"]["quadric"]
    other = max(0, cnt["tests"] - n_cubic)
    f = r["expansion"] * cnt["tests"] + r["quadratic_miss"] * other + (r["quadratic_hit"] - r["quadratic_miss"]) * min(solved, other)
    f += r["cardano"] * cb["cardano"] + r["trig"] * cb["trig"] + r["quadratic_hit"] * cb["quad"] + r["linear"] * cb["linear"]
    return float(f + r["normal_vector"] * cnt["hits"])


def ex_objects_unitsq(arr):
    return sum(1 for c in object_classes(arr) if c == "unitsq")


def algorithmic_flops(cnt, table, arr):
    """FP64 operations one frame of the wavefront kernel executes: executed units (device counters of what the product build
    executes, per surface class / culling kind / solver branch) x the counted cost of each unit (profiles/flop_table.json)."""
    u = table["units"]
    ds = table.get("div_sqrt", {})
    ex, so, cu, cb = cnt["executed_by_class"], cnt["solves_by_class"], cnt["cull_by_kind"], cnt["cubic_branches"]
    n_div = n_sqrt = 0.0

    def unit(name, count, weight=1.0):   # flops of `count` units; their divisions / square roots are tallied on the side
        nonlocal n_div, n_sqrt
        d, q = ds.get(name, [0.0, 0.0])
        n_div += count * weight * d
        n_sqrt += count * weight * q
        return count * weight * u[name]

    cross = "_cross" if (ex["quadric"] or ex["cubic"]) else ""       # scenes with general quadrics / cubics form the mixed monomials too
    kinds = arr["light_is_spherical"]
    n_l = max(1, len(kinds))
    f_pt = float(sum(1 for k in kinds if k)) / n_l                      # share of point lights (shadow rays are hits x lights)
    f = 0.0
    # degree-3 surfaces: F(o + t d) from the surface's Taylor data at the ray origin (once per hit, `cubic_points`) through the guarded solver
    # (rt_math.hpp, cubic_guarded: its own operation counts per branch; shadow rays take the shorter "decide" form of the trigonometric
    # branch); the tests the guard hands back (`cubic_refused`) pay the reference's dense expansion and solver on top
    f += ex["unitsq"] * u["test_unitsq"] + ex["quadric"] * u["test_quadric"] + ex["linear"] * u["test_linear"] + ex["cubic"] * u["test_cubic_expand"]
    f += cnt.get("cubic_points", 0) * u.get("cubic_point", 0)
    f += unit("solve_unitsq", so["unitsq"]) + unit("solve_quadric", so["quadric"]) + unit("solve_linear", so["linear"])
    n_cub_tests = max(1, ex["cubic"])
    share_primary = min(1.0, cnt["primary_rays_formed"] * sum(1 for c in object_classes(arr) if c == "cubic") / n_cub_tests) if ex["cubic"] else 0.0
    f += cb["cardano"] * u["cubic_guarded_cardano"] + cb["trig"] * (share_primary * u["cubic_guarded_trig"] + (1.0 - share_primary) * u["cubic_guarded_trig_decide"])
    f += cb["quad"] * u["cubic_guarded_quadratic"] + cb["linear"] * u["cubic_guarded_linear"]
    refused = cnt.get("cubic_refused", 0)
    if refused:
        mean_ref_solver = (cb["cardano"] * u["cubic_cardano"] + cb["trig"] * u["cubic_trig"] + cb["quad"] * u["cubic_quadratic"] + cb["linear"] * u["cubic_linear"]) / n_cub_tests
        f += refused * (u["test_cubic_dense"] + mean_ref_solver)
    n_us = max(1, ex_objects_unitsq(arr))
    f += cu["tile"] * (u["cull_tile"] + u["tile_planes"] * 4.0 / max(4.0, float(n_us)))   # a classifying lane forms its tile's planes once for n_us / 4 spheres
    f += unit("cull_primary", cu["primary"]) + cu["shadow_directional"] * u["cull_shadow_directional"]
    f += cu["shadow_point"] * u["cull_shadow_point"] + cu["records"] * u["cull_record"]
    f += unit("primary_ray" + cross, cnt["primary_rays_formed"]) + cnt["reflect_rays"] * u["reflect_ray" + cross]   # only traced tiles form primary rays
    f += cnt["shadow_rays"] * ((1.0 - f_pt) * u["shadow_ray_considered_directional"] + f_pt * u["shadow_ray_considered_point"])
    f += cnt["shadow_rays_traced"] * ((1.0 - f_pt) * u["shadow_ray_traced_directional" + cross] + f_pt * u["shadow_ray_traced_point" + cross])
    f += cnt["hit_lights_shaded"] * (1.0 - f_pt) * u["shade_directional"] + unit("shade_point", cnt["hit_lights_shaded"], f_pt)
    spheres_only = not (ex["quadric"] or ex["linear"] or ex["cubic"]) and len(arr["coefs"]) == n_us
    f += unit("hit_spheres_only" if spheres_only else "hit" + cross, cnt["hits"]) + unit("chunk_ball", cnt["hits"] / 64.0)
    return f, n_div, n_sqrt


def kernel_source_digest():
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        p = os.path.join(ROOT, rel)
        if os.path.exists(p):
            h.update(open(p, "rb").read())
    return h.hexdigest()[:16]


def rocprof_kernel_avg_ms(args, world):
    """Average duration of the product kernel in the committed rocprofv3 --kernel-trace --stats summary of this command (same
    workload only), and whether the summary still belongs to the kernel sources; None when the file is missing.  Reported next
    to this run's own HIP-event figure, never instead of it."""
    if world != 1 or args.workload_name != "config2" or args.kernel != "wavefront" or args.mode != "strict" or args.format != "rgba32f" or args.camera != "static":
        return None
    try:
        with open(os.path.join(ROOT, KERNEL_STATS)) as f:
            st = json.load(f)
        return {"avg_ms": float(st["average_ns"]) * 1e-6, "stale": st.get("kernel_source_digest") != kernel_source_digest(), "kernel": st.get("kernel"),
                "launches": st.get("calls")}
    except Exception:
        return None


def pmc_profile(args, world):
    """Counters of the dominant kernel from the committed PMC passes (separate `rocprofv3 --pmc` runs of this same command,
    tools/pmc_profile.sh; FETCH_SIZE / WRITE_SIZE in KiB, FETCH_SIZE doubled as the MI355X guide prescribes for gfx950).
    NOT measured by this run: the source is named in the JSON line, and the figures are dropped when the kernel sources
    have changed since the passes were taken or when this is not the workload they were taken on."""
    if world != 1 or args.workload_name != "config2" or args.kernel != "wavefront" or args.mode != "strict" or args.format != "rgba32f" or args.camera != "static":
        return None
    path = os.path.join(ROOT, PMC_SUMMARY)
    try:
        vals, digest = {}, None
        for line in open(path):
            f = line.split()
            if len(f) >= 3 and f[0] == "#" and f[1] == "kernel_source_digest":
                digest = f[2]
            elif len(f) >= 3 and f[1] == "mean":
                vals[f[0]] = float(f[2])
        if digest != kernel_source_digest():
            return {"source": PMC_SUMMARY, "stale": True}
        out = {"source": PMC_SUMMARY, "stale": False}
        if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
            out["traffic"] = (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0
        need = ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_THREAD_CYCLES_VALU", "SQ_INSTS_VALU")
        if all(k in vals for k in need):
            # FP64 wave-instructions x 64 lanes x the fraction of lanes active per VALU instruction.
            # SQ_THREAD_CYCLES_VALU counts active lanes x quad-cycles; an instruction of a full wave takes one quad-cycle
            # for 64 lanes, so lanes-per-instruction = THREAD_CYCLES / INSTS and the active fraction is that / 64.
            active = min(1.0, vals["SQ_THREAD_CYCLES_VALU"] / vals["SQ_INSTS_VALU"] / 64.0)
            slots = (vals["SQ_INSTS_VALU_ADD_F64"] + vals["SQ_INSTS_VALU_MUL_F64"] + 2.0 * vals["SQ_INSTS_VALU_FMA_F64"]) * 64.0
            out["fp64_lane_slots"] = slots
            out["active_lane_fraction"] = active
            out["flops"] = slots * active
        if "SQ_ACTIVE_INST_VALU" in vals and "SQ_BUSY_CYCLES" in vals and "GRBM_GUI_ACTIVE" in vals:
            # VALU busy: SQ_ACTIVE_INST_VALU is in quad-cycles summed over all waves; GRBM_GUI_ACTIVE is the sum of the 8 XCDs'
            # active cycles -> kernel cycles = GRBM / 8; SIMD-cycles available = kernel cycles x 256 CUs x 4 SIMDs
            kernel_cycles = vals["GRBM_GUI_ACTIVE"] / 8.0
            out["valu_busy"] = vals["SQ_ACTIVE_INST_VALU"] * 4.0 / (kernel_cycles * 1024.0)
            out["valu_busy_formula"] = "SQ_ACTIVE_INST_VALU x 4 / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)"
        return out
    except Exception:
        return None


def object_classes(arr):
    import numpy as np
    out = []
    for c in arr["coefs"]:
        if np.any(c[:10] != 0):
            out.append("cubic")
        elif np.any(c[13:16] != 0):
            out.append("cross")
        elif np.all(c[10:13] == 1.0):
            out.append("unitsq")
        elif np.any(c[10:13] != 0):
            out.append("square")
        else:
            out.append("linear")
    return out


# ----------------------------------------------------------------------------------------------------------------
# one wire format of the N-rank frame: what every rank renders into, what travels, how rank 0 rebuilds the frame
# ----------------------------------------------------------------------------------------------------------------
class FramePath:
    """Render -> (gather -> reassemble) for one framebuffer format.  Pipelined over two buffer sets on ONE stream per rank: the
    gather of frame k travels while frame k+1 renders, and rank 0 reassembles frame k behind its own render of frame k+1."""

    def __init__(self, env, fmt_name, gather_kind, cams):
        import numpy as np
        import torch
        import torch.distributed as dist
        self.np, self.torch, self.dist = np, torch, dist
        self.env, self.fmt_name, self.cams = env, fmt_name, cams
        pkg, args = env["pkg"], env["args"]
        self.pkg = pkg
        self.world, self.rank, self.dev, self.W, self.H = env["world"], env["rank"], env["dev"], env["W"], env["H"]
        self.dist_on, self.nccl = env["dist_on"], args.backend == "nccl"
        self.root = self.rank == 0
        self.fmt = pkg.RT_FMT_RGBA8 if fmt_name == "rgba8" else pkg.RT_FMT_RGBA32F
        self.px_dtype, self.px_bytes = (torch.uint8, 4) if fmt_name == "rgba8" else (torch.float32, 16)
        self.ren = pkg.Renderer(env["scene"], device=env["local_rank"], rank=self.rank, world=self.world, band_rows=args.band_rows, flags=env["flags"], fmt=self.fmt)
        self.mx = self.ren.max_local_rows
        if self.dist_on:
            self.stream = torch.cuda.current_stream(self.dev)
        else:   # N = 1: a stream of its own, so that the K timed frames can be captured into one hipGraph (run_timed)
            self.stream = torch.cuda.Stream(device=self.dev)
            torch.cuda.set_stream(self.stream)
        self.timed_region = "K launches"
        W, H, dev, world = self.W, self.H, self.dev, self.world
        nb = 2 if self.dist_on else 1
        self.sparse = self.dist_on and gather_kind == "sparse" and fmt_name == "rgba8"
        self.local = [torch.empty((self.mx, W, 4), dtype=self.px_dtype, device=dev) for _ in range(nb)]
        self.gathered = [torch.empty((world, self.mx, W, 4), dtype=self.px_dtype, device=dev) for _ in range(2)] if (self.root and self.dist_on) else None
        self.full = [torch.empty((H, W, 4), dtype=self.px_dtype, device=dev) for _ in range(2)] if (self.root and self.dist_on) else None
        self.works = [None, None]
        self.k = 0
        self.kernel_events = []       # (start, end) pairs around the render launches of the timed region
        self.record_kernel_events = False
        self.dense_resends = 0
        self.overflow_frames = []
        self.pending_hdr = []         # sparse: (frame k, buffer b, pinned header copy, event) not yet inspected
        self.cap = self.msg_bytes = 0
        if self.sparse:
            self._init_sparse()

    # sparse gather: capacity = tiles with content of the busiest rank (one untimed frame) + 25 % + 64, same on every rank
    def _init_sparse(self):
        np, torch, dist, pkg, ren = self.np, self.torch, self.dist, self.pkg, self.ren
        my_tiles = ((self.W + 15) // 16) * ((ren.local_rows + 15) // 16)
        probe = torch.zeros(pkg.Renderer.sparse_bytes(max(my_tiles, 1)), dtype=torch.uint8, device=self.dev)
        ren.update_sparse(probe.data_ptr(), max(my_tiles, 1), self.cams[0], stream=self.stream.cuda_stream, timed=False)
        torch.cuda.synchronize()
        need = torch.tensor([int(probe[:4].cpu().numpy().view(np.uint32)[0]), my_tiles], dtype=torch.int64, device=self.env["cdev"])
        dist.all_reduce(need, op=dist.ReduceOp.MAX)
        self.cap = int(min(int(need[1]), int(need[0]) + int(need[0]) // 4 + 64)) if not self.env["args"].sparse_capacity else self.env["args"].sparse_capacity
        self.msg_bytes = pkg.Renderer.sparse_bytes(self.cap)
        self.msg = [torch.zeros(self.msg_bytes, dtype=torch.uint8, device=self.dev) for _ in range(2)]
        self.gathered_msg = [torch.zeros((self.world, self.msg_bytes), dtype=torch.uint8, device=self.dev) for _ in range(2)] if self.root else None
        self.stamps = [torch.zeros(ren.sparse_stamp_bytes(), dtype=torch.uint8, device=self.dev) for _ in range(2)] if self.root else None
        self.asm_tag = [0, 0]   # per output buffer: 0 = first reassembly (paints everything), then 1, 2, ...
        self.hdr_host = [torch.zeros((self.world, 8), dtype=torch.uint8).pin_memory() for _ in range(2)] if self.root else None

    def cam_of(self, k):
        return self.cams[k % len(self.cams)]

    def _gather(self, send, recv):
        """The one collective of the path.  RCCL: async, the next frame's render is enqueued behind this call without waiting
        for it (it starts after the work already on this stream, so rank 0's reassembly of frame k-2 out of the same receive
        buffer is finished by then).  gloo (rehearsal only): staged through host memory."""
        dist, torch = self.dist, self.torch
        if self.nccl:
            return dist.gather(send, list(recv.unbind(0)) if self.root else None, dst=0, async_op=True)
        lc = send.cpu()
        gl = [torch.empty_like(lc) for _ in range(self.world)] if self.root else None
        w = dist.gather(lc, gl, dst=0, async_op=True)
        w.wait()
        if self.root:
            recv.copy_(torch.stack(gl))
        return w

    def _assemble(self, b):
        """root: turn what gather b delivered into the full frame b (same stream: no side stream, no events)."""
        if self.nccl:
            self.works[b].wait()   # stream-side: the render stream waits for the collective, the host does not
        if self.sparse:   # incremental: each output buffer keeps its frame, only tiles that lost their content are repainted
            self.ren.assemble_sparse_incremental(self.gathered_msg[b].data_ptr(), self.cap, self.full[b].data_ptr(), self.stamps[b].data_ptr(), self.asm_tag[b],
                                                 stream=self.stream.cuda_stream)
            self.asm_tag[b] = 1 if self.asm_tag[b] >= 0xFFFFFFF0 else self.asm_tag[b] + 1
        else:
            self.ren.assemble(self.gathered[b].data_ptr(), self.full[b].data_ptr(), stream=self.stream.cuda_stream)

    def _render(self, cam, b):
        torch = self.torch
        ev = None
        if self.record_kernel_events:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record(self.stream)
        if self.sparse:   # the kernel writes this rank's tiles with hits straight into the fixed-size message
            self.ren.update_sparse(self.msg[b].data_ptr(), self.cap, cam, stream=self.stream.cuda_stream, timed=False)
        else:
            self.ren.update(cam, dev_fb=self.local[b].data_ptr(), stream=self.stream.cuda_stream, timed=False)
        if ev:
            ev[1].record(self.stream)
            self.kernel_events.append(ev)

    def step(self):
        """One frame per rank."""
        k = self.k
        self.k = k + 1
        cam = self.cam_of(k)
        if not self.dist_on:
            self._render(cam, 0)
            return
        b = k & 1
        if self.works[b] is not None:
            self.works[b].wait()   # frame k-2 has left its send buffer (stream-side wait for RCCL, already complete for gloo)
        self._render(cam, b)
        if self.sparse:
            self.works[b] = self._gather(self.msg[b], self.gathered_msg[b] if self.root else None)
        else:
            self.works[b] = self._gather(self.local[b], self.gathered[b] if self.root else None)
        if self.root and k >= 1:
            self._finish_frame(k - 1)   # the previous frame: its gather had a whole render to arrive

    def _finish_frame(self, k):
        b = k & 1
        self._assemble(b)
        if self.sparse:
            # Did any rank's message overflow?  The headers are copied to pinned memory behind the reassembly and looked at
            # when the copy has landed (no host wait in the pipeline); an overflowed frame is sent again densely (below).
            torch = self.torch
            self.hdr_host[b].copy_(self.gathered_msg[b][:, :8].contiguous(), non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.stream)
            self.pending_hdr.append((k, b, ev))
            self._poll_overflow(block=False)

    def _poll_overflow(self, block):
        np = self.np
        keep = []
        for (k, b, ev) in self.pending_hdr:
            if block:
                ev.synchronize()
            if ev.query():
                if self.hdr_host[b].numpy().view(np.uint32)[:, 1].any():
                    self.overflow_frames.append(k)
            else:
                keep.append((k, b, ev))
        self.pending_hdr = keep

    def resend_dense(self, k):
        """A sparse frame whose message overflowed on some rank: every rank renders frame k again into a dense RGBA8 buffer and
        the rows are gathered and reassembled the general way (include/mi355rt.h: "send the dense frame instead").  Collective:
        all ranks must call it for the same k, which they learn from rank 0 (see run_timed)."""
        b = k & 1
        self.ren.update(self.cam_of(k), dev_fb=self.local[b].data_ptr(), stream=self.stream.cuda_stream, timed=False)
        w = self._gather(self.local[b], self.gathered[b] if self.root else None)
        if self.nccl:
            w.wait()   # stream-side
        if self.root:
            self.ren.assemble(self.gathered[b].data_ptr(), self.full[b].data_ptr(), stream=self.stream.cuda_stream)
            self.asm_tag[b] = 0   # the incremental state of this output buffer no longer matches it: repaint next time
        self.dense_resends += 1

    def flush(self):
        """root: reassemble the last frame (inside the timed region: K steps deliver K full frames); every rank: resend the
        frames whose sparse message overflowed."""
        if not self.dist_on:
            return
        if self.root and self.k >= 1:
            self._finish_frame(self.k - 1)
        if self.sparse:
            torch, dist = self.torch, self.dist
            if self.root:
                self._poll_overflow(block=True)
            n = torch.tensor([len(self.overflow_frames) if self.root else 0], dtype=torch.int64, device=self.env["cdev"])
            dist.broadcast(n, src=0)
            if int(n.item()):
                ks = torch.tensor(self.overflow_frames if self.root else [0] * int(n.item()), dtype=torch.int64, device=self.env["cdev"])
                dist.broadcast(ks, src=0)
                for k in ks.tolist():
                    self.resend_dense(int(k))
            self.overflow_frames = []

    def run_timed(self, steps, warmup):
        """W untimed steps, then exactly K timed steps bracketed by barrier + synchronize; returns (max-over-ranks seconds, mean
        kernel ms from HIP events on the render stream inside the timed region)."""
        torch, dist = self.torch, self.dist
        self.overflow_frames = []
        for _ in range(self.env["args"].settle if not self.dist_on else 0):   # (see --settle)
            self.step()
            self.stream.synchronize()
        for _ in range(warmup):
            self.step()
            if not self.dist_on:   # untimed: let the host see each warm-up frame's feedback words before it sizes the next launch, so that a
                self.stream.synchronize()   # handful of warm-up frames is enough for the launch order to settle (it lags by the queue depth otherwise)
        self.flush()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        self.record_kernel_events = self.dist_on   # N = 1: the timed region is K back-to-back launches, one event pair suffices
        self.kernel_events = []
        if self.dist_on:
            dist.barrier()
        # N = 1: the K frames are captured into ONE hipGraph before the clock starts and the timed region launches that graph: the
        # same K kernels with the same arguments, but the GPU no longer waits for the host between them.  (The host enqueues a frame in
        # 8 us and the GPU renders it in 45, so normally the queue is never empty -- but on a shared box the enqueuing thread is
        # descheduled for tens of ms now and then (tools/spike_probe.py: two 40 ms stalls in 4 500 frames), and one such stall inside a
        # 200-frame region would be reported as a 5x slower frame.)  --no-graph, or a failed capture, times K plain launches.
        graph = None
        if not self.dist_on and not self.env["args"].no_graph and len(self.cams) == 1:
            try:
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=self.stream):
                    for _ in range(steps):
                        self.step()
                self.timed_region = f"one hipGraph of the K = {steps} frame launches, captured before the clock starts"
            except Exception as exc:   # (capture not possible here: fall back, and say so)
                graph = None
                self.timed_region = f"K launches (graph capture failed: {type(exc).__name__})"
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ev0.record(self.stream)
        if graph is not None:
            graph.replay()
        else:
            for _ in range(steps):
                self.step()
        self.flush()
        ev1.record(self.stream)
        torch.cuda.synchronize()
        if self.dist_on:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        self.record_kernel_events = False
        tmax = torch.tensor([dt], dtype=torch.float64, device=self.env["cdev"])
        if self.dist_on:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        if self.kernel_events:
            kernel_ms = float(sum(a.elapsed_time(b) for a, b in self.kernel_events) / len(self.kernel_events))
        else:
            kernel_ms = ev0.elapsed_time(ev1) / steps
        return float(tmax.item()), kernel_ms

    def frame_check(self):
        """Outside the timed region: the reassembled N-rank frame must equal a single-context render of the same scene."""
        if not (self.dist_on and self.root) or self.W * self.H > 7680 * 4320:
            return None
        np = self.np
        ref = self.pkg.Renderer(self.env["scene"], device=self.env["local_rank"], flags=self.env["flags"], fmt=self.fmt)
        ref.update(self.cam_of(self.k - 1))
        ok = bool(np.array_equal(ref.download(), self.full[(self.k - 1) & 1].cpu().numpy()))
        ref.cleanup_update()
        return ok

    def gather_report(self):
        np = self.np
        if self.sparse:
            hdr = self.gathered_msg[(self.k - 1) & 1][:, :8].cpu().numpy().view(np.uint32)
            return {"kind": "sparse tiles", "capacity_tiles_per_rank": self.cap, "bytes_per_rank": self.msg_bytes, "dense_bytes_per_rank": int(self.mx * self.W * self.px_bytes),
                    "tiles_sent_per_rank": [int(v) for v in hdr[:, 0]], "overflow_in_last_frame": bool(hdr[:, 1].any()), "frames_resent_densely": self.dense_resends}
        return {"kind": "dense rows", "bytes_per_rank": int(self.mx * self.W * self.px_bytes)}

    def close(self):
        self.ren.cleanup_update()



# ----------------------------------------------------------------------------------------------------------------
# N = 1: every BASELINE config on this GPU (short regions), and what update() returns for one frame
# ----------------------------------------------------------------------------------------------------------------
def counters_of(pkg, scene, device, flags, cam=None):
    rc = pkg.Renderer(scene, device=device, flags=flags | pkg.RT_FLAG_COUNT)
    rc.update(cam)
    d = rc.counters_detail()
    rc.cleanup_update()
    return d


def measure_config(pkg, graft, name, device, flags, table, frames, check_rows):
    """One BASELINE config on one GPU: `frames` frames issued back to back on one stream (plain launches, one HIP event pair
    around them, after three synchronised warm-up frames), the median of synchronised single frames (`single_frame_ms`: what
    the reference's update() returns, src/update-cuda.cu:178-189), the roofline fraction from this config's own work counters,
    and -- part of the cpu_baseline leg, skipped with it -- a sample of rows compared with the oracle."""
    import numpy as np
    import torch
    scene_name, W, H, max_refl = CONFIGS[name]
    path = os.path.join(ROOT, "scenes", scene_name + ".yml")
    scene = pkg.Scene.load_from_file(path).set_size(W, H)
    if max_refl is not None:
        scene.set_max_reflections(max_refl)
    cnt = counters_of(pkg, scene, device, flags)
    arr = scene.arrays()
    flops, _, _ = algorithmic_flops(cnt, table, arr)
    dense = dense_reference_flops(cnt, table)
    capped = flops > dense
    flops = min(flops, dense)
    r = pkg.Renderer(scene, device=device, flags=flags)
    stream = torch.cuda.current_stream()
    singles = [r.update(stream=stream.cuda_stream) for _ in range(3 + max(5, min(20, frames)))][3:]   # (the first three: warm-up, launch order settles)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(frames):
        r.update(stream=stream.cuda_stream, timed=False)
    e1.record(stream)
    e1.synchronize()
    ms = e0.elapsed_time(e1) / frames
    rays = cnt["primary_rays"] + cnt["shadow_rays"] + cnt["reflect_rays"]
    out = {"config": name, "workload": f"{scene_name}.yml {W}x{H}" + (f", max_reflections {max_refl}" if max_refl is not None else "") + ", camera identity, RGBA32F",
           "frames": frames, "ms_per_step": ms, "single_frame_ms": float(np.median(singles)), "frames_per_s": 1e3 / ms, "value": rays / ms / 1e3, "unit": "Mrays/s",
           "rays_per_frame": int(rays), "tests_per_frame": int(cnt["tests"]),
           "roofline": {"bound": "valu", "achieved": flops / (ms * 1e-3) / 1e12, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": flops / (ms * 1e-3) / 1e12 / FP64_VECTOR_PEAK_TFLOPS, "algorithmic_flops_per_launch": flops,
                        "capped_at_dense_reference_count": capped, "dense_reference_flops_per_launch": dense,
                        "hbm_write_gbs": float(W) * H * 16.0 / (ms * 1e-3) / 1e9}}
    if check_rows:
        O = graft.load_oracle()
        img = r.download()
        rows = np.arange(0, H, max(1, H // 24), dtype=np.uint32)
        want = O.load_scene(path).with_size(W, H, max_refl).render(rows=rows, nthreads=min(16, os.cpu_count() or 1))
        got = img[rows][..., :3]
        diff = np.abs(got.astype(np.float64) - want)
        rel = diff / np.maximum(np.maximum(np.abs(got), np.abs(want)).astype(np.float64), 1e-300)   # (float64: 1e-300 is 0 in float32, and 0 / 0 a warning)
        out["oracle_row_sample"] = {"rows": int(rows.size), "identical": bool(np.array_equal(got, want)),
                                    "pixels_over_1e-5": int(((rel > 1e-5) & (diff > 1e-7)).any(axis=-1).sum()), "pixels": int(rows.size * W)}
    r.cleanup_update()
    return out


# ----------------------------------------------------------------------------------------------------------------
# N > 1: the 1-GPU denominator of the same frame, and the bandwidth of the links into rank 0
# ----------------------------------------------------------------------------------------------------------------
def one_gpu_frame_ms(pkg, scene, device, flags, cam, fmt, frames=5):
    """Rank 0 alone renders the WHOLE frame of the distributed workload (no sharding, no transport): frames issued back to back
    after three synchronised warm-up frames."""
    import torch
    r = pkg.Renderer(scene, device=device, flags=flags, fmt=fmt)
    stream = torch.cuda.current_stream()
    for _ in range(3):
        r.update(cam, stream=stream.cuda_stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(frames):
        r.update(cam, stream=stream.cuda_stream, timed=False)
    e1.record(stream)
    e1.synchronize()
    r.cleanup_update()
    torch.cuda.empty_cache()
    return e0.elapsed_time(e1) / frames


def link_bandwidth(dist, torch, rank, world, dev, cdev, nccl, mbytes=64):
    """GB/s into rank 0: one `mbytes` message per peer, one peer at a time (`per_peer_gbs`), then all peers at once through the
    very collective the frames use (`gather_ingest_gbs`, bytes arriving at rank 0 / time)."""
    n = mbytes << 20
    buf = torch.empty(n, dtype=torch.uint8, device=cdev)
    per_peer = []
    if world > 1:
        for peer in range(1, world):
            for it in range(2):   # the first transfer opens the connection
                torch.cuda.synchronize()
                dist.barrier()
                t0 = time.perf_counter()
                if rank == peer:
                    dist.send(buf, dst=0)
                elif rank == 0:
                    dist.recv(buf, src=peer)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
            per_peer.append(n / dt / 1e9)   # rank 0's clock is the one reported
    recv = [torch.empty(n, dtype=torch.uint8, device=cdev) for _ in range(world)] if rank == 0 else None
    for it in range(2):
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        dist.gather(buf, recv, dst=0)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    ingest = n * max(1, world - 1) / dt / 1e9
    del recv, buf
    torch.cuda.empty_cache()
    return {"message_mbytes": mbytes, "per_peer_gbs": per_peer, "gather_ingest_gbs": ingest,
            "what": "per_peer: dist.send / recv of one message, one peer at a time; gather_ingest: dist.gather of one message per rank, bytes from the other ranks "
                    "arriving at rank 0 / wall time (world 1: the self copy)"}

# ----------------------------------------------------------------------------------------------------------------
def run_rank(args, world):
    import numpy as np
    import torch
    import torch.distributed as dist
    import __graft_entry__ as graft

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the ray-tracing path has no CPU fallback")
    ndev = torch.cuda.device_count()
    if args.backend == "gloo":
        local_rank = local_rank % ndev
    elif local_rank >= ndev:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but this node shows {ndev} GPU(s) (LOCAL_RANK {local_rank}); refusing to share a GPU between RCCL ranks")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist_on = world > 1 or args.force_dist
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"bench.py: process group has {dist.get_world_size()} ranks, --gpus {args.gpus}")
    cdev = dev if args.backend == "nccl" else torch.device("cpu")  # where collectives' tensors live

    pkg = graft.load_package()
    args.workload_name, scene_name, W, H, max_refl = workload_for(world, args.workload, dist_on)
    scene = pkg.Scene.load_from_file(os.path.join(ROOT, "scenes", scene_name + ".yml")).set_size(W, H)
    if max_refl is not None:
        scene.set_max_reflections(max_refl)
    flags = pkg.RT_FLAG_FAST if args.mode == "fast" else pkg.RT_FLAG_STRICT
    flags |= {"wavefront": 0, "wavefront-nocull": pkg.RT_FLAG_NOCULL, "simple": pkg.RT_FLAG_SIMPLE}[args.kernel]
    if args.static_order:
        flags |= pkg.RT_FLAG_STATIC_ORDER
    if args.format == "auto":
        args.format = "rgba32f"
    n_orbit = 24
    cams = [orbit_pose(pkg, i, n_orbit) for i in range(n_orbit)] if args.camera == "orbit" else [pkg.IDENTITY]
    env = dict(pkg=pkg, args=args, world=world, rank=rank, local_rank=local_rank, dev=dev, cdev=cdev, W=W, H=H, scene=scene, flags=flags, dist_on=dist_on)

    # ---- N > 1: rank 0 renders the whole frame alone (the denominator of the speed-up), then the links into rank 0 are measured ----
    one_gpu = links = None
    if dist_on and not args.no_one_gpu:
        if rank == 0:
            one_gpu = {"rgba32f_ms": one_gpu_frame_ms(pkg, scene, local_rank, flags, cams[0], pkg.RT_FMT_RGBA32F),
                       "rgba8_ms": one_gpu_frame_ms(pkg, scene, local_rank, flags, cams[0], pkg.RT_FMT_RGBA8),
                       "what": f"rank 0 alone, the whole {W}x{H} frame in one context, 5 frames issued back to back after 3 synchronised warm-up frames"}
        dist.barrier()
        links = link_bandwidth(dist, torch, rank, world, dev, cdev, args.backend == "nccl")

    # ---- ray / work accounting: one counting frame per camera pose (not timed) ----
    keys = ["primary_rays", "shadow_rays", "reflect_rays", "tests", "hits", "solves", "tests_executed", "cull_evals"]
    rc = pkg.Renderer(scene, device=local_rank, rank=rank, world=world, band_rows=args.band_rows, flags=flags | pkg.RT_FLAG_COUNT)
    per_pose, detail0 = [], None
    for cam in cams:
        rc.update(cam)
        c = rc.counters()
        if detail0 is None:
            detail0 = rc.counters_detail() if args.kernel != "simple" else dict(c, executed_by_class=dict(unitsq=0, quadric=0, linear=0, cubic=0),
                                                                                solves_by_class=dict(unitsq=c["solves"], quadric=0, linear=0),
                                                                                cubic_branches=dict(cardano=0, trig=0, quad=0, linear=0))
        per_pose.append([c[k] for k in keys])
    rc.cleanup_update()
    tot = torch.tensor(per_pose, dtype=torch.int64, device=cdev)
    if dist_on:
        dist.all_reduce(tot)
    tot = tot.cpu().numpy()
    rays_per_pose = tot[:, 0] + tot[:, 1] + tot[:, 2]
    total0 = {k: int(v) for k, v in zip(keys, tot[0])}

    def rays_in(steps, warmup):  # rays of the timed frames (the orbit cycles through its poses, continuing after the warm-up)
        return int(sum(int(rays_per_pose[(warmup + i) % len(cams)]) for i in range(steps)))

    # ---- the timed path: the parity format, dense rows ----
    main = FramePath(env, args.format, "dense" if args.format == "rgba32f" else args.gather, cams)
    dt, kernel_ms = main.run_timed(args.steps, args.warmup)
    rays_timed = rays_in(args.steps, args.warmup)
    check_main = main.frame_check()
    gather_main = main.gather_report() if dist_on else None
    timed_frame_ok = None
    if not dist_on and len(cams) == 1:   # the last frame of the timed region (graph or not) against one more plain launch of the same frame
        last = main.local[0].clone()
        main.ren.update(cams[0], dev_fb=main.local[0].data_ptr(), stream=main.stream.cuda_stream, timed=False)
        main.stream.synchronize()
        timed_frame_ok = bool(torch.equal(last, main.local[0]))
        if not timed_frame_ok:
            raise SystemExit("bench.py: the last frame of the timed region differs from a plain launch of the same frame")

    # ---- N > 1: the display wire format (RGBA8, sparse tiles) timed the same way, reported next to the headline ----
    alt = None
    if dist_on and args.format == "rgba32f" and not args.no_alt:
        ap = FramePath(env, "rgba8", args.gather, cams)
        adt, akms = ap.run_timed(args.steps, args.warmup)
        acheck = ap.frame_check()
        if rank == 0:
            alt = {"framebuffer_format": "rgba8", "what": "iround(c*255) RGBA8, the wire format of the reference's CUDA back end (src/update-cuda.cu:149-156); "
                   "ranks send only their 16x16 tiles with content" if ap.sparse else "iround(c*255) RGBA8 rows",
                   "value": rays_timed / adt / 1e6, "unit": "Mrays/s", "ms_per_step": adt / args.steps * 1e3, "kernel_ms": akms,
                   "gather": ap.gather_report(), "gathered_frame_identical_to_single_gpu_frame": acheck}
        ap.close()

    # ---- N = 1, static camera: the same kernel over an orbit of 24 poses (frame time depends on the view) ----
    orbit = None
    if not dist_on and args.camera == "static" and args.workload_name == "config2" and not args.no_orbit:
        ts = []
        for i in range(n_orbit):
            cam = orbit_pose(pkg, i, n_orbit)
            for _ in range(6):   # the launch-order feedback settles on the new view: a few frames, each seen by the host before the next is issued
                main.ren.update(cam, dev_fb=main.local[0].data_ptr(), stream=main.stream.cuda_stream, timed=False)
                main.stream.synchronize()
            batches = []
            for _ in range(3):   # median of three batches: one host hiccup while enqueuing (tens of ms on a busy box) would otherwise own the pose
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(main.stream)
                for _ in range(10):
                    main.ren.update(cam, dev_fb=main.local[0].data_ptr(), stream=main.stream.cuda_stream, timed=False)
                e1.record(main.stream)
                e1.synchronize()
                batches.append(e0.elapsed_time(e1) / 10.0)
            ts.append(float(np.median(batches)))
        ts = np.array(ts) * 1e3
        orbit = {"poses": n_orbit, "what": "steady-state time per frame at each pose of an orbit around the scene, measured like the headline (6 settling frames, then the median of three batches of 10 frames issued back to back, "
                 "one HIP event pair on the render stream); `--camera orbit` times a camera that moves every frame instead",
                 "median_us": float(np.median(ts)), "max_us": float(ts.max()), "min_us": float(ts.min())}

    if rank == 0:
        arr = scene.arrays()
        cnt = dict(detail0)
        table, table_src = load_flop_table()
        n_div = n_sqrt = 0.0
        if args.kernel == "simple":  # the simple kernel evaluates every reference test, solving inline: dense count
            flops_launch = dense_reference_flops(cnt, table)
        else:
            flops_launch, n_div, n_sqrt = algorithmic_flops(cnt, table, arr)
        dense_local = dense_reference_flops(cnt, table)
        capped = flops_launch > dense_local
        if capped:   # the executed-algorithm count may never exceed the dense as-written figure
            flops_launch = dense_local
        achieved = flops_launch / (kernel_ms * 1e-3) / 1e12
        step_s = dt / args.steps
        dense_frame = float(np.mean([290.0 * t for t in tot[:, 3]]))   # whole job, per frame: 286 + a quadratic miss per reference test
        fb_bytes = float(main.ren.local_rows) * W * main.px_bytes
        pmc = pmc_profile(args, world)
        roof = {"bound": "valu", "achieved": achieved, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / FP64_VECTOR_PEAK_TFLOPS,
                "traffic": (pmc or {}).get("traffic"), "traffic_unit": "bytes/launch",
                "traffic_source": (f"{pmc['source']} (committed PMC passes of this command; not measured by this run)" if pmc and not pmc.get("stale") else
                                   ("PMC summary is older than the kernel sources: dropped" if pmc else None)),
                "kernel": "trace_tile_kernel" if args.kernel == "simple" else "wavefront_tile_kernel",
                "kernel_ms": kernel_ms, "kernel_ms_source": "HIP events on the render stream inside the timed region", "timed_region": main.timed_region, "timed_region_last_frame_identical_to_plain_launch": timed_frame_ok,
                "algorithmic_flops_per_launch": flops_launch, "algorithmic_flops_source": f"{table_src} (counting scalar over the kernel's own math) x device work counters by surface class",
                "capped_at_dense_reference_count": capped, "dense_reference_flops_per_launch": dense_local,
                "work_units_per_launch": {k: v for k, v in cnt.items()},
                "reference_dense_flops_per_frame": dense_frame, "reference_equivalent_tflops": dense_frame / step_s / 1e12,
                "time_vs_dense_algorithm_at_100pct_fp64_peak": (dense_frame / (FP64_VECTOR_PEAK_TFLOPS * 1e12 * world)) / step_s,
                "hbm_write_gbs": fb_bytes / (kernel_ms * 1e-3) / 1e9, "hbm_frac": fb_bytes / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
        rp = rocprof_kernel_avg_ms(args, world)
        if rp is not None and not rp["stale"]:
            roof["rocprof_kernel_avg_ms"] = rp["avg_ms"]
            roof["rocprof_kernel_avg_source"] = (f"{KERNEL_STATS} (rocprofv3 --kernel-trace --stats of this command, committed, {rp['launches']} launches of {rp['kernel']}; "
                                                 "begin-to-end of the kernel alone; taken with the kernel sources this run was built from)")
        elif rp is not None:
            roof["rocprof_kernel_avg_source"] = f"{KERNEL_STATS} is older than the kernel sources: dropped"
        if pmc and "flops" in pmc:
            roof["pmc_flops_per_launch"] = pmc["flops"]
            roof["pmc_flops_formula"] = "(SQ_INSTS_VALU_ADD_F64 + SQ_INSTS_VALU_MUL_F64 + 2 x SQ_INSTS_VALU_FMA_F64) x 64 x active-lane fraction (SQ_THREAD_CYCLES_VALU / SQ_INSTS_VALU / 64)"
            roof["pmc_active_lane_fraction"] = pmc["active_lane_fraction"]
            # the table counts a division / square root as ONE operation; the hardware counters see the instructions the compiler
            # expands them into: 12 (v_mul + 5 FMA-class + v_div_fmas) resp. 14 (2 v_mul + 6 FMA-class) instruction-flops in this build's ISA
            in_pmc_terms = flops_launch + (DIV_FLOPS_IN_ISA - 1.0) * n_div + (SQRT_FLOPS_IN_ISA - 1.0) * n_sqrt
            roof["table_in_pmc_terms"] = in_pmc_terms
            roof["table_in_pmc_terms_what"] = (f"table + {DIV_FLOPS_IN_ISA - 1:.0f} x {n_div:.4g} divisions + {SQRT_FLOPS_IN_ISA - 1:.0f} x {n_sqrt:.4g} square roots "
                                               "(their expansion into FP64 multiplies / FMAs, which the instruction counters count and the one-flop rule does not)")
            roof["table_vs_pmc"] = in_pmc_terms / pmc["flops"]
            roof["table_vs_pmc_within_10pct"] = bool(capped or 0.9 <= roof["table_vs_pmc"] <= 1.1)
            if not roof["table_vs_pmc_within_10pct"]:   # said loudly, in the line and on stderr; only a gross disagreement (an accounting bug) stops the run
                print(f"bench.py: WARNING: flop table ({in_pmc_terms:.4g} in instruction terms) and PMC-derived count ({pmc['flops']:.4g}) disagree by more than 10 %",
                      file=sys.stderr, flush=True)
                if not (0.75 <= roof["table_vs_pmc"] <= 1.25):
                    raise SystemExit("bench.py: flop table and PMC-derived count disagree by more than 25 %")
        if pmc and "valu_busy" in pmc:
            roof["valu_busy"], roof["valu_busy_formula"] = pmc["valu_busy"], pmc["valu_busy_formula"]
        cam_txt = "camera identity" if args.camera == "static" else f"camera moving along an orbit of {n_orbit} poses (a new pose every frame)"
        result = {
            "metric": {"config2": "Mrays/sec, 20spheres.yml @1920x1080", "config5": "Mrays/sec, 20spheres.yml @7680x4320 (BASELINE config 5), rows over the ranks",
                       "config2w": "Mrays/sec, 20spheres.yml @1920x1080 x N pixels (weak-scaled)"}.get(args.workload_name, f"Mrays/sec, {args.workload_name}"),
            "value": rays_timed / dt / 1e6, "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": step_s * 1e3, "frames_per_s": args.steps / dt,
            "higher_is_better": True, "scaling": "weak" if (args.workload_name == "config2w" or not dist_on) else "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "rccl_ranks": dist.get_world_size() if (dist_on and args.backend == "nccl") else 0,
            "config": {"workload": f"BASELINE {args.workload_name}: {scene_name}.yml {W}x{H}, {cam_txt}, {args.format.upper()} framebuffer" + (f", rows in bands of {args.band_rows} over {world} rank(s)" if dist_on else ""),
                       "objects": int(arr["coefs"].shape[0]),
                       "lights": int(arr["light_p"].shape[0]), "rays_per_frame": int(rays_per_pose[0]) if len(cams) == 1 else float(np.mean(rays_per_pose)),
                       "tests_per_frame": total0["tests"], "kernel_mode": args.mode, "kernel": args.kernel, "framebuffer_format": args.format,
                       "parallelism": f"rows band-cyclic x{world} (band {args.band_rows}), {args.backend} gather to rank 0 + device reassembly" if dist_on else "single GPU"},
            "roofline": roof,
        }
        if orbit:
            result["camera_orbit"] = orbit
        if check_main is not None:
            result["config"]["gathered_frame_identical_to_single_gpu_frame"] = check_main
        if gather_main:
            result["config"]["gather"] = gather_main
        if alt:
            result["config"]["alt"] = alt
        if os.environ.get("MI355RT_LIB"):
            result["config"]["MI355RT_LIB"] = os.environ["MI355RT_LIB"]   # (another build of the library was measured, not the in-tree product)
        if dist_on and one_gpu:
            # ---- the contract of BASELINE config 5: how much faster than ONE GPU rendering the same frame? ----
            result["one_gpu_same_frame"] = one_gpu
            result["speedup_vs_1gpu_" + args.workload_name] = {"rgba32f_dense": one_gpu["rgba32f_ms"] / (step_s * 1e3),
                                                                "rgba8_sparse" if (alt and alt["gather"]["kind"] == "sparse tiles") else "rgba8": (one_gpu["rgba8_ms"] / alt["ms_per_step"]) if alt else None,
                                                                "what": "time of one GPU rendering the whole frame / time per step of the N ranks incl. gather and reassembly; target of BASELINE.json: >= 3.5 at N = 8"}
            bw = links["gather_ingest_gbs"] if links else None
            into_root = float(gather_main["bytes_per_rank"]) * max(0, world - 1)
            result["links_into_rank0"] = links
            result["gather_bound_ms"] = {"rgba32f_dense": into_root / (bw * 1e9) * 1e3 if bw else None,
                                         "rgba8_alt": (float(alt["gather"]["bytes_per_rank"]) * max(0, world - 1) / (bw * 1e9) * 1e3) if (alt and bw) else None,
                                         "bytes_into_rank0_rgba32f_dense": into_root,
                                         "what": "bytes arriving at rank 0 per frame / the ingest bandwidth measured above (world 1: nothing arrives, 0): the step cannot be shorter"}
            # DESIGN.md section 7's expectation, written down before the 8-GPU run: the render shrinks with N, the gather does not
            pred = {}
            for n in (2, 4, 8):
                per_rank = float(W) * H * 16.0 / n
                for gbs in (50.0, 75.0):
                    g_ms = per_rank / (gbs * 1e9) * 1e3   # the N-1 senders use distinct links: the slowest one bounds the gather
                    pred[f"n{n}_link{int(gbs)}GBs"] = {"render_ms": one_gpu["rgba32f_ms"] / n, "gather_ms": g_ms, "step_ms": max(one_gpu["rgba32f_ms"] / n, g_ms),
                                                      "speedup": one_gpu["rgba32f_ms"] / max(one_gpu["rgba32f_ms"] / n, g_ms)}
            result["predicted_rgba32f_dense"] = {"table": pred, "what": "step = max(render / N, one rank's rows over one xGMI link at 50 / 75 GB/s), pipelined over two buffer sets (DESIGN.md section 7): "
                                                 "the RGBA32F frame is bound by the gather as soon as N > 1; only the RGBA8 sparse transport (config.alt) can follow the kernel"}
        if not dist_on and args.workload_name == "config2" and args.kernel == "wavefront" and args.camera == "static" and not args.no_configs:
            # ---- all five BASELINE configs on this GPU + what update() returns for one frame ----
            result["configs"] = []
            for name in ("config1", "config2", "config3", "config4", "config5"):
                result["configs"].append(measure_config(pkg, graft, name, local_rank, flags, table, 5 if name == "config5" else 20, not args.no_cpu_baseline))
            result["single_frame_ms"] = result["configs"][1]["single_frame_ms"]
            result["single_frame_ms_what"] = ("median of 20 synchronised single frames of the headline workload, one HIP event pair around each launch: what update() returns "
                                              "(src/update-cuda.cu:178-189); the headline's ms_per_step is per frame of frames issued back to back")
        if not args.no_cpu_baseline and world == 1:
            O = graft.load_oracle()
            osc = O.load_scene(os.path.join(ROOT, "scenes", scene_name + ".yml")).with_size(W, H, max_refl)
            cam0 = cams[0]
            t = time.perf_counter()
            for _ in range(args.cpu_frames):
                osc.render(cam0, counters=False, nthreads=1)
            cdt = time.perf_counter() - t
            rpf = int(rays_per_pose[0])
            result["cpu_baseline"] = {"value": rpf * args.cpu_frames / cdt / 1e6, "unit": "Mrays/s", "cores": 1, "kind": "port",
                                      "frames_per_s": args.cpu_frames / cdt,
                                      "sample": f"{args.cpu_frames} full frames of the same workload ({W}x{H}), 1 thread (the reference's CPU path is serial, "
                                                f"src/update-cpu.cpp:125-133), oracle built gcc -O2 -ffp-contract=off; {cdt:.1f} s"}
            ncpu = os.cpu_count() or 1
            nfr = max(2, min(4 * args.cpu_frames, ncpu))
            t = time.perf_counter()
            for _ in range(nfr):
                osc.render(cam0, counters=False, nthreads=ncpu)
            adt = time.perf_counter() - t
            result["cpu_baseline_all_cores"] = {"value": rpf * nfr / adt / 1e6, "unit": "Mrays/s", "cores": ncpu, "kind": "port", "frames_per_s": nfr / adt,
                                                "sample": f"{nfr} full frames, rows of the frame distributed over {ncpu} threads (os.cpu_count() of this box), same oracle build; {adt:.1f} s"}
            result["speedup_vs_cpu_1thread"] = result["value"] / result["cpu_baseline"]["value"]
            result["speedup_vs_cpu_all_cores"] = result["value"] / result["cpu_baseline_all_cores"]["value"]
        print(json.dumps(result), flush=True)
    main.close()
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="auto", help="auto (default): BASELINE config2 on one GPU, BASELINE config5 (the 8K frame, rows over the ranks) with N > 1 | "
                                                       "config1 .. config5 | config2w (1080p x N pixels, the weak-scaled workload of rounds 1-2)")
    ap.add_argument("--mode", default="strict", choices=["strict", "fast"])
    ap.add_argument("--kernel", default="wavefront", choices=["wavefront", "wavefront-nocull", "simple"],
                    help="A/B switch; the product default is the culling wavefront kernel")
    ap.add_argument("--camera", default="static", choices=["static", "orbit"],
                    help="static = identity (BASELINE: fixed-camera synthetic frames); orbit = a new pose every frame (24 poses around the scene)")
    ap.add_argument("--static-order", action="store_true", help="A/B: RT_FLAG_STATIC_ORDER (no launch-order feedback)")
    ap.add_argument("--band-rows", type=int, default=16)
    ap.add_argument("--format", default="auto", choices=["auto", "rgba32f", "rgba8"],
                    help="framebuffer / wire format of the HEADLINE measurement.  rgba32f (default at every N) = the CPU back end's un-quantised floats, "
                         "the parity format.  rgba8 = iround(c*255), what the reference's CUDA back end writes (src/update-cuda.cu:149-156); at N > 1 it "
                         "is always timed as well and reported as config.alt")
    ap.add_argument("--gather", default="sparse", choices=["sparse", "dense"],
                    help="what a rank sends to rank 0 in the RGBA8 measurement (N > 1).  dense = its rows; sparse = only its 16x16 tiles that are not "
                         "pure background, with their ids, in a fixed-size message (rt_render_sparse / rt_assemble_sparse)")
    ap.add_argument("--sparse-capacity", type=int, default=0, help="tiles per rank a sparse message holds (0 = busiest rank of one untimed frame + 25 %% + 64)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL (the product path). gloo stages the gather through host memory and lets several ranks share "
                         "one GPU: only for rehearsing the N>1 code path on a 1-GPU box")
    ap.add_argument("--force-dist", action="store_true", help="N = 1: run the N > 1 code path (process group, gather, reassembly) with one rank")
    ap.add_argument("--no-alt", action="store_true")
    ap.add_argument("--no-orbit", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="N = 1: skip the `configs` array (all five BASELINE configs, short regions)")
    ap.add_argument("--no-one-gpu", action="store_true", help="N > 1: skip rank 0's single-GPU render of the whole frame and the link measurement")
    ap.add_argument("--no-graph", action="store_true", help="N = 1: time K plain launches instead of one hipGraph of them")
    ap.add_argument("--settle", type=int, default=0, help="N = 1: untimed frames in front of the --warmup steps (experiments)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=2)
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            sys.exit(spawn_ranks(args.gpus))
        world = 1
    else:
        world = int(os.environ["WORLD_SIZE"])
    if world != args.gpus:   # never fall through to a run that measures something else than what was asked for
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    run_rank(args, world)


if __name__ == "__main__":
    main()
