#!/usr/bin/env python3
"""bench.py -- throughput of the per-pixel ray-tracing hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one frame: every pixel's primary ray, nearest hit over all surfaces, shadow rays to all lights,
Lambert shading and the mirror-bounce loop, written to an RGBA32F framebuffer resident in HBM.
Metric (BASELINE.json): Mrays/s on scenes/20spheres.yml; a ray = one primary, shadow or reflection ray
(SURVEY.md 8(d)); rays per frame are counted by the kernel itself (RT_FLAG_COUNT pass before timing).

N = 1: BASELINE config 2, 20spheres @ 1920x1080, camera = identity (the reference host's start-up pose).
N > 1: weak scaling -- the same scene at N x the pixels (same 16:9 aspect), rows band-cyclic over the ranks
(no data-path communication while rendering), one RCCL gather of the tiles to rank 0 per frame + a device
reassembly kernel on rank 0, as BASELINE.json's north_star prescribes.

Prints ONE JSON line (rank 0).  The oracle (oracle/) is used ONLY for the `cpu_baseline` leg.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

FP64_VECTOR_PEAK_TFLOPS = 78.6   # MI355X: 256 CU x 4 SIMD x 16 lanes/clk x 2 flop x 2.4 GHz = half the FP32 vector
                                 # peak (157.3 TF, MI355X_MICROARCH.md chip table); checked by profiles/*fp64_peak*
HBM_PEAK_GBS = 8000.0


def workload_for(n_gpus, name):
    if name == "config5":
        return "20spheres", 7680, 4320, None
    if name == "config3":
        return "reflection_test", 1920, 1080, 4
    if name == "config4":
        return "clebsch", 3840, 2160, None
    s = math.sqrt(n_gpus)
    w = int(round(1920 * s / 16.0)) * 16
    h = int(round(w * 9 / 16.0))
    return "20spheres", w, h, None


# FP64 operations per unit of work of the KERNEL'S OWN algorithm (strict build: every mul / add / sub / div / sqrt is
# one separately rounded operation = 1 flop; DESIGN.md "Flop accounting" derives each figure from rt_wavefront.hip /
# rt_math.hpp).  The reference's dense as-written count is 286 + solver per test; these are far
# below it because absent coefficient groups, shared per-ray monomials, the primary-ray t0 table and culling remove
# work -- which is why `achieved` must not be computed from the dense figure.
FLOPS = {
    "test_executed": {"unitsq": 16.0, "quadric": 50.0, "linear": 13.0, "cubic": 316.0},  # t2,t1,t0 + discriminant
    "solve": 20.0,          # recompute t1,t0 (13) + delta (3) + sqrt + (-t1 -+ sqrt) + 2*t2 + division
    "cull_eval": 40.0,      # one bounding-volume decision (relevant_mask / primary_cone_mask), one lane
    "primary_ray": 66.0,    # pixel -> direction (45, incl. normalise) + monomials (21)
    "shadow_ray": 14.0,     # FP32 round trip + mixed monomials; point lights also form d*d (same order)
    "reflect_ray": 40.0,    # reflect + bias + full monomials
    "hit": 100.0,           # point (6) + gradient normal (~80 on the 20-coefficient form) + shadow bias (6) + shading dots
}


def algorithmic_flops(cnt, classes):
    """FP64 operations one frame of the wavefront kernel executes, from the kernel's own work counters."""
    per_class = {"unitsq": "unitsq", "square": "quadric", "cross": "quadric", "linear": "linear", "cubic": "cubic"}
    mix = sum(FLOPS["test_executed"][per_class[c]] for c in classes) / max(1, len(classes))
    return (cnt["tests_executed"] * mix + cnt["solves"] * FLOPS["solve"] + cnt["cull_evals"] * FLOPS["cull_eval"] +
            cnt["primary_rays"] * FLOPS["primary_ray"] + cnt["shadow_rays"] * FLOPS["shadow_ray"] +
            cnt["reflect_rays"] * FLOPS["reflect_ray"] + cnt["hits"] * FLOPS["hit"])


def pmc_traffic_bytes(args, world):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (profiles/r01_final_pmc_summary.txt:
    separate rocprofv3 --pmc runs of this same command; FETCH_SIZE / WRITE_SIZE are in KiB, FETCH_SIZE doubled as the
    MI355X guide prescribes for gfx950).  Only valid for the default single-GPU workload the passes were taken on."""
    if world != 1 or args.workload != "config2" or args.kernel != "wavefront" or args.mode != "strict" or args.format != "rgba32f":
        return None
    path = os.path.join(ROOT, "profiles", "r01_final_pmc_summary.txt")
    try:
        vals = {}
        for line in open(path):
            f = line.split()
            if len(f) >= 3 and f[0] in ("FETCH_SIZE", "WRITE_SIZE"):
                vals[f[0]] = float(f[2])
        return (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0
    except Exception:
        return None


def object_classes(arr):
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="config2", help="config2 (default, weak-scaled with --gpus) | config3 | config4 | config5")
    ap.add_argument("--mode", default="strict", choices=["strict", "fast"])
    ap.add_argument("--kernel", default="wavefront", choices=["wavefront", "wavefront-nocull", "simple"],
                    help="A/B switch; the product default is the culling wavefront kernel")
    ap.add_argument("--band-rows", type=int, default=16)
    ap.add_argument("--format", default="auto", choices=["auto", "rgba32f", "rgba8"],
                    help="framebuffer / wire format.  rgba32f = the CPU back end's un-quantised floats (the parity format; "
                         "default on one GPU).  rgba8 = iround(c*255) RGBA8, the format the reference's CUDA back end writes "
                         "(src/update-cuda.cu:149-156); default when the frame is gathered (N > 1): the gather to rank 0 is "
                         "xGMI-bound and a float frame is 4x the bytes of the frame a display needs")
    ap.add_argument("--gather", default="sparse", choices=["sparse", "dense"],
                    help="what a rank sends to rank 0 (N > 1, rgba8).  dense = its rows; sparse = only its 16x16 tiles that are not "
                         "pure background, with their ids, in a fixed-size message (rt_render_sparse / rt_assemble_sparse): the gather "
                         "is xGMI-bound and 83 %% of this workload's tiles are background")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL (the product path). gloo stages the gather through host memory and lets several ranks share "
                         "one GPU: only for rehearsing the N>1 code path on a 1-GPU box")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=2)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the ray-tracing path has no CPU fallback")
    if args.backend == "gloo":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    cdev = dev if args.backend == "nccl" else torch.device("cpu")  # where collectives' tensors live

    pkg = graft.load_package()
    scene_name, W, H, max_refl = workload_for(world, args.workload)
    scene = pkg.Scene.load_from_file(os.path.join(ROOT, "scenes", scene_name + ".yml")).set_size(W, H)
    if max_refl is not None:
        scene.set_max_reflections(max_refl)
    flags = pkg.RT_FLAG_FAST if args.mode == "fast" else pkg.RT_FLAG_STRICT
    flags |= {"wavefront": 0, "wavefront-nocull": pkg.RT_FLAG_NOCULL, "simple": pkg.RT_FLAG_SIMPLE}[args.kernel]
    band = args.band_rows
    cam = pkg.IDENTITY
    if args.format == "auto":
        args.format = "rgba32f" if world == 1 else "rgba8"
    fmt = pkg.RT_FMT_RGBA8 if args.format == "rgba8" else pkg.RT_FMT_RGBA32F
    px_dtype, px_bytes = (torch.uint8, 4.0) if args.format == "rgba8" else (torch.float32, 16.0)

    # ---- ray accounting: one counting frame (not timed) ----
    rc = pkg.Renderer(scene, device=local_rank, rank=rank, world=world, band_rows=band, flags=flags | pkg.RT_FLAG_COUNT, fmt=fmt)
    rc.update(cam)
    cnt = rc.counters()
    rc.cleanup_update()
    keys = ["primary_rays", "shadow_rays", "reflect_rays", "tests", "hits", "solves", "tests_executed", "cull_evals"]
    tot = torch.tensor([cnt[k] for k in keys], dtype=torch.int64, device=cdev)
    if world > 1:
        dist.all_reduce(tot)
    total = {k: int(v) for k, v in zip(keys, tot.tolist())}
    rays_per_frame = total["primary_rays"] + total["shadow_rays"] + total["reflect_rays"]

    # ---- the timed path ----
    ren = pkg.Renderer(scene, device=local_rank, rank=rank, world=world, band_rows=band, flags=flags, fmt=fmt)
    mx = ren.max_local_rows
    stream = torch.cuda.current_stream(dev)
    root = rank == 0
    # double-buffered so that the gather / reassembly of frame k overlaps the rendering of frame k+1 (N > 1)
    local = [torch.empty((mx, W, 4), dtype=px_dtype, device=dev) for _ in range(2 if world > 1 else 1)]
    gathered = [torch.empty((world, mx, W, 4), dtype=px_dtype, device=dev) for _ in range(2)] if (root and world > 1) else None
    full = [torch.empty((H, W, 4), dtype=px_dtype, device=dev) for _ in range(2)] if (root and world > 1) else None
    works = [None, None]
    state = {"k": 0}

    # sparse gather: capacity = tiles with content of the busiest rank (one untimed frame) + 25 % + 64, same on every rank
    sparse = world > 1 and args.gather == "sparse" and args.format == "rgba8"
    cap = msg_bytes = 0
    msg = gathered_msg = stamps = asm_tag = None
    if sparse:
        my_tiles = ((W + 15) // 16) * ((ren.local_rows + 15) // 16)
        probe = torch.zeros(pkg.Renderer.sparse_bytes(max(my_tiles, 1)), dtype=torch.uint8, device=dev)
        ren.update_sparse(probe.data_ptr(), max(my_tiles, 1), cam, stream=stream.cuda_stream, timed=False)
        torch.cuda.synchronize()
        need = torch.tensor([int(probe[:4].cpu().numpy().view(np.uint32)[0]), my_tiles], dtype=torch.int64, device=cdev)
        dist.all_reduce(need, op=dist.ReduceOp.MAX)
        cap = int(min(int(need[1]), int(need[0]) + int(need[0]) // 4 + 64))
        msg_bytes = pkg.Renderer.sparse_bytes(cap)
        msg = [torch.zeros(msg_bytes, dtype=torch.uint8, device=dev) for _ in range(2)]
        gathered_msg = [torch.zeros((world, msg_bytes), dtype=torch.uint8, device=dev) for _ in range(2)] if root else None
        stamps = [torch.zeros(ren.sparse_stamp_bytes(), dtype=torch.uint8, device=dev) for _ in range(2)] if root else None
        asm_tag = [0, 0]  # per output buffer: 0 = first reassembly (paints everything), then 1, 2, ...
        del probe

    def assemble(b):
        """root: turn what gather b delivered into the full frame b (same stream: no side stream, no events -- the host
        cost of this Python loop matters at 60 us per frame)."""
        if args.backend == "nccl":
            works[b].wait()   # stream-side: the render stream waits for the collective, the host does not
        recv = gathered_msg[b] if sparse else gathered[b]
        if sparse:  # incremental: each output buffer keeps its frame, only tiles that lost their content are repainted
            ren.assemble_sparse_incremental(recv.data_ptr(), cap, full[b].data_ptr(), stamps[b].data_ptr(), asm_tag[b], stream=stream.cuda_stream)
            asm_tag[b] = 1 if asm_tag[b] >= 0xFFFFFFF0 else asm_tag[b] + 1
        else:
            ren.assemble(recv.data_ptr(), full[b].data_ptr(), stream=stream.cuda_stream)

    def step():
        """One frame per rank.  Pipelined over two buffer sets: the gather of frame k travels while frame k+1 renders,
        and rank 0 reassembles frame k-1 behind its own render of frame k (everything on one stream per rank)."""
        k = state["k"]
        state["k"] = k + 1
        if world == 1:
            ren.update(cam, dev_fb=local[0].data_ptr(), stream=stream.cuda_stream, timed=False)
            return
        b = k & 1
        if works[b] is not None:
            works[b].wait()   # frame k-2 has left its send buffer (stream-side wait for NCCL, already complete for gloo)
        if sparse:  # the kernel writes this rank's tiles with hits straight into the fixed-size message
            ren.update_sparse(msg[b].data_ptr(), cap, cam, stream=stream.cuda_stream, timed=False)
        else:
            ren.update(cam, dev_fb=local[b].data_ptr(), stream=stream.cuda_stream, timed=False)
        send = msg[b] if sparse else local[b]
        recv = (gathered_msg[b] if sparse else gathered[b]) if root else None
        if args.backend == "nccl":
            # RCCL gather over xGMI, the only collective of the path; async: the next frame's render is enqueued behind
            # this call without waiting for it.  (It starts after the work already on this stream, so rank 0's reassembly
            # of frame k-2 out of the same receive buffer is finished by then.)
            works[b] = dist.gather(send, list(recv.unbind(0)) if root else None, dst=0, async_op=True)
        else:  # rehearsal only: host-staged gather through gloo
            lc = send.cpu()
            gl = [torch.empty_like(lc) for _ in range(world)] if root else None
            works[b] = dist.gather(lc, gl, dst=0, async_op=True)
            works[b].wait()
            if root:
                recv.copy_(torch.stack(gl))
        if root and k >= 1:
            assemble((k - 1) & 1)   # the previous frame: its gather had a whole render to arrive

    def flush():
        """root: reassemble the last frame (inside the timed region: K steps deliver K full frames)."""
        if world > 1 and root and state["k"] >= 1:
            assemble((state["k"] - 1) & 1)

    for _ in range(args.warmup):
        step()
    flush()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record(stream)
    for _ in range(args.steps):
        step()
    flush()
    ev1.record(stream)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=cdev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    ev_ms = ev0.elapsed_time(ev1)

    # per-launch duration of the dominant kernel from HIP events on its own stream (N=1: the timed region is K
    # back-to-back launches of it; N>1: measured in a short extra pass, the timed region also holds the gather)
    if world == 1:
        kernel_ms = ev_ms / args.steps
    else:
        kernel_ms = float(np.mean([ren.update(cam, dev_fb=local[0].data_ptr(), stream=stream.cuda_stream, timed=True) for _ in range(10)]))

    # outside the timed region: the reassembled N-rank frame must equal a single-context render of the same scene
    frame_check = None
    if world > 1 and rank == 0 and W * H <= 7680 * 4320:
        ref = pkg.Renderer(scene, device=local_rank, flags=flags, fmt=fmt)
        ref.update(cam)
        last = (state["k"] - 1) & 1
        frame_check = bool(np.array_equal(ref.download(), full[last].cpu().numpy()))
        ref.cleanup_update()

    result = None
    if rank == 0:
        arr = scene.arrays()
        local_cnt = {k: cnt[k] for k in keys}
        if args.kernel == "simple":  # the simple kernel evaluates every reference test, solving inline
            local_cnt["tests_executed"], local_cnt["solves"] = local_cnt["tests"], local_cnt["tests"] // 16
        flops_launch = algorithmic_flops(local_cnt, object_classes(arr))
        achieved = flops_launch / (kernel_ms * 1e-3) / 1e12
        dense_flops = total["tests"] * 290.0  # the reference's as-written count: 286 expansion + ~4 solver per test
        dense_equiv = dense_flops / (dt / args.steps) / 1e12
        fb_bytes = float(ren.local_rows) * W * px_bytes
        result = {
            "metric": "Mrays/sec, 20spheres.yml @1920x1080 (weak-scaled with --gpus)" if args.workload == "config2" else f"Mrays/sec, {args.workload}",
            "value": rays_per_frame * args.steps / dt / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "frames_per_s": args.steps / dt,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{scene_name}.yml {W}x{H}, camera identity, {args.format.upper()} framebuffer", "objects": int(arr["coefs"].shape[0]),
                       "lights": int(arr["light_p"].shape[0]), "rays_per_frame": rays_per_frame, "tests_per_frame": total["tests"],
                       "kernel_mode": args.mode, "kernel": args.kernel, "framebuffer_format": args.format,
                       "parallelism": f"rows band-cyclic x{world} (band {band}), gather to rank 0" if world > 1 else "single GPU"},
            # bound: the FP64 vector (VALU) pipe -- no dense contraction exists in this path, so no MFMA; HBM traffic is
            # the 16 B/pixel framebuffer write only.  `achieved` counts the operations the kernel's own algorithm executes.
            "roofline": {"bound": "valu", "achieved": achieved, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP64_VECTOR_PEAK_TFLOPS, "traffic": pmc_traffic_bytes(args, world), "traffic_unit": "bytes/launch (PMC)",
                         "kernel": "trace_tile_kernel" if args.kernel == "simple" else "wavefront_tile_kernel",
                         "kernel_ms": kernel_ms, "algorithmic_flops_per_launch": flops_launch,
                         "work_units_per_launch": {k: local_cnt[k] for k in keys},
                         "reference_dense_flops_per_frame": dense_flops, "reference_equivalent_tflops": dense_equiv,
                         "time_vs_dense_algorithm_at_100pct_fp64_peak": (dense_flops / (FP64_VECTOR_PEAK_TFLOPS * 1e12)) / (dt / args.steps),
                         "hbm_write_gbs": fb_bytes / (kernel_ms * 1e-3) / 1e9, "hbm_frac": fb_bytes / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
        }
        if frame_check is not None:
            result["config"]["gathered_frame_identical_to_single_gpu_frame"] = frame_check
        if world > 1 and sparse:
            hdr = gathered_msg[(state["k"] - 1) & 1][:, :8].cpu().numpy().view(np.uint32)
            result["config"]["gather"] = {"kind": "sparse tiles", "capacity_tiles_per_rank": cap, "bytes_per_rank": msg_bytes,
                                          "dense_bytes_per_rank": int(mx * W * px_bytes), "tiles_sent_per_rank": [int(v) for v in hdr[:, 0]],
                                          "overflow": bool(hdr[:, 1].any())}
        elif world > 1:
            result["config"]["gather"] = {"kind": "dense rows", "bytes_per_rank": int(mx * W * px_bytes)}
        if not args.no_cpu_baseline and world == 1:
            O = graft.load_oracle()
            osc = O.load_scene(os.path.join(ROOT, "scenes", scene_name + ".yml")).with_size(W, H, max_refl)
            t = time.perf_counter()
            for _ in range(args.cpu_frames):
                _, ocnt = osc.render(cam, counters=False, nthreads=1), None
            cdt = time.perf_counter() - t
            result["cpu_baseline"] = {"value": rays_per_frame * args.cpu_frames / cdt / 1e6, "unit": "Mrays/s", "cores": 1, "kind": "port",
                                      "frames_per_s": args.cpu_frames / cdt,
                                      "sample": f"{args.cpu_frames} full frames of the same workload ({W}x{H}), 1 thread (the reference's CPU path is serial), "
                                                f"oracle built gcc -O2 -ffp-contract=off; {cdt:.1f} s"}
            result["speedup_vs_cpu_1thread"] = result["value"] / result["cpu_baseline"]["value"]
        print(json.dumps(result), flush=True)
    ren.cleanup_update()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
