#!/usr/bin/env python3
"""All five BASELINE configs on one GPU: frame time, frames/s, Mrays/s, parity summary vs the oracle on a row sample.
`us` is the time per frame of frames issued back to back on one stream, single launches (one HIP event pair around the batch; bench.py's
headline launches its timed frames as one hipGraph, which saves the ~3 us between two dependent launches: 42.2 instead of 43.6-45 us for config 2);
`alone` is one frame rendered into an idle GPU with a host synchronisation after it (event pair around that launch)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402
import torch  # noqa: E402

pkg, O = graft.load_package(), graft.load_oracle()
CFG = [(1, "quadratic", 640, 480, None), (2, "20spheres", 1920, 1080, None), (3, "reflection_test", 1920, 1080, 4),
       (4, "clebsch", 3840, 2160, None), (5, "20spheres", 7680, 4320, None)]
for cid, name, w, h, mr in CFG:
    path = os.path.join(ROOT, "scenes", name + ".yml")
    sc = pkg.Scene.load_from_file(path).set_size(w, h)
    if mr is not None:
        sc.set_max_reflections(mr)
    rc = pkg.Renderer(sc, device=0, flags=pkg.RT_FLAG_COUNT)
    rc.update()
    cnt = rc.counters()
    rc.cleanup_update()
    for mode, fl in (("strict", 0), ("fast", pkg.RT_FLAG_FAST)):
        r = pkg.Renderer(sc, device=0, flags=fl)
        for _ in range(3):
            r.update()
        t = np.array([r.update() for _ in range(20)])
        stream = torch.cuda.current_stream()
        reps = max(10, min(200, int(20000 / (np.median(t) * 1e3))))
        batches = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(reps):
                r.update(stream=stream.cuda_stream, timed=False)
            e1.record(stream)
            torch.cuda.synchronize()
            batches.append(e0.elapsed_time(e1) / reps)
        img = r.download()
        r.cleanup_update()
        rows = np.arange(0, h, max(1, h // 24), dtype=np.uint32)
        want = O.load_scene(path).with_size(w, h, mr).render(rows=rows, nthreads=8)
        got = img[rows][..., :3].astype(np.float64)
        diff = np.abs(got - want)
        rel = diff / np.maximum(np.maximum(np.abs(got), np.abs(want)), 1e-300)
        bad = int(((rel > 1e-5) & (diff > 1e-7)).any(axis=-1).sum())
        ms = float(np.median(batches))
        print(f"config {cid} {name:16s} {w}x{h} {mode:6s}: {ms*1e3:9.1f} us (alone {float(np.median(t))*1e3:7.1f})  {1e3/ms:9.1f} frames/s  {cnt['rays_total']/ms/1e3:9.1f} Mrays/s  "
              f"rays {cnt['rays_total']}  sample rows identical={np.array_equal(img[rows][..., :3], want)}  px>1e-5: {bad}/{rows.size*w}")
