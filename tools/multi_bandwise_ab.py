#!/usr/bin/env python3
"""rt_render_multi on a one-GPU box (device list [0, 0, 0, 0] x 2 parts = 8 contexts, device copies instead of RCCL): the classic transport
(rows into rank-major slots, then rt_assemble over the whole frame) against RT_MULTI_BANDWISE (one strided copy per context straight into
the frame, no reassembly).  `ms` is rt_render_multi's own figure: device time on the root from the start of its render to the complete
frame.  What differs between the two is root-side memory traffic only -- the rendering is the same eight contexts.
usage: python tools/multi_bandwise_ab.py [width height]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

pkg = graft.load_package()
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (7680, 4320)
sc = pkg.Scene.load_from_file(os.path.join(ROOT, "scenes", "20spheres.yml")).set_size(W, H)
single = pkg.Renderer(sc, device=0)
for _ in range(4):
    single.update()
t_single = float(np.median([single.update() for _ in range(9)]))
want = single.download()
print(f"20spheres {W}x{H}: one context {t_single * 1e3:.0f} us")
for fmt, name in ((pkg.RT_FMT_RGBA32F, "rgba32f"), (pkg.RT_FMT_RGBA8, "rgba8")):
    res = {}
    for label, flags in (("classic", 0), ("bandwise", pkg.RT_MULTI_BANDWISE)):
        m = pkg.MultiRenderer(sc, [0, 0, 0, 0], band_rows=16, parts=2, fmt=fmt, flags=flags)
        for _ in range(4):
            m.update()
        res[label] = float(np.median([m.update() for _ in range(9)]))
        if fmt == pkg.RT_FMT_RGBA32F:
            assert np.array_equal(m.download(), want), label
        m.cleanup_update()
    frame_mb = W * H * (16 if fmt == pkg.RT_FMT_RGBA32F else 4) / 1e6
    print(f"  {name:8s} 8 contexts on one GPU: classic {res['classic'] * 1e3:7.0f} us   bandwise {res['bandwise'] * 1e3:7.0f} us   "
          f"saved {1e3 * (res['classic'] - res['bandwise']):6.0f} us  (frame {frame_mb:.0f} MB: the reassembly pass reads and writes it once)")
