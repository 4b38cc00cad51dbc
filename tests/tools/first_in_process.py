#!/usr/bin/env python3
"""Render a scene as the very first GPU work of a process and compare it with the simple kernel's frame.

Regression check for the scratch-spill hazard described in tools/check_spills.py: a kernel that reloads a VGPR spilled
under a narrowed EXEC mask reads stale scratch, which happens to hold the right values when an earlier launch of the
same process left them there -- so the failure only shows in a fresh process (it was found as 160 unwritten tiles in
the first FMA-build frame of quadratic.yml).  usage: first_in_process.py <scene> <flags>; prints "OK" or the damage."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

pkg = graft.load_package()
name, flags = sys.argv[1], int(sys.argv[2])
sc = pkg.Scene.load_from_file(os.path.join(ROOT, "scenes", name + ".yml")).set_size(320, 240)


def render(fl, frames):
    r = pkg.Renderer(sc, device=0, flags=fl)
    out = []
    for _ in range(frames):
        r.update(None)
        out.append(r.download().copy())
    r.cleanup_update()
    return out


first = render(flags, 3)
ref = render((flags & pkg.RT_FLAG_FAST) | pkg.RT_FLAG_SIMPLE, 1)[0]
bad = [int(np.any(im != ref, axis=-1).sum()) for im in first]
unwritten = [int((im[..., 3] == 0).sum()) for im in first]
print("OK" if not any(bad) and not any(unwritten) else f"MISMATCH {name} flags={flags}: differing px per frame {bad}, unwritten px {unwritten}")
