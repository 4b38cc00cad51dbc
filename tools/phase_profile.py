#!/usr/bin/env python3
"""Per-phase wave-cycle shares of the wavefront kernel.  Needs a diagnostic library: make -C cuda-ray-tracer_amd clean &&
make -C cuda-ray-tracer_amd STAMPS=1, and MI355RT_DEBUG_COUNTERS=1 in the environment.  Shares only -- a stamped build
is slower than the product build (MI355X guide, "In-kernel stamps")."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["MI355RT_DEBUG_COUNTERS"] = "1"
os.environ["MI355RT_ALLOW_DIAGNOSTIC"] = "1"
import __graft_entry__ as graft  # noqa: E402

pkg = graft.load_package()
name = sys.argv[1] if len(sys.argv) > 1 else "20spheres"
W = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
H = int(sys.argv[3]) if len(sys.argv) > 3 else 1080
flags = int(sys.argv[4]) if len(sys.argv) > 4 else 0
sc = pkg.Scene.load_from_file(os.path.join(ROOT, "scenes", name + ".yml")).set_size(W, H)
r = pkg.Renderer(sc, device=0, flags=flags)
for _ in range(3):
    ms = r.update()
d = r.debug_counters()[8:20]
names = ["stage scene->LDS", "setup/primary dir", "A nearest", "A normal+compact", "barrier after A", "A' balls + barrier",
         "B shadow items", "barrier after B", "C shade", "D blend + barrier", "store", "-"]
if os.environ.get("TIMELINE_LEAN_NAMES") or os.environ.get("MI355RT_LEAN") == "always":   # the lean instantiation stamps its set-up in pieces (slots 3, 4, 7, 8)
    names[3], names[4], names[7], names[8] = "setup: args + launch-order decode", "setup: decode barrier", "setup: tile -> staging written", "setup: staging barrier"
    names[1] = "setup: camera tables + primary dir"
    names[11] = "classify / paint / index slots"
tot = sum(d)
print(f"{name} {W}x{H} flags={flags}: kernel {ms*1e3:.1f} us (stamped build); total wave-cycles {tot:.3e}")
for n, v in zip(names, d):
    print(f"  {n:36s} {v:14d}  {100.0*v/max(tot,1):5.1f} %")
