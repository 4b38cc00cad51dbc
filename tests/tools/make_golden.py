#!/usr/bin/env python3
"""Writes tests/golden/frames_96x72.npz: float-RGB frames of all 8 scenes at 96x72 for two cameras, rendered by
the CPU ORACLE (oracle/rt_oracle.c).  These are regression fixtures of the oracle -- the reference ships no
rendered outputs and cannot be built here (see oracle/rt_oracle.h) -- kept so that a silent change of the
oracle or of the HIP path shows up against committed data."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

SCENES = ["quadratic", "20spheres", "reflection_test", "clebsch", "cayley", "cubic", "dingdong", "monkey_saddle"]
W, H = 96, 72
CAMERAS = {"identity": O.IDENTITY, "moved": O.camera_matrix(pos=(0.7, 0.9, -2.5), yaw_deg=84.0, pitch_deg=6.0)}

out = {"cam_identity": CAMERAS["identity"], "cam_moved": CAMERAS["moved"]}
for name in SCENES:
    s = O.load_scene(os.path.join(ROOT, "scenes", name + ".yml")).with_size(W, H)
    for cname, cam in CAMERAS.items():
        out[f"{name}__{cname}"] = s.render(cam=cam)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "frames_96x72.npz"), **out)
print("wrote", len(out) - 2, "frames")
