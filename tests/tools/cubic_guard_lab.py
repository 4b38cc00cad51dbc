#!/usr/bin/env python3
"""CPU experiment behind cubic_guarded (rt_math.hpp): how often does the guard hand a degree-3 test back to the dense path, and
does it ever answer differently from the oracle when it answers itself?  Runs the repository's degree-3 scenes and N random
scenes of tests/tools/fuzz_cubic.py's generator through tests/tools/cubic_guard_lab.cpp (which links the oracle: test
infrastructure).  No GPU.   usage: python tests/tools/cubic_guard_lab.py [n_fuzz_scenes] [verbose]"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as graft  # noqa: E402

O = graft.load_oracle()
O.build()
if not hasattr(O, "IDENTITY"):
    O.IDENTITY = np.eye(4).reshape(16)


class LabStats(C.Structure):
    _fields_ = [("tests", C.c_uint64 * 2), ("fallback", C.c_uint64 * 2), ("decision_diff", C.c_uint64 * 2), ("value_diff", C.c_uint64 * 2),
                ("worst_rel", C.c_double * 2), ("fb_reason", C.c_uint64 * 8), ("blocks", C.c_uint64 * 2), ("blocks_refusing", C.c_uint64 * 2)]


def build():
    out = os.path.join(ROOT, "tests", "tools", "bin", "libcubic_guard_lab.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.run(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-I" + os.path.join(ROOT, "tools", "flopcount_shim"),
                    "-I" + os.path.join(ROOT, "cuda-ray-tracer_amd", "csrc"), os.path.join(ROOT, "tests", "tools", "cubic_guard_lab.cpp"), "-o", out,
                    "-L" + os.path.join(ROOT, "oracle"), "-lrt_oracle", "-Wl,-rpath," + os.path.join(ROOT, "oracle")], check=True)
    lib = C.CDLL(out)
    lib.lab_run.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(LabStats), C.c_int]
    return lib


def run(lib, osc, cam, verbose):
    st = LabStats()
    sc = osc.c_scene()
    cam = np.ascontiguousarray(O.IDENTITY if cam is None else cam, dtype=np.float64).reshape(16)
    lib.lab_run(C.byref(sc), cam.ctypes.data_as(C.POINTER(C.c_double)), C.byref(st), verbose)
    return st


def show(name, st):
    for k, kind in enumerate(("primary", "shadow")):
        n = st.tests[k]
        if n:
            print(f"{name:28s} {kind:8s} tests {n:9d}  dense fallback {100.0 * st.fallback[k] / n:6.2f} %  decisions differ {st.decision_diff[k]:4d}  "
                  f"roots differ > 1e-7 {st.value_diff[k]:4d}  worst rel {st.worst_rel[k]:.2e}  8x8 blocks with a refusal {100.0 * st.blocks_refusing[k] / max(st.blocks[k], 1):5.1f} %", flush=True)
    if sum(st.fb_reason):
        print("    refused by: " + ", ".join(f"{w} {st.fb_reason[i]}" for i, w in enumerate(("sharpness", "|t3|~EPS", "discriminant", "trig root~EPS", "quad discriminant", "quad root~EPS", "|t2|,|t1|~EPS", "root~EPS/max_t")) if st.fb_reason[i]))


def fuzz_scene(seed):
    """tests/tools/fuzz_cubic.py's generator, on the oracle's scene class (no GPU library needed)."""
    rng = np.random.default_rng(88000 + seed)
    w, h = int(rng.integers(40, 200)), int(rng.integers(30, 150))
    s = O.Scene(w, h, float(rng.uniform(25, 80)), int(rng.integers(0, 4)), rng.uniform(0, 1, 3))
    for _ in range(int(rng.integers(1, 3))):
        q = np.zeros(20)
        q[:10] = rng.uniform(-1, 1, 10) * (rng.random(10) < rng.uniform(0.2, 1.0))
        q[10:16] = rng.uniform(-1, 1, 6) * (rng.random(6) < 0.8)
        q[16:19] = rng.uniform(-2, 2, 3)
        q[19] = rng.uniform(-4, 4)
        s.add_object(q, rng.uniform(0, 1, 3), 0.0)
    for _ in range(int(rng.integers(1, 5))):
        l = O.OrcLight()
        if rng.random() < 0.5:
            O.lib().orc_light_directional(C.c_float(float(rng.uniform(0.3, 1.5))), O._d3(rng.normal(size=3) + np.array([0, -1.0, 0])), O._f3(rng.uniform(0, 1, 3)), C.byref(l))
        else:
            O.lib().orc_light_spherical(C.c_float(float(rng.uniform(50, 500))), O._d3(rng.uniform([-8, 2, -12], [8, 12, 6])), O._f3(rng.uniform(0, 1, 3)), C.byref(l))
        s.lights.append(l)
    cam = O.camera_matrix(pos=(float(rng.uniform(-2, 2)), float(rng.uniform(-1, 2)), float(rng.uniform(-12, -6))), yaw_deg=float(rng.uniform(80, 100)),
                          pitch_deg=float(rng.uniform(-10, 10)))
    return s, cam


def main():
    n_fuzz = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    verbose = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    lib = build()
    tot = LabStats()
    for name, (w, h) in (("clebsch", (400, 300)), ("clebsch", (1920, 1080)), ("cayley", (400, 300)), ("cubic", (400, 300)), ("dingdong", (400, 300)), ("monkey_saddle", (400, 300))):
        osc = O.load_scene(os.path.join(ROOT, "scenes", name + ".yml")).with_size(w, h)
        st = run(lib, osc, None, verbose)
        show(f"{name} {w}x{h}", st)
    for seed in range(n_fuzz):
        osc, cam = fuzz_scene(seed)
        st = run(lib, osc, cam, verbose)
        for k in range(2):
            tot.tests[k] += st.tests[k]; tot.fallback[k] += st.fallback[k]; tot.decision_diff[k] += st.decision_diff[k]; tot.value_diff[k] += st.value_diff[k]
            tot.worst_rel[k] = max(tot.worst_rel[k], st.worst_rel[k])
        for i in range(8):
            tot.fb_reason[i] += st.fb_reason[i]
        for k in range(2):
            tot.blocks[k] += st.blocks[k]; tot.blocks_refusing[k] += st.blocks_refusing[k]
        if st.decision_diff[0] + st.decision_diff[1] + st.value_diff[0] + st.value_diff[1]:
            show(f"fuzz seed {seed}", st)
    show(f"{n_fuzz} fuzz scenes", tot)


if __name__ == "__main__":
    main()
