#!/usr/bin/env python3
"""Static instruction census of the lean instantiation of the render kernel (rt_wavefront.hip compiled the way `make` compiles the strict
build, -save-temps): the whole kernel, and the part between the two s_memtime reads that bracket a block's light loop (shadow tests +
shading).  Scalar-unit audit of DESIGN.md: how many scalar instructions stand next to the vector ones, and of what kind.  Static counts
(every instruction once, whatever its trip count); the dynamic totals per frame are in profiles/r03_pmc_summary.txt.
usage: python tools/isa_census.py > profiles/r03_isa_census.txt"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "cuda-ray-tracer_amd", "csrc")
tmp = tempfile.mkdtemp(prefix="isa_census_")
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-fPIC", "-std=c++17", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-mllvm",
                "-amdgpu-kernarg-preload-count=12", "-DRT_VARIANT=strict", "-ffp-contract=off", "-save-temps", "-c", os.path.join(CSRC, "rt_wavefront.hip"), "-o", "wf.o"],
               cwd=tmp, check=True, capture_output=True)
asm = open(os.path.join(tmp, [f for f in os.listdir(tmp) if f.endswith("gfx950.s")][0])).read().splitlines()
start = next(i for i, l in enumerate(asm) if re.match(r"_ZN10rtw_strict21wavefront_tile_kernelILb0ELb0ELb0ELb0ELb1E.*:", l))
end = next(i for i in range(start, len(asm)) if asm[i].startswith(".Lfunc_end"))
body = asm[start:end]


def kind(op):
    if op.startswith("v_"):
        if "f64" in op:
            return "VALU f64"
        if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
            return "VALU lane <-> scalar (SGPR spills, uniform values)"
        if op.startswith("v_cmp"):
            return "VALU compare"
        if op.startswith(("v_cndmask", "v_mov")):
            return "VALU select / move"
        return "VALU other (f32, integer, conversions)"
    if op.startswith("s_load") or op.startswith("s_buffer"):
        return "scalar memory"
    if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_swappc")):
        return "branch"
    if op.startswith(("s_waitcnt", "s_nop", "s_sleep", "s_barrier", "s_endpgm")):
        return "wait / nop / barrier"
    if op.startswith("s_"):
        return "SALU"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "vector memory"
    return "other"


def census(lines, title):
    c = collections.Counter()
    for l in lines:
        m = re.match(r"\s+([a-z_0-9]+)\b", l)
        if m and not l.lstrip().startswith((".", ";")):
            c[kind(m.group(1))] += 1
    tot = sum(c.values())
    print(f"{title}: {tot} instructions")
    for k, v in sorted(c.items(), key=lambda kv: -kv[1]):
        print(f"  {k:52s} {v:6d}  {100.0 * v / tot:5.1f} %")
    return c


census(body, "lean instantiation, whole kernel (wavefront_tile_kernel<false, false, false, false, true>, strict)")
mt = [i for i, l in enumerate(body) if "s_memtime" in l]
if len(mt) >= 2:
    census(body[mt[0]:mt[-1]], "  ... of which between the s_memtime reads around a block's lights (point-light pass, ordered pass, their shadow tests and solves)")
