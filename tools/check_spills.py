#!/usr/bin/env python3
"""Build-time guard: no instantiation of the render kernels may spill VGPRs to scratch -- or have a private segment at all.

Why: with ROCm 7.2's compiler a VGPR spill can land at the top of a join block BEFORE the `s_or_b64 exec` that
re-enables the lanes of a divergent region (seen in the FMA build of the quadric instantiation: the store ran with one
active lane, the reload with all of them), so the other lanes read back whatever the scratch slot held -- correct-looking
when an earlier launch of the same process left the same values there, garbage (unwritten tiles, wild stores) when the
kernel was the first one of the process.  Occupancy targets in rt_wavefront.hip are therefore chosen so that nothing
spills, and `make` runs this check on the remarks its compile rules save.

A private segment without any spill (the register allocator reserves a spill slot and an emergency slot, then folds the spill
away: "ScratchSize 20" with no scratch instruction in the code) is refused too: a kernel with one costs ~0.7 us more per launch
(same-box A/B at 1080p: all-empty frame 15.7 -> 16.3 us, start pose 46.1 -> 46.8 us, the same 0.6-0.7 us at every pose).
Small, unrelated source changes make it come and go (an opaque asm copy of a value was enough).
usage: check_spills.py [extra hipcc flags...]   |   check_spills.py --logs build/csrc/*.remarks"""
import re
import subprocess
import sys
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "cuda-ray-tracer_amd", "csrc")


def parse(label, text):
    """-Rpass-analysis=kernel-resource-usage remarks -> one line per kernel; returns the number of kernels that spill VGPRs or have a private segment."""
    bad, name, row = 0, None, {}
    for line in text.splitlines():
        m = re.search(r"remark: +Function Name: (\S+)", line)
        if m:
            name, row = m.group(1), {}
            continue
        m = re.search(r"remark: +([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
        if m and name:
            row[m.group(1).strip()] = int(m.group(2))
            if m.group(1).strip().startswith("LDS Size"):
                t = re.search(r"ILb(\d)ELb(\d)ELb(\d)ELb(\d)ELb(\d)E", name)
                tag = "<count=%s gq=%s cubic=%s mirror=%s lean=%s>" % t.groups() if t else name[:40]
                spills = row.get("VGPRs Spill", 0)
                print(f"{label:28s} {tag:40s} VGPRs {row.get('VGPRs', 0):3d}  occupancy {row.get('Occupancy', 0)}  SGPR spills {row.get('SGPRs Spill', 0):3d}"
                      f"  VGPR spills {spills:3d}  scratch {row.get('ScratchSize', 0)}" + ("   <-- VGPR SPILL" if spills else ("   <-- PRIVATE SEGMENT" if row.get('ScratchSize', 0) else "")))
                bad += (spills != 0) or (row.get('ScratchSize', 0) != 0)
                name = None
    return bad


bad = 0
if len(sys.argv) > 1 and sys.argv[1] == "--logs":   # remarks written by the Makefile's compile rules
    for path in sys.argv[2:]:
        bad += parse(os.path.basename(path).replace(".remarks", ""), open(path).read())
else:
    for src in ("rt_wavefront.hip", "rt_kernels.hip"):
        for variant, contract in (("strict", "off"), ("fast", "fast")):
            cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", f"-ffp-contract={contract}", f"-DRT_VARIANT={variant}",
                   f"-DRT_FAST={1 if variant == 'fast' else 0}", "-I" + CSRC, "-I" + os.path.join(ROOT, "include"), "-c", os.path.join(CSRC, src),
                   "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"] + (["-mllvm", "-amdgpu-kernarg-preload-count=12"] if src == "rt_wavefront.hip" else []) + sys.argv[1:]   # (the Makefile's flags)
            bad += parse(f"{src} {variant}", subprocess.run(cmd, capture_output=True, text=True).stderr)
if bad:
    print(f"{bad} kernel(s) spill VGPRs or have a private segment: lower their occupancy target (wf_occupancy in rt_wavefront.hip) / perturb the source")
    sys.exit(1)
print("no VGPR spills, no private segment")
