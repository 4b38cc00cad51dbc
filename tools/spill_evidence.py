#!/usr/bin/env python3
"""Evidence for the "no kernel may spill VGPRs" rule: recompiles the wavefront kernel as it was BEFORE that
rule (git f33cdf6^, the FMA build that lost 160 tiles in the first frame of a fresh process, gpurun_out/r1/crash*.log) and the
current one, and prints, for each: the register / spill table of the compiler's resource remarks, every scratch (spill)
instruction with the nearest instruction that touches EXEC before and after it, and -- for the current build -- the count of
scratch instructions (none) and the SGPR-spill instructions (v_writelane / v_readlane, which do not depend on EXEC).
usage: python tools/spill_evidence.py > profiles/r02_spill_evidence.txt        (CPU only: hipcc cross-compiles)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "cuda-ray-tracer_amd", "csrc")
PRE_FIX = "f33cdf6^"


def compile_fast(src_dir, extra):
    out = os.path.join(src_dir, "wf_fast.s")
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-I" + src_dir, "-I" + os.path.join(ROOT, "include"), "-DRT_VARIANT=fast", "-DRT_FAST=1",
           "-ffp-contract=fast", "--cuda-device-only", "-S", os.path.join(src_dir, "rt_wavefront.hip"), "-o", out, "-Rpass-analysis=kernel-resource-usage"] + extra
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.exit(r.stderr[-3000:])
    return open(out).read(), r.stderr


def table(remarks):
    rows, name, row = [], None, {}
    for line in remarks.splitlines():
        m = re.search(r"remark: +Function Name: (\S+)", line)
        if m:
            name, row = m.group(1), {}
            continue
        m = re.search(r"remark: +([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
        if m and name:
            row[m.group(1).strip()] = int(m.group(2))
            if m.group(1).strip().startswith("LDS Size") and "wavefront_tile_kernel" in name:
                t = re.search(r"ILb(\d)ELb(\d)ELb(\d)ELb(\d)E", name)
                rows.append((t.groups(), row))
                name = None
    return rows


def functions(asm):
    for f in re.split(r"\n(?=_ZN\w*wavefront_tile_kernel)", asm)[1:]:
        name = f.split(":")[0]
        t = re.search(r"ILb(\d)ELb(\d)ELb(\d)ELb(\d)E", name)
        if t and ".Lfunc_end" in f:
            yield t.groups(), f[:f.index(".Lfunc_end")].split("\n")


def report(title, asm, remarks, spill_lines=True):
    print("=" * 120)
    print(title)
    print("=" * 120)
    print("instantiation <count gq cubic mirror>   VGPRs  occupancy  SGPR spills  VGPR spills  scratch bytes")
    for tag, row in table(remarks):
        print("  <%s %s %s %s>   %5d  %9d  %11d  %11d  %13d" % (tag + (row.get("VGPRs", 0), row.get("Occupancy", 0), row.get("SGPRs Spill", 0), row.get("VGPRs Spill", 0), row.get("ScratchSize", 0))))
    n_scratch = n_lane = 0
    for tag, body in functions(asm):
        for i, l in enumerate(body):
            code = l.split(";")[0]
            if "v_writelane_b32" in code or "v_readlane_b32" in code:
                n_lane += 1
            if "scratch_" in code:
                n_scratch += 1
                if not spill_lines:
                    continue
                j = i - 1
                while j > 0 and not re.search(r"\bexec\b", body[j].split(";")[0]):
                    j -= 1
                k = i + 1
                while k < len(body) - 1 and not re.search(r"\bexec\b", body[k].split(";")[0]):
                    k += 1
                print("  <%s %s %s %s> line %5d  %-46s  EXEC before (-%3d): %-38s after (+%3d): %s" % (tag + (i, code.strip()[:46], i - j, body[j].strip()[:38], k - i, body[k].strip()[:40])))
    print(f"scratch instructions in all instantiations: {n_scratch};  v_writelane / v_readlane (SGPR spills and cross-lane reads, EXEC-independent): {n_lane}")


with tempfile.TemporaryDirectory() as d:
    for f in ("rt_wavefront.hip", "rt_math.hpp", "rt_scene_dev.h"):
        src = subprocess.run(["git", "-C", ROOT, "show", f"{PRE_FIX}:cuda-ray-tracer_amd/csrc/{f}"], capture_output=True, text=True, check=True).stdout
        open(os.path.join(d, f), "w").write(src)
    asm, rem = compile_fast(d, [])
    report(f"BEFORE the rule: rt_wavefront.hip at git {PRE_FIX}, FMA build (-ffp-contract=fast), this image's hipcc", asm, rem)
with tempfile.TemporaryDirectory() as d:
    for f in ("rt_wavefront.hip", "rt_wavefront_math.hpp", "rt_math.hpp", "rt_scene_dev.h"):
        open(os.path.join(d, f), "w").write(open(os.path.join(CSRC, f)).read())
    asm, rem = compile_fast(d, ["-mllvm", "-amdgpu-kernarg-preload-count=12"])
    report("NOW: the current rt_wavefront.hip, FMA build, same compiler", asm, rem, spill_lines=True)
    ex = [l.strip() for _, body in functions(asm) for l in body if "v_writelane_b32" in l][:3]
    print("SGPR spills look like this (a scalar register parked in ONE lane of a VGPR; v_writelane_b32 / v_readlane_b32 address the lane explicitly and,\n"
          "by the ISA, ignore EXEC -- so which lanes are active when they run does not matter):")
    for l in ex:
        print("   ", l)
print("""
What this shows, and what it does not.
 * The pre-rule FMA build does spill VGPRs to scratch in 8 of its 16 instantiations (the general-quadric one, the one of the round-1 failure: 2 registers, 12 bytes),
   and the current build has no scratch instruction in any instantiation: the make-time check (tools/check_spills.py) and tests/test_abi.py keep it so.
 * Round 1 attributed the failure (160 unwritten tiles / one GPU fault in the first FMA-build frame of a fresh process, never after an earlier launch had run) to a spill
   store placed BEFORE the `s_or_b64 exec` that re-enables the lanes of a divergent region.  That placement is NOT visible in this recompilation: in the general-quadric
   instantiation the store follows the EXEC restore of its block (listed above).  The misplacement is therefore recorded as unconfirmed.  What is established: the failure
   needed (a) a kernel with scratch spills and (b) a process whose scratch memory had not been written before; with (a) removed it has not occurred again (every GPU test
   renders three frames per context, four scene / build combinations are rendered as the first GPU work of a fresh process, 12 000 fuzz scenes in round 1).  The rule
   stays, as a defence whose cost is known (the occupancy targets in wf_occupancy()).
""")
