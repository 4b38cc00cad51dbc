#!/usr/bin/env python3
"""Re-author the reference's scene descriptions as this repo's own scene files.

Runs only in the build container (it reads /root/reference/scenes/*.yml with PyYAML); the files it
writes under scenes/ are what tests, bench.py and the GPU box use -- /root/reference does not travel.
Only the numeric content is carried over; layout, key order and comments are this repo's.
tests/test_scenes.py re-checks value equality against the reference whenever it is present.

Format spec followed: /root/reference/presentation/Instrukcja.md:17-33, keys/defaults
/root/reference/src/scene.cpp:97-201.
"""
import os
import sys

import yaml

REF = "/root/reference/scenes"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scenes")

COEF_ORDER = ["x3", "y3", "z3", "x2y", "xy2", "x2z", "xz2", "y2z", "yz2", "xyz",
              "x2", "y2", "z2", "xy", "xz", "yz", "x", "y", "z", "c"]


def num(v):
    return repr(v)


def seq(v):
    return "[" + ", ".join(num(x) for x in v) + "]"


def emit(name, d):
    lines = [f"# {name}: {len(d['objects'])} object(s), {len(d['light_sources'])} light(s)",
             "# scene description for the MI355X implicit-surface ray tracer (units: world space, degrees)",
             f"width: {num(d['width'])}", f"height: {num(d['height'])}", f"fov: {num(d['fov'])}"]
    if "max_reflections" in d:
        lines.append(f"max_reflections: {num(d['max_reflections'])}")
    if "bg_color" in d:
        lines.append(f"bg_color: {seq(d['bg_color'])}")
    lines.append("")
    lines.append("light_sources:")
    for l in d["light_sources"]:
        lines.append(f"  - type: {l['type']}")
        for k in ("direction", "position"):
            if k in l:
                lines.append(f"    {k}: {seq(l[k])}")
        if "intensity" in l:
            lines.append(f"    intensity: {num(l['intensity'])}")
        if "color" in l:
            lines.append(f"    color: {seq(l['color'])}")
    lines.append("")
    lines.append("objects:")
    for o in d["objects"]:
        lines.append(f"  - type: {o['type']}")
        for k in ("center", "origin", "normal"):
            if k in o:
                lines.append(f"    {k}: {seq(o[k])}")
        if "radius" in o:
            lines.append(f"    radius: {num(o['radius'])}")
        if "coefficients" in o:
            lines.append("    coefficients:")
            for c in COEF_ORDER:
                if c in o["coefficients"]:
                    lines.append(f"      {c}: {num(o['coefficients'][c])}")
            extra = set(o["coefficients"]) - set(COEF_ORDER)
            assert not extra, extra
        lines.append(f"    color: {seq(o['color'])}")
        if "reflection_ratio" in o:
            lines.append(f"    reflection_ratio: {num(o['reflection_ratio'])}")
        known = {"type", "center", "origin", "normal", "radius", "coefficients", "color", "reflection_ratio"}
        assert set(o) <= known, set(o) - known
    return "\n".join(lines) + "\n"


def main():
    if not os.path.isdir(REF):
        sys.exit("reference scenes not present; nothing to do")
    os.makedirs(OUT, exist_ok=True)
    for fn in sorted(os.listdir(REF)):
        if not fn.endswith(".yml"):
            continue
        with open(os.path.join(REF, fn)) as f:
            d = yaml.safe_load(f)
        text = emit(fn[:-4], d)
        assert yaml.safe_load(text) == d, fn
        with open(os.path.join(OUT, fn), "w") as f:
            f.write(text)
        print("wrote", fn)


if __name__ == "__main__":
    main()
