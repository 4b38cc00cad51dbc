import os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["MI355RT_ALLOW_DIAGNOSTIC"] = "1"
import __graft_entry__ as graft
pkg = graft.load_package()
lib = C.CDLL(pkg.LIB_PATH)
lib.rt_wavefront_trips_strict.argtypes = [C.c_void_p, C.c_uint32]
def orbit(i, n=24):
    a = 2.0 * np.pi * i / n
    pos = (5.0 + 14.0 * np.sin(a), 2.0 + 2.0 * np.sin(2 * a), 15.0 - 14.0 * np.cos(a))
    return pkg.camera_matrix(pos, float(np.degrees(np.arctan2(15.0 - pos[2], 5.0 - pos[0]))), float(-np.degrees(np.arctan2(pos[1] - 2.0, 14.0))))
sc = pkg.Scene.load_from_file(os.path.join(ROOT, "scenes", "20spheres.yml")).set_size(1920, 1080)
r = pkg.Renderer(sc, device=0, flags=0)
pose = int(sys.argv[1]); cam = orbit(pose) if pose >= 0 else pkg.IDENTITY
for t in [int(v) for v in sys.argv[2:]]:
    out = (C.c_ulonglong * 8)()
    for _ in range(3): r.update(cam)
    lib.rt_wavefront_trips_strict(out, t)
    ms = r.update(cam)
    lib.rt_wavefront_trips_strict(out, t)
    o = list(out)
    print(f"pose {pose} tile {t}: items {o[3]}  survivor iterations {o[4]} ({o[4]/max(o[3],1):.1f}/item)  items with a solve {o[0]}  lanes with candidates {o[5]}  solve trips {o[1]} ({o[1]/max(o[0],1):.1f}/item with)  lanes active in trips {o[2]} ({o[2]/max(o[1],1):.1f}/trip)")
