import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import __graft_entry__ as graft
pkg = graft.load_package()
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
base = pkg.Scene.load_from_file(os.path.join(ROOT, "scenes", "20spheres.yml")).set_size(1920, 1080)
a = base.arrays()
for nl in (0, 1, 4, 10, 19):
    d = pkg.desc_from_arrays(1920, 1080, a["vertical_fov"], a["bg_color"], a["max_reflections"], a["coefs"], a["reflection"], a["albedo"],
                             a["light_is_spherical"][:nl], a["light_p"][:nl], a["light_color"][:nl])
    r = pkg.Renderer(d, device=0)
    for _ in range(3): r.update()
    t = np.median([r.update() for _ in range(30)])
    r.cleanup_update()
    print(f"lights={nl:2d}: {t*1e3:7.1f} us")
