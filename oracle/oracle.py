"""ctypes front end of the CPU oracle (oracle/rt_oracle.c) -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module (see
oracle/rt_oracle.h).  It also holds an independent scene loader: PyYAML for the syntax + a restatement of
the reference loader's defaults and validation (/root/reference/src/scene.cpp:97-201) feeding the
oracle's C factories, which the product's own C++ YAML loader is compared against.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

NCOEF = 20
COEF_NAMES = ["x3", "y3", "z3", "x2y", "xy2", "x2z", "xz2", "y2z", "yz2", "xyz",
              "x2", "y2", "z2", "xy", "xz", "yz", "x", "y", "z", "c"]


class OrcObject(C.Structure):
    _fields_ = [("c", C.c_double * NCOEF), ("reflection_ratio", C.c_float), ("color", C.c_float * 3)]


class OrcLight(C.Structure):
    _fields_ = [("is_spherical", C.c_int32), ("pad_", C.c_int32), ("p", C.c_double * 3),
                ("color", C.c_float * 3), ("pad2_", C.c_float)]


class OrcScene(C.Structure):
    _fields_ = [("px_width", C.c_uint32), ("px_height", C.c_uint32), ("vertical_fov", C.c_double),
                ("bg_color", C.c_float * 3), ("max_reflections", C.c_uint32),
                ("n_objects", C.c_uint32), ("n_lights", C.c_uint32),
                ("objects", C.POINTER(OrcObject)), ("lights", C.POINTER(OrcLight))]


class OrcCounters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in
                ("primary_rays", "shadow_rays", "reflect_rays", "tests", "br_cardano", "br_trig",
                 "br_quad_hit", "br_quad_miss", "br_linear", "br_none", "normals", "surface_colors")]

    def as_dict(self):
        d = {n: int(getattr(self, n)) for n, _ in self._fields_}
        d["rays_total"] = d["primary_rays"] + d["shadow_rays"] + d["reflect_rays"]
        return d


def build(force=False):
    """Compile librt_oracle.so with the committed Makefile (gcc -O2 -ffp-contract=off)."""
    so = os.path.join(_HERE, "librt_oracle.so")
    src = [os.path.join(_HERE, f) for f in ("rt_oracle.c", "rt_oracle.h", "Makefile")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
        subprocess.run(["make", "-C", _HERE, "-B", "librt_oracle.so"], check=True, capture_output=True)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        dp, fp = C.POINTER(C.c_double), C.POINTER(C.c_float)
        L.orc_radians.restype = C.c_double
        L.orc_radians.argtypes = [C.c_double]
        L.orc_surface_sphere.argtypes = [dp, C.c_double, dp]
        L.orc_surface_plane.argtypes = [dp, dp, dp]
        L.orc_surface_dingdong.argtypes = [dp, dp]
        L.orc_surface_clebsch.argtypes = [dp]
        L.orc_surface_cayley.argtypes = [dp]
        L.orc_light_directional.argtypes = [C.c_float, dp, fp, C.POINTER(OrcLight)]
        L.orc_light_spherical.argtypes = [C.c_float, dp, fp, C.POINTER(OrcLight)]
        L.orc_intersect_ray.restype = C.c_double
        L.orc_intersect_ray.argtypes = [dp, dp, dp]
        L.orc_intersect_ray_ex.restype = C.c_double
        L.orc_intersect_ray_ex.argtypes = [dp, dp, dp, dp, C.POINTER(C.c_int)]
        L.orc_normal_vector.argtypes = [dp, dp, dp]
        L.orc_shadow_ray.argtypes = [C.POINTER(OrcLight), dp, fp, dp]
        L.orc_surface_color.argtypes = [C.POINTER(OrcLight), dp, dp, fp, fp]
        L.orc_reflect_ray.argtypes = [dp, dp, dp]
        L.orc_primary_dir.argtypes = [C.POINTER(OrcScene), dp, C.c_int, C.c_int, dp]
        L.orc_render_pixel.argtypes = [C.POINTER(OrcScene), dp, C.c_int, C.c_int, fp, C.POINTER(OrcCounters)]
        L.orc_render_rows.argtypes = [C.POINTER(OrcScene), dp, C.POINTER(C.c_uint32), C.c_uint32, fp,
                                      C.POINTER(OrcCounters), C.c_int]
        L.orc_checksum.restype = C.c_double
        L.orc_checksum.argtypes = [fp, C.c_uint64]
        L.orc_camera_matrix.argtypes = [dp, C.c_double, C.c_double, dp]
        _LIB = L
    return _LIB


def _d3(v):
    return (C.c_double * 3)(*[float(x) for x in v])


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


class SceneError(Exception):
    """Mirror of SceneException for the independent loader below."""


IDENTITY = np.eye(4, dtype=np.float64).reshape(-1).copy()  # column-major identity dmat4


class Scene:
    """Scene as the oracle sees it: numpy arrays + an OrcScene view of them."""

    def __init__(self, width, height, fov_deg, max_reflections=5, bg_color=(1.0, 1.0, 1.0)):
        self.width, self.height = int(width), int(height)
        self.vertical_fov = lib().orc_radians(float(fov_deg))  # Scene::Scene, scene.cpp:20
        self.fov_deg = float(fov_deg)
        self.max_reflections = int(max_reflections)
        self.bg_color = np.asarray(bg_color, dtype=np.float32)
        self.objects = []  # OrcObject
        self.lights = []  # OrcLight

    # ---- arrays in the layout of the product's rt_scene_desc (include/mi355rt.h) ----
    @property
    def coefs(self):
        return np.array([list(o.c) for o in self.objects], dtype=np.float64).reshape(-1, NCOEF)

    @property
    def reflection(self):
        return np.array([o.reflection_ratio for o in self.objects], dtype=np.float32)

    @property
    def albedo(self):
        return np.array([list(o.color) for o in self.objects], dtype=np.float32).reshape(-1, 3)

    @property
    def light_is_spherical(self):
        return np.array([l.is_spherical for l in self.lights], dtype=np.uint8)

    @property
    def light_p(self):
        return np.array([list(l.p) for l in self.lights], dtype=np.float64).reshape(-1, 3)

    @property
    def light_color(self):
        return np.array([list(l.color) for l in self.lights], dtype=np.float32).reshape(-1, 3)

    def add_object(self, coefs, color, reflection_ratio=0.0):
        o = OrcObject()
        for i, v in enumerate(coefs):
            o.c[i] = float(v)
        o.reflection_ratio = float(reflection_ratio)
        for i in range(3):
            o.color[i] = float(color[i])
        self.objects.append(o)
        return o

    def c_scene(self):
        objs = (OrcObject * max(1, len(self.objects)))(*self.objects)
        lights = (OrcLight * max(1, len(self.lights)))(*self.lights)
        s = OrcScene()
        s.px_width, s.px_height = self.width, self.height
        s.vertical_fov = self.vertical_fov
        for i in range(3):
            s.bg_color[i] = float(self.bg_color[i])
        s.max_reflections = self.max_reflections
        s.n_objects, s.n_lights = len(self.objects), len(self.lights)
        s.objects = C.cast(objs, C.POINTER(OrcObject))
        s.lights = C.cast(lights, C.POINTER(OrcLight))
        s._keep = (objs, lights)
        return s

    def with_size(self, width, height, max_reflections=None):
        """Bench/test override of px_width/px_height/max_reflections (public fields, scene.h:19-22)."""
        import copy
        s = copy.copy(self)
        s.width, s.height = int(width), int(height)
        if max_reflections is not None:
            s.max_reflections = int(max_reflections)
        return s

    # ---- rendering ----
    def render(self, cam=None, rows=None, counters=False, nthreads=1):
        """float32 [n_rows, W, 3], row 0 = bottom (update-cpu.cpp:121-133)."""
        cam = np.ascontiguousarray(IDENTITY if cam is None else cam, dtype=np.float64).reshape(16)
        if rows is None:
            n_rows, rows_p = self.height, None
        else:
            rows = np.ascontiguousarray(rows, dtype=np.uint32)
            n_rows, rows_p = len(rows), rows.ctypes.data_as(C.POINTER(C.c_uint32))
        out = np.empty((n_rows, self.width, 3), dtype=np.float32)
        cnt = OrcCounters() if counters else None
        sc = self.c_scene()
        lib().orc_render_rows(C.byref(sc), cam.ctypes.data_as(C.POINTER(C.c_double)), rows_p, n_rows,
                              out.ctypes.data_as(C.POINTER(C.c_float)),
                              C.byref(cnt) if counters else None, int(nthreads))
        return (out, cnt.as_dict()) if counters else out


def checksum(img):
    img = np.ascontiguousarray(img, dtype=np.float32)
    return lib().orc_checksum(img.ctypes.data_as(C.POINTER(C.c_float)), img.size)


def camera_matrix(pos=(0.0, 0.0, 0.0), yaw_deg=90.0, pitch_deg=0.0):
    out = np.empty(16, dtype=np.float64)
    lib().orc_camera_matrix(_d3(pos), float(yaw_deg), float(pitch_deg), out.ctypes.data_as(C.POINTER(C.c_double)))
    return out


# --------------------------------------------------------------------------------------------------
# Independent scene loader: PyYAML + restated defaults / validation of src/scene.cpp
# --------------------------------------------------------------------------------------------------
def _validate_color(c):
    # validate_color, src/scene-exception.cpp:3-11
    if any((np.float32(x) < 0.0 or np.float32(x) > 1.0) for x in c):
        raise SceneError("Invalid color")


def _validate_positive(what, v):
    # validate_positive, include/scene-exception.h:26-34 (rejects < 0 only)
    if v < 0:
        raise SceneError(f"Negative value for {what}")


def _vec3(node, key, default=None, required=False):
    if key not in node:
        if required:
            raise SceneError(f"Value '{key}' undefined")
        return default
    v = node[key]
    if not isinstance(v, list) or len(v) != 3:
        raise SceneError(f"Value '{key}' is invalid")
    return [float(x) for x in v]


def surface_from_node(node):
    """parse_surface, src/scene.cpp:97-151"""
    L = lib()
    out = (C.c_double * NCOEF)()
    if "type" not in node:
        raise SceneError("Value 'type' undefined")
    t = node["type"]
    if t == "sphere":
        radius = float(node.get("radius", 1.0))
        _validate_positive("sphere radius", radius)
        L.orc_surface_sphere(_d3(_vec3(node, "center", [0.0, 0.0, 0.0])), radius, out)
    elif t == "plane":
        L.orc_surface_plane(_d3(_vec3(node, "origin", [0.0, 0.0, 0.0])), _d3(_vec3(node, "normal", [0.0, 1.0, 0.0])), out)
    elif t == "dingDong":
        L.orc_surface_dingdong(_d3(_vec3(node, "origin", [0.0, 0.0, 0.0])), out)
    elif t == "clebsch":
        L.orc_surface_clebsch(out)
    elif t == "cayley":
        L.orc_surface_cayley(out)
    elif t == "polynomial":
        co = node.get("coefficients")
        if co is None:
            raise SceneError("Value 'coefficients' undefined")
        if not isinstance(co, dict):
            raise SceneError("Value 'coefficients' must be a mapping")
        for i, n in enumerate(COEF_NAMES):
            out[i] = float(co.get(n, 0.0))
    else:
        raise SceneError(f"Unknown surface type: '{t}'")
    return list(out)


def load_scene(path):
    """Scene::load_from_file, src/scene.cpp:154-203 (syntax by PyYAML)."""
    import yaml
    try:
        with open(path) as f:
            d = yaml.safe_load(f)
    except OSError:
        raise SceneError(f"Cannot read the file {path}")
    except yaml.YAMLError as e:
        raise SceneError(f"YAML parser error: {e}")
    for k in ("width", "height", "fov"):
        if k not in d:
            raise SceneError(f"Value '{k}' undefined")
    bg = _vec3(d, "bg_color", [1.0, 1.0, 1.0])  # BG_COLOR default is WHITE, scene.cpp:7
    _validate_color(bg)
    sc = Scene(d["width"], d["height"], d["fov"], d.get("max_reflections", 5), bg)
    for k in ("objects", "light_sources"):
        if k not in d:
            raise SceneError(f"Value '{k}' undefined")
        if not isinstance(d[k], list):
            raise SceneError(f"Value '{k}' must be a sequence")
    for node in d["objects"]:
        coefs = surface_from_node(node)
        refl = np.float32(node.get("reflection_ratio", 0.0))
        color = _vec3(node, "color", required=True)
        _validate_positive("object reflection ratio", refl)
        _validate_color(color)
        sc.add_object(coefs, color, refl)
    L = lib()
    for node in d["light_sources"]:
        if "type" not in node:
            raise SceneError("Value 'type' undefined")
        t = node["type"]
        intensity = float(np.float32(node.get("intensity", 1.0)))
        color = _vec3(node, "color", [1.0, 1.0, 1.0])
        light = OrcLight()
        if t == "directional":
            v = _vec3(node, "direction", required=True)
            _validate_positive("light intensity", intensity)
            _validate_color(color)
            L.orc_light_directional(intensity, _d3(v), _f3(color), C.byref(light))
        elif t == "spherical":
            v = _vec3(node, "position", required=True)
            _validate_positive("light intensity", intensity)
            _validate_color(color)
            L.orc_light_spherical(intensity, _d3(v), _f3(color), C.byref(light))
        else:
            raise SceneError("Light source type must be 'spherical' or 'directional'")
        sc.lights.append(light)
    return sc
