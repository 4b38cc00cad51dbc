"""Python host binding of the MI355X-native ray-tracing path (ctypes over the C ABI, include/mi355rt.h).

The directory name carries a hyphen (it mirrors the reference's repository name), so import it through
``__graft_entry__.load_package()`` which registers it as module ``cuda_ray_tracer_amd``.

Names follow the reference's host interface (include/update.h, include/scene.h):
``Scene.load_from_file``, ``Renderer`` = ``init_update`` / ``update`` / ``cleanup_update``.
There is NO CPU fallback: constructing a ``Renderer`` without a usable GPU raises ``RtError``.
PyTorch is optional plumbing here (device tensors can be handed in as raw pointers + a stream handle).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MI355RT_LIB") or os.path.join(_HERE, "libmi355rt.so")   # MI355RT_LIB: another build of the same ABI, for A/B measurements
UPDATE_LIB_PATH = os.path.join(_HERE, "libmi355rt_update.so")
MULTI_LIB_PATH = os.path.join(_HERE, "libmi355rt_multi.so")   # several GPUs behind one call; the only library that links RCCL

RT_NCOEF = 20
RT_FLAG_STRICT, RT_FLAG_FAST, RT_FLAG_COUNT, RT_FLAG_SIMPLE, RT_FLAG_NOCULL, RT_FLAG_STATIC_ORDER, RT_FLAG_NOSCAN, RT_FLAG_PLAIN_ORDER = 0, 1, 2, 4, 8, 16, 32, 64
RT_FLAG_NOSPLIT = 128
RT_FLAG_NOLEAN = 256
RT_FMT_RGBA32F, RT_FMT_RGBA8 = 0, 1
RT_ERR_NO_DEVICE = -4

# every symbol include/mi355rt.h declares (tests check the built library exports all of them)
ABI_SYMBOLS = [
    "rt_abi_version", "rt_last_error", "rt_set_last_error", "rt_scene_load_file", "rt_scene_new", "rt_scene_add_object",
    "rt_scene_add_light", "rt_surface_make", "rt_scene_set_size", "rt_scene_set_max_reflections",
    "rt_scene_get_desc", "rt_scene_free", "rt_camera_matrix", "rt_create", "rt_render", "rt_local_rows", "rt_max_local_rows",
    "rt_row_map", "rt_pixel_bytes", "rt_device_fb", "rt_download", "rt_assemble", "rt_sparse_bytes", "rt_render_sparse", "rt_pack_sparse", "rt_assemble_sparse", "rt_sparse_stamp_bytes",
    "rt_assemble_sparse_incremental",
    "rt_get_counters", "rt_get_counters_detail", "rt_debug_counters", "rt_debug_stamp_rows", "rt_destroy",
]
# ... and the ones libmi355rt_multi.so exports
MULTI_ABI_SYMBOLS = ["rt_create_multi", "rt_render_multi", "rt_multi_wait", "rt_multi_fb", "rt_multi_stream", "rt_multi_download", "rt_multi_info",
                     "rt_multi_destroy"]
RT_MULTI_SELF_EXCHANGE = 0x10000
RT_MULTI_BANDWISE = 0x20000


class RtError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"mi355rt error {code}: {message}")
        self.code = code
        self.message = message


class SceneException(RtError):
    """Scene description rejected -- what the reference reports as SceneException (scene-exception.h)."""


class SceneDesc(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("vertical_fov", C.c_double),
                ("bg_color", C.c_float * 3), ("max_reflections", C.c_uint32),
                ("n_objects", C.c_uint32), ("n_lights", C.c_uint32),
                ("coefs", C.POINTER(C.c_double)), ("reflection", C.POINTER(C.c_float)),
                ("albedo", C.POINTER(C.c_float)), ("light_is_spherical", C.POINTER(C.c_uint8)),
                ("light_p", C.POINTER(C.c_double)), ("light_color", C.POINTER(C.c_float))]


class Config(C.Structure):
    _fields_ = [("device", C.c_int32), ("rank", C.c_uint32), ("world", C.c_uint32), ("band_rows", C.c_uint32),
                ("flags", C.c_uint32), ("format", C.c_uint32)]


class CountersDetail(C.Structure):
    _fields_ = [("tests_executed", C.c_uint64 * 4), ("solves", C.c_uint64 * 3), ("cull_evals", C.c_uint64 * 5), ("cubic_branch", C.c_uint64 * 4),
                ("shadow_rays_traced", C.c_uint64), ("hit_lights_shaded", C.c_uint64), ("primary_rays_formed", C.c_uint64), ("cubic_points", C.c_uint64), ("cubic_refused", C.c_uint64)]


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("primary_rays", "shadow_rays", "reflect_rays", "tests", "hits", "solves", "tests_executed", "cull_evals")]

    def as_dict(self):
        d = {n: int(getattr(self, n)) for n, _ in self._fields_}
        d["rays_total"] = d["primary_rays"] + d["shadow_rays"] + d["reflect_rays"]
        return d


def build(verbose=False):
    """Compile libmi355rt.so / libmi355rt_update.so in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    r = subprocess.run(["make", "-C", _HERE, "-j4", "all"], capture_output=True, text=True)
    if verbose or r.returncode != 0:
        print(r.stdout[-4000:])
        print(r.stderr[-4000:])
    if r.returncode != 0:
        raise RuntimeError("building libmi355rt.so failed")
    return LIB_PATH


_lib = None


def lib():
    """The loaded C-ABI library.  Fails loudly when it has not been built: there is no fallback path."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RtError(-3, f"{LIB_PATH} is missing: run __graft_entry__.build() (no CPU fallback exists)")
        # PyTorch-ROCm ships its own libamdhip64; if it is going to be used in this process (device tensors,
        # torch.distributed) it must be the copy that gets loaded, so load it BEFORE our library pulls in the
        # system one -- two different HIP runtimes in one process leave the second without devices.
        try:
            import torch  # noqa: F401
        except Exception:  # torch is plumbing, not a requirement of the C ABI
            pass
        L = C.CDLL(LIB_PATH)
        vp, dp, fp = C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_float)
        L.rt_abi_version.restype = C.c_int
        if (L.rt_abi_version() & 0x4000) and not os.environ.get("MI355RT_ALLOW_DIAGNOSTIC"):
            raise RtError(-3, f"{LIB_PATH} is a diagnostic build (make STAMPS=1 / DEBUG_EXITS=1 / SPILLS_OK=1 ...), not the product: rebuild with plain "
                              "`make`, or set MI355RT_ALLOW_DIAGNOSTIC=1 for a measurement")
        L.rt_last_error.restype = C.c_char_p
        L.rt_scene_load_file.argtypes = [C.c_char_p, C.POINTER(vp)]
        L.rt_scene_new.argtypes = [C.c_uint32, C.c_uint32, C.c_double, C.c_uint32, fp, C.POINTER(vp)]
        L.rt_scene_add_object.argtypes = [vp, dp, C.c_float, fp]
        L.rt_scene_add_light.argtypes = [vp, C.c_int, C.c_float, dp, fp]
        L.rt_surface_make.argtypes = [C.c_int, dp, dp, dp]
        L.rt_scene_set_size.argtypes = [vp, C.c_uint32, C.c_uint32]
        L.rt_scene_set_max_reflections.argtypes = [vp, C.c_uint32]
        L.rt_scene_get_desc.argtypes = [vp, C.POINTER(SceneDesc)]
        L.rt_scene_free.argtypes = [vp]
        L.rt_scene_free.restype = None
        L.rt_camera_matrix.argtypes = [dp, C.c_double, C.c_double, dp]
        L.rt_create.argtypes = [C.POINTER(vp), C.POINTER(SceneDesc), C.POINTER(Config)]
        L.rt_render.argtypes = [vp, dp, vp, vp, fp]
        L.rt_local_rows.argtypes = [vp, C.POINTER(C.c_uint32)]
        L.rt_max_local_rows.argtypes = [vp, C.POINTER(C.c_uint32)]
        L.rt_row_map.argtypes = [vp, C.POINTER(C.c_uint32)]
        L.rt_pixel_bytes.argtypes = [vp]
        L.rt_pixel_bytes.restype = C.c_size_t
        L.rt_device_fb.argtypes = [vp]
        L.rt_device_fb.restype = vp
        L.rt_download.argtypes = [vp, vp, C.c_size_t]
        L.rt_assemble.argtypes = [vp, vp, vp, vp]
        L.rt_sparse_bytes.argtypes = [C.c_uint32]
        L.rt_sparse_bytes.restype = C.c_size_t
        L.rt_pack_sparse.argtypes = [vp, vp, vp, C.c_uint32, vp]
        L.rt_render_sparse.argtypes = [vp, dp, vp, C.c_uint32, vp, fp]
        L.rt_assemble_sparse.argtypes = [vp, vp, C.c_uint32, vp, vp]
        L.rt_sparse_stamp_bytes.argtypes = [vp]
        L.rt_sparse_stamp_bytes.restype = C.c_size_t
        L.rt_assemble_sparse_incremental.argtypes = [vp, vp, C.c_uint32, vp, vp, C.c_uint32, vp]
        L.rt_get_counters.argtypes = [vp, C.POINTER(Counters)]
        L.rt_get_counters_detail.argtypes = [vp, C.POINTER(CountersDetail)]
        L.rt_debug_counters.argtypes = [vp, C.POINTER(C.c_uint64)]
        L.rt_debug_stamp_rows.argtypes = [vp, vp, C.c_size_t, C.POINTER(C.c_size_t)]
        L.rt_destroy.argtypes = [vp]
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        msg = lib().rt_last_error().decode("utf-8", "replace")
        raise (SceneException if rc == -2 else RtError)(rc, msg)


def _dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


IDENTITY = np.eye(4, dtype=np.float64).reshape(16).copy()

_mlib = None


def multi_lib():
    """libmi355rt_multi.so (rt_create_multi / rt_render_multi ...).  Loaded on demand: it links RCCL."""
    global _mlib
    if _mlib is None:
        lib()   # the base library (and torch's HIP runtime, if torch is around) first
        if not os.path.exists(MULTI_LIB_PATH):
            raise RtError(-3, f"{MULTI_LIB_PATH} is missing: run __graft_entry__.build()")
        M = C.CDLL(MULTI_LIB_PATH)
        vp = C.c_void_p
        M.rt_create_multi.argtypes = [C.POINTER(vp), C.POINTER(SceneDesc), C.POINTER(C.c_int), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
        M.rt_render_multi.argtypes = [vp, C.POINTER(C.c_double), vp, C.POINTER(C.c_float)]
        M.rt_multi_wait.argtypes = [vp]
        M.rt_multi_fb.argtypes = [vp]
        M.rt_multi_fb.restype = vp
        M.rt_multi_stream.argtypes = [vp]
        M.rt_multi_stream.restype = vp
        M.rt_multi_download.argtypes = [vp, vp, C.c_size_t]
        M.rt_multi_info.argtypes = [vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        M.rt_multi_destroy.argtypes = [vp]
        _mlib = M
    return _mlib


class Scene:
    """Handle on a scene owned by the library (rt_scene).  Mirrors the reference's Scene (scene.h:17-36)."""

    def __init__(self, handle):
        self._h = C.c_void_p(handle)

    @staticmethod
    def load_from_file(path):
        h = C.c_void_p()
        _check(lib().rt_scene_load_file(os.fsencode(path), C.byref(h)))
        return Scene(h.value)

    @staticmethod
    def new(width, height, fov_deg, max_reflections=5, bg_color=(1.0, 1.0, 1.0)):
        h = C.c_void_p()
        bg = np.asarray(bg_color, dtype=np.float32)
        _check(lib().rt_scene_new(int(width), int(height), float(fov_deg), int(max_reflections), _fptr(bg), C.byref(h)))
        return Scene(h.value)

    def add_object(self, coefs, color, reflection_ratio=0.0):
        c = np.ascontiguousarray(coefs, dtype=np.float64)
        assert c.size == RT_NCOEF
        col = np.asarray(color, dtype=np.float32)
        _check(lib().rt_scene_add_object(self._h, _dptr(c), float(reflection_ratio), _fptr(col)))

    def add_light(self, kind, v, color=(1.0, 1.0, 1.0), intensity=1.0):
        vv = np.asarray(v, dtype=np.float64)
        col = np.asarray(color, dtype=np.float32)
        _check(lib().rt_scene_add_light(self._h, 1 if kind == "spherical" else 0, float(intensity), _dptr(vv), _fptr(col)))

    def set_size(self, width, height):
        _check(lib().rt_scene_set_size(self._h, int(width), int(height)))
        return self

    def set_max_reflections(self, n):
        _check(lib().rt_scene_set_max_reflections(self._h, int(n)))
        return self

    def desc(self):
        d = SceneDesc()
        _check(lib().rt_scene_get_desc(self._h, C.byref(d)))
        return d

    def arrays(self):
        """Copies of the flat scene arrays (for tests)."""
        d = self.desc()
        no, nl = d.n_objects, d.n_lights

        def arr(p, n, dt):
            return np.ctypeslib.as_array(p, shape=(n,)).astype(dt).copy() if n else np.zeros(0, dt)
        return dict(width=d.width, height=d.height, vertical_fov=d.vertical_fov, bg_color=np.array(list(d.bg_color), np.float32),
                    max_reflections=d.max_reflections,
                    coefs=arr(d.coefs, no * RT_NCOEF, np.float64).reshape(no, RT_NCOEF), reflection=arr(d.reflection, no, np.float32),
                    albedo=arr(d.albedo, no * 3, np.float32).reshape(no, 3), light_is_spherical=arr(d.light_is_spherical, nl, np.uint8),
                    light_p=arr(d.light_p, nl * 3, np.float64).reshape(nl, 3), light_color=arr(d.light_color, nl * 3, np.float32).reshape(nl, 3))

    def close(self):
        if self._h:
            lib().rt_scene_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def camera_matrix(pos=(0.0, 0.0, 0.0), yaw_deg=90.0, pitch_deg=0.0):
    """The reference host's camera_matrix() (src/ray-tracer.cpp:44-58) for a pose; 16 doubles, column-major."""
    p = np.asarray(pos, dtype=np.float64)
    out = np.empty(16, dtype=np.float64)
    _check(lib().rt_camera_matrix(_dptr(p), float(yaw_deg), float(pitch_deg), _dptr(out)))
    return out


def surface_make(kind, a=None, b=None):
    names = {"sphere": 0, "plane": 1, "dingDong": 2, "clebsch": 3, "cayley": 4}
    out = np.zeros(RT_NCOEF, dtype=np.float64)
    aa = np.asarray(a if a is not None else [0, 0, 0], dtype=np.float64)
    bb = np.asarray(b if b is not None else [0, 0, 0], dtype=np.float64)
    if bb.size == 1:
        bb = np.array([float(bb.reshape(-1)[0]), 0.0, 0.0])
    _check(lib().rt_surface_make(names[kind], _dptr(aa), _dptr(bb), _dptr(out)))
    return out


def desc_from_arrays(width, height, vertical_fov, bg_color, max_reflections, coefs, reflection, albedo,
                     light_is_spherical, light_p, light_color):
    """Build an rt_scene_desc from numpy arrays (kept alive on the returned object)."""
    keep = dict(coefs=np.ascontiguousarray(coefs, np.float64), reflection=np.ascontiguousarray(reflection, np.float32),
                albedo=np.ascontiguousarray(albedo, np.float32), kind=np.ascontiguousarray(light_is_spherical, np.uint8),
                light_p=np.ascontiguousarray(light_p, np.float64), light_color=np.ascontiguousarray(light_color, np.float32))
    d = SceneDesc()
    d.width, d.height, d.vertical_fov, d.max_reflections = int(width), int(height), float(vertical_fov), int(max_reflections)
    for i in range(3):
        d.bg_color[i] = float(bg_color[i])
    d.n_objects, d.n_lights = keep["reflection"].size, keep["kind"].size
    d.coefs, d.reflection, d.albedo = _dptr(keep["coefs"]), _fptr(keep["reflection"]), _fptr(keep["albedo"])
    d.light_is_spherical = keep["kind"].ctypes.data_as(C.POINTER(C.c_uint8))
    d.light_p, d.light_color = _dptr(keep["light_p"]), _fptr(keep["light_color"])
    d._keep = keep
    return d


from .sharding import (band_rows_of_rank, max_local_rows, assemble_index, gather_to_root, assemble_torch, sparse_words, bg_rgba8,  # noqa: E402,F401
                       pack_sparse_numpy, assemble_sparse_numpy)


class Renderer:
    """init_update / update / cleanup_update (reference include/update.h:6-8) as an object."""

    def __init__(self, scene, device=-1, rank=0, world=1, band_rows=8, flags=RT_FLAG_STRICT, fmt=RT_FMT_RGBA32F):
        self._h = None
        d = scene.desc() if isinstance(scene, Scene) else scene
        self._desc = d
        cfg = Config(int(device), int(rank), int(world), int(band_rows), int(flags), int(fmt))
        h = C.c_void_p()
        _check(lib().rt_create(C.byref(h), C.byref(d), C.byref(cfg)))
        self._h = h
        self.width, self.height = d.width, d.height
        self.fmt = fmt
        n = C.c_uint32()
        _check(lib().rt_local_rows(self._h, C.byref(n)))
        self.local_rows = n.value
        _check(lib().rt_max_local_rows(self._h, C.byref(n)))
        self.max_local_rows = n.value
        self.pixel_bytes = lib().rt_pixel_bytes(self._h)

    # init_update is the constructor; these two complete the reference's trio
    def update(self, cam=None, dev_fb=None, stream=None, timed=True):
        """Render one frame; returns device milliseconds (what the reference's update() returns) or None."""
        cam = np.ascontiguousarray(IDENTITY if cam is None else cam, dtype=np.float64).reshape(16)
        ms = C.c_float(0.0)
        _check(lib().rt_render(self._h, _dptr(cam), C.c_void_p(dev_fb) if dev_fb else None,
                               C.c_void_p(stream) if stream else None, C.byref(ms) if timed else None))
        return ms.value if timed else None

    def cleanup_update(self):
        if self._h:
            lib().rt_destroy(self._h)
            self._h = None

    close = cleanup_update

    def __del__(self):
        try:
            self.cleanup_update()
        except Exception:
            pass

    def row_map(self):
        rows = np.zeros(self.local_rows, dtype=np.uint32)
        if self.local_rows:
            _check(lib().rt_row_map(self._h, rows.ctypes.data_as(C.POINTER(C.c_uint32))))
        return rows

    def device_fb(self):
        return lib().rt_device_fb(self._h)

    def download(self):
        """Local rows as numpy: float32 [rows, W, 4] (RGBA32F) or uint8 [rows, W, 4] (RGBA8)."""
        dt = np.uint8 if self.fmt == RT_FMT_RGBA8 else np.float32
        out = np.empty((self.local_rows, self.width, 4), dtype=dt)
        if out.size:
            _check(lib().rt_download(self._h, out.ctypes.data_as(C.c_void_p), out.nbytes))
        return out

    def assemble(self, gathered_ptr, full_ptr, stream=None):
        _check(lib().rt_assemble(self._h, C.c_void_p(gathered_ptr), C.c_void_p(full_ptr), C.c_void_p(stream) if stream else None))

    # sparse transport of an RGBA8 frame (tiles with content only): see include/mi355rt.h
    @staticmethod
    def sparse_bytes(capacity_tiles):
        return int(lib().rt_sparse_bytes(int(capacity_tiles)))

    def update_sparse(self, msg_ptr, capacity_tiles, cam=None, stream=None, timed=True):
        """update() whose output is a sparse message (tiles with hits only) instead of a framebuffer."""
        cam = np.ascontiguousarray(IDENTITY if cam is None else cam, dtype=np.float64).reshape(16)
        ms = C.c_float(0.0)
        _check(lib().rt_render_sparse(self._h, _dptr(cam), C.c_void_p(msg_ptr), int(capacity_tiles), C.c_void_p(stream) if stream else None,
                                      C.byref(ms) if timed else None))
        return ms.value if timed else None

    def pack_sparse(self, msg_ptr, capacity_tiles, fb_ptr=None, stream=None):
        _check(lib().rt_pack_sparse(self._h, C.c_void_p(fb_ptr) if fb_ptr else None, C.c_void_p(msg_ptr), int(capacity_tiles),
                                    C.c_void_p(stream) if stream else None))

    def assemble_sparse(self, gathered_ptr, capacity_tiles, full_ptr, stream=None):
        _check(lib().rt_assemble_sparse(self._h, C.c_void_p(gathered_ptr), int(capacity_tiles), C.c_void_p(full_ptr), C.c_void_p(stream) if stream else None))

    def sparse_stamp_bytes(self):
        return int(lib().rt_sparse_stamp_bytes(self._h))

    def assemble_sparse_incremental(self, gathered_ptr, capacity_tiles, full_ptr, stamps_ptr, frame_tag, stream=None):
        _check(lib().rt_assemble_sparse_incremental(self._h, C.c_void_p(gathered_ptr), int(capacity_tiles), C.c_void_p(full_ptr), C.c_void_p(stamps_ptr),
                                                    int(frame_tag), C.c_void_p(stream) if stream else None))

    def debug_counters(self):
        out = (C.c_uint64 * 32)()
        _check(lib().rt_debug_counters(self._h, out))
        return [int(v) for v in out]

    def stamp_rows(self):
        """Diagnostic builds: [n_rows, 16] uint64 per-wave rows of the last frame (see rt_debug_stamp_rows)."""
        n = C.c_size_t()
        _check(lib().rt_debug_stamp_rows(self._h, None, 0, C.byref(n)))
        out = np.zeros((n.value, 16), dtype=np.uint64)
        if n.value:
            _check(lib().rt_debug_stamp_rows(self._h, out.ctypes.data_as(C.c_void_p), n.value, C.byref(n)))
        return out

    def counters(self):
        c = Counters()
        _check(lib().rt_get_counters(self._h, C.byref(c)))
        return c.as_dict()

    def counters_detail(self):
        """counters() plus the executed work split by surface class / culling kind / cubic solver branch (what bench.py's
        flop accounting multiplies with the per-unit costs of profiles/flop_table.json)."""
        d = self.counters()
        x = CountersDetail()
        _check(lib().rt_get_counters_detail(self._h, C.byref(x)))
        d["executed_by_class"] = dict(zip(("unitsq", "quadric", "linear", "cubic"), (int(v) for v in x.tests_executed)))
        d["solves_by_class"] = dict(zip(("unitsq", "quadric", "linear"), (int(v) for v in x.solves)))
        d["cull_by_kind"] = dict(zip(("tile", "primary", "shadow_directional", "shadow_point", "records"), (int(v) for v in x.cull_evals)))
        d["shadow_rays_traced"], d["hit_lights_shaded"], d["primary_rays_formed"] = int(x.shadow_rays_traced), int(x.hit_lights_shaded), int(x.primary_rays_formed)
        d["cubic_branches"] = dict(zip(("cardano", "trig", "quad", "linear"), (int(v) for v in x.cubic_branch)))
        d["cubic_points"] = int(x.cubic_points)
        d["cubic_refused"] = int(x.cubic_refused)
        return d


class MultiRenderer:
    """init_update / update / cleanup_update over several GPUs of one node (rt_create_multi ...): the frame ends up on devices[0]."""

    def __init__(self, scene, devices, band_rows=16, parts=1, flags=RT_FLAG_STRICT, fmt=RT_FMT_RGBA32F):
        self._h = None
        d = scene.desc() if isinstance(scene, Scene) else scene
        self._desc = d
        devs = (C.c_int * len(devices))(*[int(v) for v in devices])
        h = C.c_void_p()
        _check(multi_lib().rt_create_multi(C.byref(h), C.byref(d), devs, len(devices), int(band_rows), int(parts), int(flags), int(fmt)))
        self._h = h
        self.width, self.height, self.fmt = d.width, d.height, fmt
        n, t = C.c_uint32(), C.c_uint32()
        _check(multi_lib().rt_multi_info(self._h, C.byref(n), C.byref(t)))
        self.n_contexts, self.transport = n.value, {0: "in place", 1: "device copies", 2: "rccl"}[t.value]

    def update(self, cam=None, full_ptr=None, timed=True):
        cam = np.ascontiguousarray(IDENTITY if cam is None else cam, dtype=np.float64).reshape(16)
        ms = C.c_float(0.0)
        _check(multi_lib().rt_render_multi(self._h, _dptr(cam), C.c_void_p(full_ptr) if full_ptr else None, C.byref(ms) if timed else None))
        return ms.value if timed else None

    def wait(self):
        _check(multi_lib().rt_multi_wait(self._h))

    def download(self):
        dt = np.uint8 if self.fmt == RT_FMT_RGBA8 else np.float32
        out = np.empty((self.height, self.width, 4), dtype=dt)
        _check(multi_lib().rt_multi_download(self._h, out.ctypes.data_as(C.c_void_p), out.nbytes))
        return out

    def cleanup_update(self):
        if self._h:
            multi_lib().rt_multi_destroy(self._h)
            self._h = None

    close = cleanup_update

    def __del__(self):
        try:
            self.cleanup_update()
        except Exception:
            pass
