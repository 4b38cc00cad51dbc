// rt_capi.cpp -- C ABI of libmi355rt.so (include/mi355rt.h): scene handles, render contexts, launches.
//
// Host side of the HIP path.  No CPU fallback: every render entry point fails with RT_ERR_NO_DEVICE /
// RT_ERR_DEVICE when there is no usable GPU.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "mi355rt.h"
#include "rt_scene_dev.h"
#include "rt_math.hpp" // (rtm::cubic_at, host side: the Taylor data of degree-3 objects at the frame's ray origin)
#include "scene-exception.h"
#include "scene.h"
#include "camera.h"

// kernels, one set per floating-point contraction mode (rt_kernels.hip)
extern "C" hipError_t rt_launch_trace_strict(const FrameArgs *, const DevObject *, const DevLight *, void *, unsigned long long *, int, int, hipStream_t);
extern "C" hipError_t rt_launch_trace_fast(const FrameArgs *, const DevObject *, const DevLight *, void *, unsigned long long *, int, int, hipStream_t);
extern "C" hipError_t rt_launch_wavefront_strict(const FrameArgs *, const DevObject *, const DevLight *, void *, unsigned long long *, int, const double *, const double *, hipStream_t);
extern "C" hipError_t rt_launch_wavefront_fast(const FrameArgs *, const DevObject *, const DevLight *, void *, unsigned long long *, int, const double *, const double *, hipStream_t);

extern "C" size_t rt_wavefront_lds_bytes_strict(uint32_t, uint32_t, int, uint32_t, int, uint32_t);

extern "C" hipError_t rt_launch_assemble_strict(const void *, void *, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, int, hipStream_t);
extern "C" hipError_t rt_launch_pack_sparse_strict(const void *, void *, uint32_t, uint32_t, uint32_t, uint32_t, hipStream_t);
extern "C" hipError_t rt_launch_assemble_sparse_strict(const void *, void *, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, void *, uint32_t, uint32_t, hipStream_t);

namespace {

thread_local std::string g_last_error;

int fail(int code, const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

#define RT_HIP(call)                                                                                         \
    do {                                                                                                     \
        hipError_t e_ = (call);                                                                              \
        if (e_ != hipSuccess) return fail(RT_ERR_DEVICE, "%s failed: %s", #call, hipGetErrorString(e_));      \
    } while (0)

} // namespace

// Flat arrays the descriptor points into, kept next to the C++ scene model.
struct rt_scene {
    Scene scene;
    std::vector<double> coefs, light_p;
    std::vector<float> reflection, albedo, light_color;
    std::vector<uint8_t> light_kind;

    void flatten()
    {
        const size_t no = scene.objects.size(), nl = scene.lights.size();
        coefs.resize(no * RT_NCOEF);
        reflection.resize(no);
        albedo.resize(no * 3);
        for (size_t i = 0; i < no; i++) {
            const Object &o = scene.objects[i];
            std::memcpy(&coefs[i * RT_NCOEF], o.surface.data(), sizeof(double) * RT_NCOEF);
            reflection[i] = o.reflection_ratio;
            albedo[3 * i + 0] = o.color.x;
            albedo[3 * i + 1] = o.color.y;
            albedo[3 * i + 2] = o.color.z;
        }
        light_p.resize(nl * 3);
        light_color.resize(nl * 3);
        light_kind.resize(nl);
        for (size_t i = 0; i < nl; i++) {
            const LightSource &l = scene.lights[i];
            light_kind[i] = l.is_spherical ? 1 : 0;
            for (int k = 0; k < 3; k++) {
                light_p[3 * i + k] = l.p[k];
                light_color[3 * i + k] = l.light_color[k];
            }
        }
    }
};

struct rt_ctx {
    int device = 0;
    rt_config cfg{};
    FrameArgs fa{};
    uint32_t local_rows = 0, max_local_rows = 0;
    size_t pixel_bytes = 16;
    DevObject *d_obj = nullptr;
    DevLight *d_light = nullptr;
    void *d_fb = nullptr;
    unsigned long long *d_counters = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t ev_done = nullptr;      // recorded behind every render: a render on ANOTHER stream waits for it (frames of a context are ordered)

    hipStream_t last_stream = nullptr;
    bool rendered = false;
    bool captured = false;      // the last render was recorded into a stream capture: ev_done was not (an event recorded inside a capture orders nothing outside it)
    bool counted = false;
    bool zero_counters = false; // diagnostic builds: clear counters[] before every render
    uint64_t *d_stamps = nullptr;
    double *d_camx = nullptr, *d_camy = nullptr; // per-column / per-row camera-plane coordinates
    size_t n_stamp_rows = 0;
    uint64_t frame = 0; // renders so far: selects the launch-order generation (FrameArgs::order_state); 64 bits: frame % 3 must never skip
    uint32_t tag = 0;   // frame tag of the scan workgroups' tile words (FrameArgs::tile_state); unique per render, never 0
    uint32_t *h_listed = nullptr; // host-mapped words the kernel writes (FrameArgs::ord_host)
    uint32_t ord_split = 0;       // FrameArgs::ord_split of non-sparse frames
    std::vector<double> cub_coefs; // the 20 coefficients of the first RT_CUB_AT_MAX degree-3 objects (FrameArgs::cub_at is formed from them every frame)
    bool lean_ok = false;         // the scene qualifies for the wave-per-block instantiation (FrameArgs::lean; dense frames only)
    bool lean_now = true;         // ... and it renders the current frames (it does not while few tiles have hits: see render_impl)
    uint32_t wg_slots = 1536;     // workgroup slots of the device for these kernels (six per CU)
    int lean_force = 0;           // MI355RT_LEAN=always / never (experiments)
    bool ord_on = true;           // launch-order feedback in use (off while most tiles have hits)
};

// ---------------------------------------------------------------------------------------------------
#ifdef RT_DIAGNOSTIC_BUILD
extern "C" int rt_abi_version(void) { return RT_ABI_VERSION | RT_ABI_DIAGNOSTIC; } // stamps / experiment exits / spills allowed: not the product
#else
extern "C" int rt_abi_version(void) { return RT_ABI_VERSION; }
#endif

extern "C" const char *rt_last_error(void) { return g_last_error.c_str(); }

extern "C" void rt_set_last_error(const char *message) { g_last_error = message ? message : ""; }

// ---- scene --------------------------------------------------------------------------------------------
extern "C" int rt_scene_load_file(const char *path, rt_scene **out)
{
    if (!path || !out) return fail(RT_ERR_INVALID, "rt_scene_load_file: null argument");
    *out = nullptr;
    try {
        rt_scene *s = new rt_scene();
        try {
            s->scene = Scene::load_from_file(path);
        } catch (...) {
            delete s;
            throw;
        }
        s->flatten();
        *out = s;
        return RT_OK;
    } catch (const SceneException &e) {
        return fail(RT_ERR_SCENE, "%s", e.what());
    } catch (const std::bad_alloc &) {
        return fail(RT_ERR_NOMEM, "out of memory");
    } catch (const std::exception &e) {
        return fail(RT_ERR_SCENE, "%s", e.what());
    }
}

extern "C" int rt_scene_new(uint32_t width, uint32_t height, double fov_deg, uint32_t max_reflections,
                            const float bg_color[3], rt_scene **out)
{
    if (!out || !bg_color) return fail(RT_ERR_INVALID, "rt_scene_new: null argument");
    *out = nullptr;
    try {
        rt_scene *s = new rt_scene();
        try {
            s->scene = Scene(width, height, fov_deg, max_reflections, glm::vec3(bg_color[0], bg_color[1], bg_color[2]));
        } catch (...) {
            delete s;
            throw;
        }
        s->flatten();
        *out = s;
        return RT_OK;
    } catch (const SceneException &e) {
        return fail(RT_ERR_SCENE, "%s", e.what());
    } catch (const std::exception &e) {
        return fail(RT_ERR_NOMEM, "%s", e.what());
    }
}

extern "C" int rt_scene_add_object(rt_scene *s, const double coefs[RT_NCOEF], float reflection_ratio, const float color[3])
{
    if (!s || !coefs || !color) return fail(RT_ERR_INVALID, "rt_scene_add_object: null argument");
    try {
        SurfaceCoefs sc{};
        std::memcpy(sc.data(), coefs, sizeof(double) * RT_NCOEF);
        s->scene.objects.push_back(Object(sc, reflection_ratio, glm::vec3(color[0], color[1], color[2])));
        s->flatten();
        return RT_OK;
    } catch (const SceneException &e) {
        return fail(RT_ERR_SCENE, "%s", e.what());
    } catch (const std::exception &e) {
        return fail(RT_ERR_NOMEM, "%s", e.what());
    }
}

extern "C" int rt_scene_add_light(rt_scene *s, int is_spherical, float intensity, const double v[3], const float color[3])
{
    if (!s || !v || !color) return fail(RT_ERR_INVALID, "rt_scene_add_light: null argument");
    try {
        const glm::dvec3 dv(v[0], v[1], v[2]);
        const glm::vec3 c(color[0], color[1], color[2]);
        s->scene.lights.push_back(is_spherical ? LightSource::spherical(intensity, dv, c) : LightSource::directional(intensity, dv, c));
        s->flatten();
        return RT_OK;
    } catch (const SceneException &e) {
        return fail(RT_ERR_SCENE, "%s", e.what());
    } catch (const std::exception &e) {
        return fail(RT_ERR_NOMEM, "%s", e.what());
    }
}

extern "C" int rt_surface_make(int kind, const double a[3], const double b[3], double out_coefs[RT_NCOEF])
{
    if (!out_coefs) return fail(RT_ERR_INVALID, "rt_surface_make: null output");
    if ((kind <= 2 && !a) || (kind <= 1 && !b)) return fail(RT_ERR_INVALID, "rt_surface_make: null argument");
    try {
        SurfaceCoefs sc{};
        switch (kind) {
        case 0: sc = SurfaceCoefs::sphere(glm::dvec3(a[0], a[1], a[2]), b[0]); break;
        case 1: sc = SurfaceCoefs::plane(glm::dvec3(a[0], a[1], a[2]), glm::dvec3(b[0], b[1], b[2])); break;
        case 2: sc = SurfaceCoefs::dingDong(glm::dvec3(a[0], a[1], a[2])); break;
        case 3: sc = SurfaceCoefs::clebsch(); break;
        case 4: sc = SurfaceCoefs::cayley(); break;
        default: return fail(RT_ERR_INVALID, "rt_surface_make: unknown kind %d", kind);
        }
        std::memcpy(out_coefs, sc.data(), sizeof(double) * RT_NCOEF);
        return RT_OK;
    } catch (const SceneException &e) {
        return fail(RT_ERR_SCENE, "%s", e.what());
    }
}

extern "C" int rt_scene_set_size(rt_scene *s, uint32_t width, uint32_t height)
{
    if (!s) return fail(RT_ERR_INVALID, "rt_scene_set_size: null scene");
    s->scene.px_width = width;
    s->scene.px_height = height;
    return RT_OK;
}

extern "C" int rt_scene_set_max_reflections(rt_scene *s, uint32_t max_reflections)
{
    if (!s) return fail(RT_ERR_INVALID, "rt_scene_set_max_reflections: null scene");
    s->scene.max_reflections = max_reflections;
    return RT_OK;
}

extern "C" int rt_scene_get_desc(const rt_scene *s, rt_scene_desc *out)
{
    if (!s || !out) return fail(RT_ERR_INVALID, "rt_scene_get_desc: null argument");
    std::memset(out, 0, sizeof(*out));
    out->width = s->scene.px_width;
    out->height = s->scene.px_height;
    out->vertical_fov = s->scene.vertical_fov;
    out->bg_color[0] = s->scene.bg_color.x;
    out->bg_color[1] = s->scene.bg_color.y;
    out->bg_color[2] = s->scene.bg_color.z;
    out->max_reflections = s->scene.max_reflections;
    out->n_objects = (uint32_t) s->scene.objects.size();
    out->n_lights = (uint32_t) s->scene.lights.size();
    out->coefs = s->coefs.data();
    out->reflection = s->reflection.data();
    out->albedo = s->albedo.data();
    out->light_is_spherical = s->light_kind.data();
    out->light_p = s->light_p.data();
    out->light_color = s->light_color.data();
    return RT_OK;
}

extern "C" void rt_scene_free(rt_scene *s) { delete s; }

extern "C" int rt_camera_matrix(const double pos[3], double yaw_deg, double pitch_deg, double out_cam[16])
{
    if (!pos || !out_cam) return fail(RT_ERR_INVALID, "rt_camera_matrix: null argument");
    Camera c;
    c.position = glm::dvec3(pos[0], pos[1], pos[2]);
    c.yaw = yaw_deg;
    c.pitch = pitch_deg;
    const glm::dmat4 m = c.matrix();
    for (int col = 0; col < 4; col++)
        for (int row = 0; row < 4; row++) out_cam[col * 4 + row] = m[col][row];
    return RT_OK;
}

// ---- render -------------------------------------------------------------------------------------------
static uint32_t classify(const double *c)
{
    uint32_t cls = 0;
    for (int i = K_X3; i <= K_XYZ; i++)
        if (c[i] != 0.0) cls |= RT_CLS_CUBIC;
    if (cls & RT_CLS_CUBIC) return RT_CLS_CUBIC; // dense path handles everything
    if (c[K_X2] != 0.0 || c[K_Y2] != 0.0 || c[K_Z2] != 0.0) cls |= RT_CLS_SQUARE;
    if (c[K_XY] != 0.0 || c[K_XZ] != 0.0 || c[K_YZ] != 0.0) cls |= RT_CLS_CROSS;
    if (!(cls & RT_CLS_CROSS) && c[K_X2] == 1.0 && c[K_Y2] == 1.0 && c[K_Z2] == 1.0) cls |= RT_CLS_UNITSQ;
    return cls;
}

static uint32_t rows_of_rank(uint32_t height, uint32_t band, uint32_t world, uint32_t rank)
{
    // bands b = rank, rank + world, ... ; the last band of the image may be partial
    uint32_t n_bands = (height + band - 1) / band, rows = 0;
    for (uint32_t b = rank; b < n_bands; b += world) {
        uint32_t y0 = b * band;
        rows += (y0 + band <= height) ? band : height - y0;
    }
    return rows;
}

static int create_impl(rt_ctx **out, const rt_scene_desc *sd, const rt_config *cfg_in);

extern "C" int rt_create(rt_ctx **out, const rt_scene_desc *sd, const rt_config *cfg_in)
{
    try { // host-side packing allocates; nothing may propagate through the C ABI
        return create_impl(out, sd, cfg_in);
    } catch (const std::bad_alloc &) {
        return fail(RT_ERR_NOMEM, "rt_create: out of host memory");
    } catch (const std::exception &e) {
        return fail(RT_ERR_INVALID, "rt_create: %s", e.what());
    }
}

static int create_impl(rt_ctx **out, const rt_scene_desc *sd, const rt_config *cfg_in)
{
    if (!out || !sd) return fail(RT_ERR_INVALID, "rt_create: null argument");
    *out = nullptr;
    rt_config cfg{};
    cfg.device = -1;
    cfg.world = 1;
    if (cfg_in) cfg = *cfg_in;
    if (cfg.world == 0) cfg.world = 1;
    if (cfg.band_rows == 0) cfg.band_rows = 8;
    if (cfg.rank >= cfg.world) return fail(RT_ERR_INVALID, "rt_create: rank %u >= world %u", cfg.rank, cfg.world);
    if (cfg.format > RT_FMT_RGBA8) return fail(RT_ERR_INVALID, "rt_create: unknown format %u", cfg.format);
    if (sd->width == 0 || sd->height == 0) return fail(RT_ERR_INVALID, "rt_create: empty image %ux%u", sd->width, sd->height);
    if ((sd->n_objects && (!sd->coefs || !sd->reflection || !sd->albedo)) ||
        (sd->n_lights && (!sd->light_is_spherical || !sd->light_p || !sd->light_color)))
        return fail(RT_ERR_INVALID, "rt_create: null scene array");

    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0)
        return fail(RT_ERR_NO_DEVICE, "rt_create: no HIP device available (%s); this library has no CPU fallback",
                    e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
    int device = cfg.device;
    if (device < 0) RT_HIP(hipGetDevice(&device));
    if (device >= ndev) return fail(RT_ERR_INVALID, "rt_create: device %d out of range (%d devices)", device, ndev);
    RT_HIP(hipSetDevice(device));

    // owned by a guard until the very end: whatever throws or fails on the way (host-side packing allocates), rt_destroy releases
    // the context and every device buffer it already holds
    struct Guard {
        rt_ctx *p;
        ~Guard() { if (p) { std::string keep = g_last_error; rt_destroy(p); g_last_error = keep; } }
    } guard{new (std::nothrow) rt_ctx()};
    rt_ctx *ctx = guard.p;
    if (!ctx) return fail(RT_ERR_NOMEM, "out of memory");
    ctx->device = device;
    ctx->cfg = cfg;
    ctx->pixel_bytes = cfg.format == RT_FMT_RGBA8 ? 4 : 16;
    ctx->local_rows = rows_of_rank(sd->height, cfg.band_rows, cfg.world, cfg.rank);
    for (uint32_t r = 0; r < cfg.world; r++) {
        uint32_t n = rows_of_rank(sd->height, cfg.band_rows, cfg.world, r);
        if (n > ctx->max_local_rows) ctx->max_local_rows = n;
    }

    FrameArgs &fa = ctx->fa;
    std::memset(&fa, 0, sizeof(fa));
    fa.aspect = (double) sd->width / sd->height;       // Scene::aspect_ratio, include/scene.h:32-33
    fa.tan_half_fov = std::tan(0.5 * sd->vertical_fov); // init_update, src/update-cpu.cpp:28
    fa.bg[0] = sd->bg_color[0];
    fa.bg[1] = sd->bg_color[1];
    fa.bg[2] = sd->bg_color[2];
    fa.bg[3] = 1.0f;
    fa.width = sd->width;
    fa.height = sd->height;
    fa.n_obj = sd->n_objects;
    fa.n_lights = sd->n_lights;
    fa.max_refl = sd->max_reflections;
    fa.rank = cfg.rank;
    fa.world = cfg.world;
    fa.band_rows = cfg.band_rows;
    fa.local_rows = ctx->local_rows;
    fa.tiles_x = (sd->width + RT_TILE - 1) / RT_TILE;
    fa.n_tiles = fa.tiles_x * ((ctx->local_rows + RT_TILE - 1) / RT_TILE);
    fa.rgba8 = cfg.format == RT_FMT_RGBA8 ? 1u : 0u;
    fa.ord_plain = (cfg.flags & RT_FLAG_PLAIN_ORDER) ? 1u : 0u;
    ctx->ord_split = (cfg.flags & RT_FLAG_NOSPLIT) ? 0u : RT_ORD_SPLIT_CLASSES;
    if (const char *e = std::getenv("MI355RT_SPLIT_CLASSES")) ctx->ord_split = (uint32_t) std::atoi(e) & 15u; // (experiments)
    fa.has_mirror = 0;
    for (uint32_t i = 0; i < sd->n_objects; i++)
        if ((double) sd->reflection[i] > 1e-7) fa.has_mirror = 1; // EPS of the reflection loop, src/update-cpu.cpp:101
    std::vector<DevObject> objs(sd->n_objects);
    uint32_t n_cullable = 0;
    for (uint32_t i = 0; i < sd->n_objects; i++) {
        DevObject &o = objs[i];
        std::memset(&o, 0, sizeof(o));
        std::memcpy(o.c, sd->coefs + (size_t) i * RT_NCOEF, sizeof(double) * RT_NCOEF);
        o.albedo[0] = sd->albedo[3 * i + 0];
        o.albedo[1] = sd->albedo[3 * i + 1];
        o.albedo[2] = sd->albedo[3 * i + 2];
        o.refl = sd->reflection[i];
        o.cls = classify(o.c);
        // bounding sphere of a sphere: centre -k/2, r^2 = |centre|^2 - c (src/surface.cpp:4-15 inverted)
        o.bs_radius = INFINITY;
        if (o.cls & RT_CLS_UNITSQ) {
            const double cx = -0.5 * o.c[K_X], cy = -0.5 * o.c[K_Y], cz = -0.5 * o.c[K_Z];
            const double r2 = cx * cx + cy * cy + cz * cz - o.c[K_C];
            if (r2 > 0.0 && std::isfinite(r2)) {
                o.bs_center[0] = cx;
                o.bs_center[1] = cy;
                o.bs_center[2] = cz;
                o.bs_radius = std::sqrt(r2);
                n_cullable++;
            }
        }
    }
    // culling costs one bounding-volume decision per (object, light, 64-hit chunk); worth it from a handful
    // of bounded objects upwards
    fa.cull = (!(cfg.flags & RT_FLAG_NOCULL) && n_cullable >= 4) ? 1u : 0u;
    fa.all_cullable = (fa.cull && n_cullable == sd->n_objects) ? 1u : 0u;

    // per-class tables behind the object array (rt_scene_dev.h)
    std::vector<UsEntry> t_us;
    std::vector<GqEntry> t_gq;
    std::vector<LinEntry> t_lin;
    std::vector<uint32_t> t_cub;
    for (uint32_t i = 0; i < sd->n_objects; i++) {
        const DevObject &o = objs[i];
        if (o.cls & RT_CLS_CUBIC) {
            if (t_cub.size() < RT_CUB_AT_MAX) ctx->cub_coefs.insert(ctx->cub_coefs.end(), o.c, o.c + RT_NCOEF);
            t_cub.push_back(i);
        } else if (o.cls & RT_CLS_UNITSQ) {
            UsEntry e{};
            e.kx = o.c[K_X]; e.ky = o.c[K_Y]; e.kz = o.c[K_Z]; e.c = o.c[K_C];
            e.r = o.bs_radius;
            e.inv_r = (o.bs_radius < INFINITY) ? 1.0 / o.bs_radius : 0.0;
            e.orig = i;
            // window of the reference's own t0 inside which a shadow ray leaving this sphere towards a directional light in front of the
            // surface cannot be blocked by this sphere (rt_wavefront.hip, own_sphere_skippable): (1e-10 (r^2 + 1) + 1e-20 S^2, (r + 1)^2),
            // S = 2 |centre|_1 + 3 r + 3; rounded inwards to FP32.  No window (+inf, 0) for spheres without a real radius.
            e.own_lo = INFINITY;
            e.own_hi = 0.0f;
            if (o.bs_radius < INFINITY && o.bs_radius > 0.0) {
                const double r = o.bs_radius, S = 2.0 * (std::fabs(o.bs_center[0]) + std::fabs(o.bs_center[1]) + std::fabs(o.bs_center[2])) + 3.0 * r + 3.0;
                const double lo = 1e-10 * (r * r + 1.0) + 1e-20 * S * S, hi = (r + 1.0) * (r + 1.0);
                float flo = (float) lo, fhi = (float) hi;
                if (!((double) flo > lo)) flo = std::nextafterf(flo, INFINITY);
                if (!((double) fhi < hi)) fhi = std::nextafterf(fhi, -INFINITY);
                if (std::isfinite(lo) && std::isfinite(hi) && (double) flo > lo && (double) fhi < hi && flo < fhi) {
                    e.own_lo = flo;
                    e.own_hi = fhi;
                }
            }
            t_us.push_back(e);
        } else if (o.cls & (RT_CLS_SQUARE | RT_CLS_CROSS)) {
            GqEntry e{};
            e.x2 = o.c[K_X2]; e.y2 = o.c[K_Y2]; e.z2 = o.c[K_Z2];
            e.xy = o.c[K_XY]; e.xz = o.c[K_XZ]; e.yz = o.c[K_YZ];
            e.kx = o.c[K_X]; e.ky = o.c[K_Y]; e.kz = o.c[K_Z]; e.c = o.c[K_C];
            e.orig = i;
            t_gq.push_back(e);
        } else {
            LinEntry e{};
            e.kx = o.c[K_X]; e.ky = o.c[K_Y]; e.kz = o.c[K_Z]; e.c = o.c[K_C];
            e.orig = i;
            t_lin.push_back(e);
        }
    }
    auto up16 = [](size_t v) { return (v + 15) & ~(size_t) 15; };
    fa.n_us = (uint32_t) t_us.size();
    fa.n_gq = (uint32_t) t_gq.size();
    fa.n_lin = (uint32_t) t_lin.size();
    fa.n_cub = (uint32_t) t_cub.size();
    size_t off = up16(sizeof(DevObject) * objs.size());
    fa.off_us = (uint32_t) off; off = up16(off + sizeof(UsEntry) * t_us.size());
    fa.off_gq = (uint32_t) off; off = up16(off + sizeof(GqEntry) * t_gq.size());
    fa.off_lin = (uint32_t) off; off = up16(off + sizeof(LinEntry) * t_lin.size());
    fa.off_cub = (uint32_t) off; off = up16(off + sizeof(uint32_t) * t_cub.size());
    fa.off_mat = (uint32_t) off; off = up16(off + sizeof(MatEntry) * objs.size());
    fa.scene_bytes = (uint32_t) (off ? off : 16);
    fa.stage_bytes = fa.scene_bytes - fa.off_us;
    std::vector<unsigned char> blob(fa.scene_bytes, 0);
    if (!objs.empty()) std::memcpy(blob.data(), objs.data(), sizeof(DevObject) * objs.size());
    if (!t_us.empty()) std::memcpy(blob.data() + fa.off_us, t_us.data(), sizeof(UsEntry) * t_us.size());
    if (!t_gq.empty()) std::memcpy(blob.data() + fa.off_gq, t_gq.data(), sizeof(GqEntry) * t_gq.size());
    if (!t_lin.empty()) std::memcpy(blob.data() + fa.off_lin, t_lin.data(), sizeof(LinEntry) * t_lin.size());
    if (!t_cub.empty()) std::memcpy(blob.data() + fa.off_cub, t_cub.data(), sizeof(uint32_t) * t_cub.size());
    for (size_t i = 0; i < objs.size(); i++) {
        MatEntry m{};
        m.albedo[0] = objs[i].albedo[0];
        m.albedo[1] = objs[i].albedo[1];
        m.albedo[2] = objs[i].albedo[2];
        m.refl = objs[i].refl;
        std::memcpy(blob.data() + fa.off_mat + i * sizeof(MatEntry), &m, sizeof(m));
    }
    std::vector<DevLight> lights(sd->n_lights);
    for (uint32_t i = 0; i < sd->n_lights; i++) {
        DevLight &l = lights[i];
        std::memset(&l, 0, sizeof(l));
        for (int k = 0; k < 3; k++) {
            l.p[k] = sd->light_p[3 * i + k];
            l.color[k] = sd->light_color[3 * i + k];
        }
        l.spherical = sd->light_is_spherical[i] ? 1u : 0u;
        for (int k = 0; k < 3; k++) l.sdir[k] = (double) (float) l.p[k];
        l.dxx = l.sdir[0] * l.sdir[0];
        l.dyy = l.sdir[1] * l.sdir[1];
        l.dzz = l.sdir[2] * l.sdir[2];
        l.dxy = l.sdir[0] * l.sdir[1];
        l.dxz = l.sdir[0] * l.sdir[2];
        l.dyz = l.sdir[1] * l.sdir[2];
        l.u2 = (l.dxx + l.dyy) + l.dzz;
        l.inv_uu = l.u2 > 0.0 ? 1.0 / l.u2 : 0.0;
        l.len_u = 1.001 * std::sqrt(l.u2);
        bool finite = std::isfinite(l.color[0]) && std::isfinite(l.color[1]) && std::isfinite(l.color[2]);
        for (uint32_t k = 0; k < sd->n_objects * 3u && finite; k++) finite = std::isfinite(sd->albedo[k]);
        l.backface_exact = (!l.spherical && finite) ? 1u : 0u;
    }
    fa.lights_plain = 1u;
    std::vector<LightK> lightk(sd->n_lights); // the same lights as the lean path reads them (rt_scene_dev.h)
    for (uint32_t i = 0; i < sd->n_lights; i++) {
        const DevLight &l = lights[i];
        LightK &k = lightk[i];
        std::memset(&k, 0, sizeof(k));
        for (int c = 0; c < 3; c++) { k.p[c] = l.p[c]; k.sdir[c] = l.sdir[c]; k.color[c] = l.color[c]; }
        k.u2 = l.u2; k.inv_uu = l.inv_uu; k.len_u = l.len_u;
        k.four_u2 = 4.0 * l.u2;
        k.s_yz = std::fabs(l.sdir[1]) + std::fabs(l.sdir[2]);
        k.s_xz = std::fabs(l.sdir[0]) + std::fabs(l.sdir[2]);
        k.s_xy = std::fabs(l.sdir[0]) + std::fabs(l.sdir[1]);
        k.flags = (l.spherical ? 1u : 0u) | (l.backface_exact ? 2u : 0u) | (std::fabs(l.u2) > 1e-7 ? 4u : 0u); // EPS of include/surface_impl.h:16,138
        if (!l.spherical && (k.flags & 6u) != 6u) fa.lights_plain = 0u;
    }

    // the wave-per-block instantiation: unit spheres only, every one with a bounding radius, no mirror (sparse frames take the other one)
    ctx->lean_ok = !(cfg.flags & (RT_FLAG_SIMPLE | RT_FLAG_NOLEAN)) && fa.all_cullable && fa.n_us == sd->n_objects && !fa.has_mirror && fa.n_gq == 0 && fa.n_lin == 0 &&
                   fa.n_cub == 0;
    if (sd->n_lights > 64u) ctx->lean_ok = false; // (its point-light pass keeps one bit per light and lane)
    fa.pt_mask[0] = fa.pt_mask[1] = 0u;
    for (uint32_t i = 0; i < sd->n_lights && i < 64u; i++)
        if (sd->light_is_spherical[i]) fa.pt_mask[i >> 5] |= 1u << (i & 31u);
    if (std::getenv("MI355RT_NOLEAN")) ctx->lean_ok = false; // (experiments)
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device) == hipSuccess && cus > 0) ctx->wg_slots = 6u * (uint32_t) cus;
        if (const char *e = std::getenv("MI355RT_LEAN")) ctx->lean_force = (e[0] == 'a') ? 1 : ((e[0] == 'n') ? -1 : 0);
    }

    if (!(cfg.flags & RT_FLAG_SIMPLE) && rt_wavefront_lds_bytes_strict(fa.stage_bytes, sd->n_lights, (int) fa.has_mirror, fa.cull ? fa.n_us : 0u, 0, fa.n_cub) > 160u * 1024u) {
        return fail(RT_ERR_SCENE, "rt_create: scene needs %zu bytes of LDS per workgroup (limit 160 KiB)",
                    rt_wavefront_lds_bytes_strict(fa.stage_bytes, sd->n_lights, (int) fa.has_mirror, fa.cull ? fa.n_us : 0u, 0, fa.n_cub));
    }
    int rc = RT_OK;
    auto hip_ok = [&](hipError_t err, const char *what) {
        if (err != hipSuccess && rc == RT_OK) rc = fail(RT_ERR_DEVICE, "%s failed: %s", what, hipGetErrorString(err));
        return err == hipSuccess;
    };
    const size_t fb_bytes = (size_t) (ctx->local_rows ? ctx->local_rows : 1) * sd->width * ctx->pixel_bytes;
    hip_ok(hipMalloc((void **) &ctx->d_obj, blob.size()), "hipMalloc(scene)") &&
        hip_ok(hipMalloc((void **) &ctx->d_light, (sizeof(DevLight) + sizeof(LightK)) * (lights.size() ? lights.size() : 1)), "hipMalloc(lights)") &&
        hip_ok(hipMalloc(&ctx->d_fb, fb_bytes), "hipMalloc(framebuffer)") &&
        hip_ok(hipMalloc((void **) &ctx->d_counters, sizeof(unsigned long long) * 64), "hipMalloc(counters)") &&
        hip_ok(hipMemset(ctx->d_counters, 0, sizeof(unsigned long long) * 64), "hipMemset(counters)") &&
        hip_ok(hipMemcpy(ctx->d_obj, blob.data(), blob.size(), hipMemcpyHostToDevice), "hipMemcpy(scene)") &&
        hip_ok(lights.empty() ? hipSuccess : hipMemcpy(ctx->d_light, lights.data(), sizeof(DevLight) * lights.size(), hipMemcpyHostToDevice), "hipMemcpy(lights)") &&
        hip_ok(lights.empty() ? hipSuccess : hipMemcpy(ctx->d_light + lights.size(), lightk.data(), sizeof(LightK) * lightk.size(), hipMemcpyHostToDevice), "hipMemcpy(light table)") &&
        hip_ok(hipEventCreate(&ctx->ev0), "hipEventCreate") && hip_ok(hipEventCreate(&ctx->ev1), "hipEventCreate") &&
        hip_ok(hipEventCreateWithFlags(&ctx->ev_done, hipEventDisableTiming), "hipEventCreate");
    if (rc != RT_OK) return rc;
    {
        // camera-plane coordinates of every pixel column / row: render_pixel's camera_x / camera_y
        // (src/update-cpu.cpp:84-87) depend only on the pixel index and the scene, so they are evaluated here once,
        // with the same IEEE operations in the same order (this file is compiled with -ffp-contract=off)
        std::vector<double> cx(sd->width), cy(sd->height);
        for (uint32_t x = 0; x < sd->width; x++) {
            const double ndc_x = ((int) x + 0.5) / (int) sd->width;
            cx[x] = (2.0 * ndc_x - 1.0) * fa.aspect * fa.tan_half_fov;
        }
        for (uint32_t y = 0; y < sd->height; y++) {
            const double ndc_y = ((int) y + 0.5) / (int) sd->height;
            cy[y] = (2.0 * ndc_y - 1.0) * fa.tan_half_fov;
        }
        hip_ok(hipMalloc((void **) &ctx->d_camx, sizeof(double) * sd->width), "hipMalloc(camx)") &&
            hip_ok(hipMalloc((void **) &ctx->d_camy, sizeof(double) * sd->height), "hipMalloc(camy)") &&
            hip_ok(hipMemcpy(ctx->d_camx, cx.data(), sizeof(double) * sd->width, hipMemcpyHostToDevice), "hipMemcpy(camx)") &&
            hip_ok(hipMemcpy(ctx->d_camy, cy.data(), sizeof(double) * sd->height, hipMemcpyHostToDevice), "hipMemcpy(camy)");
        if (rc == RT_OK && !(cfg.flags & (RT_FLAG_SIMPLE | RT_FLAG_STATIC_ORDER)) && fa.n_tiles > 0 && fa.n_tiles <= RT_ORD_MAX_TILES) {
            // launch-order feedback: three generations, all empty (first frame = index order)
            fa.ord_stride = (RT_ORD_HDR + 17u * fa.n_tiles + 15u) & ~15u;
            // ... and behind them one word per tile, the last frame in which one of a split tile's two workgroups entered the tile (FrameArgs::ord_frame)
            const size_t bytes = sizeof(uint32_t) * (3u * (size_t) fa.ord_stride + fa.n_tiles);
            hip_ok(hipMalloc((void **) &fa.order_state, bytes), "hipMalloc(order)") && hip_ok(hipMemset(fa.order_state, 0, bytes), "hipMemset(order)");
            // the kernel reports the number of listed tiles through one host-mapped word; without it (allocation
            // refused) every launch simply carries n_tiles list slots
            if (rc == RT_OK && hipHostMalloc((void **) &ctx->h_listed, 64, hipHostMallocMapped | hipHostMallocPortable) == hipSuccess) {
                ctx->h_listed[0] = 0;
                ctx->h_listed[1] = 0;
                ctx->h_listed[2] = 0xFFFFFFFFu; // (nothing known yet: the wave-per-block instantiation starts)
                if (hipHostGetDevicePointer((void **) &fa.ord_host, ctx->h_listed, 0) != hipSuccess) {
                    (void) hipHostFree(ctx->h_listed);
                    ctx->h_listed = nullptr;
                    fa.ord_host = nullptr;
                }
            } else {
                ctx->h_listed = nullptr;
                (void) hipGetLastError();
            }
        }
        if (rc == RT_OK && !(cfg.flags & RT_FLAG_SIMPLE) && fa.all_cullable && fa.n_tiles > 0) {
            // one word per tile for the scan workgroups (rt_wavefront.hip, scan_tiles); all zero = "no frame has classified it"
            hip_ok(hipMalloc((void **) &fa.tile_state, sizeof(uint32_t) * fa.n_tiles), "hipMalloc(tile state)") &&
                hip_ok(hipMemset(fa.tile_state, 0, sizeof(uint32_t) * fa.n_tiles), "hipMemset(tile state)");
        }
        if (rc != RT_OK) return rc;
    }
    ctx->zero_counters = std::getenv("MI355RT_DEBUG_COUNTERS") != nullptr;
    if (ctx->zero_counters) { // room for the stamp rows of a diagnostic (STAMPS=1) build: one per wave
        ctx->n_stamp_rows = (size_t) fa.n_tiles * 9 + 96; // one row per wave of the (up to) 2 * n_tiles + n_tiles / 16 + n_tiles / 64 + 2 workgroups of a launch
        if (hipMalloc((void **) &ctx->d_stamps, ctx->n_stamp_rows * 16 * sizeof(uint64_t) + 8) != hipSuccess) ctx->d_stamps = nullptr;
        else (void) hipMemset(ctx->d_stamps, 0, ctx->n_stamp_rows * 16 * sizeof(uint64_t));
    }
    guard.p = nullptr;
    *out = ctx;
    return RT_OK;
}

static int render_impl(rt_ctx *ctx, const double cam[16], void *dev_fb, void *stream_, float *ms, bool sparse, uint32_t sparse_cap)
{
    hipStream_t stream = (hipStream_t) stream_;
    FrameArgs &fa = ctx->fa;
    fa.sparse = sparse ? 1u : 0u;
    fa.sparse_cap = sparse ? sparse_cap : 0u;
    // Which instantiation renders a scene of unit spheres?  The wave-per-block one ("lean") executes a quarter fewer instructions per
    // frame and wins wherever the GPU is full (4K 120 -> 91 us, 8K 425 -> 306, the 1080p start pose 44 -> 38).  The general one splits
    // costly tiles over two workgroups and every tile's lights over its four waves, which is what counts while few tiles have hits and
    // the frame ends with its slowest wave (orbit poses 5 / 6 at 1080p: 40 us against 54).  The previous frames' count of tiles with
    // hits decides, with a hysteresis: lean from 0.66 of the workgroup slots up, back below 0.62 (the orbit of tools/flythrough_bench.py: the
    // general one wins every pose below 950 tiles for 1536 slots and loses every pose above 975; a wider band kept poses 3 and 4 on the wrong side).
    if (ctx->lean_ok && ctx->h_listed && ctx->lean_force == 0) {
        const uint32_t tiles = ((volatile uint32_t *) ctx->h_listed)[2]; // tiles with hits a few frames ago (listed, or the census' estimate while the lists are off)
        if (ctx->lean_now ? (uint64_t) tiles * 100u < (uint64_t) ctx->wg_slots * 62u : (uint64_t) tiles * 100u >= (uint64_t) ctx->wg_slots * 66u) ctx->lean_now = !ctx->lean_now;
    } else {
        ctx->lean_now = ctx->lean_force >= 0;
    }
    static const bool debug_order = std::getenv("MI355RT_DEBUG_ORDER") != nullptr; // (diagnostics: what the previous frames' kernels reported back)
    if (debug_order && ctx->h_listed)
        std::fprintf(stderr, "mi355rt: frame %llu: tiles with hits %u, list slots wanted %u, census %u, schedule %s\n", (unsigned long long) ctx->frame, ((volatile uint32_t *) ctx->h_listed)[2],
                     ((volatile uint32_t *) ctx->h_listed)[0], ((volatile uint32_t *) ctx->h_listed)[1], ctx->lean_now ? "lean" : "general");
    fa.lean = (ctx->lean_ok && ctx->lean_now && !sparse) ? 1u : 0u;
    std::memcpy(fa.cam, cam, sizeof(double) * 16);
    // g_ray_origin = camera_matrix * (0,0,0,1), src/update-cpu.cpp:123 -- glm order (m0*x + m1*y) + (m2*z + m3*w)
    for (int r = 0; r < 3; r++) fa.origin[r] = (cam[0 + r] * 0.0 + cam[4 + r] * 0.0) + (cam[8 + r] * 0.0 + cam[12 + r] * 1.0);

    for (size_t j = 0; j * RT_NCOEF < ctx->cub_coefs.size(); j++) { // degree-3 objects: F, grad F, half Hessian at the frame's ray origin (rt_math.hpp, cubic_at)
        const rtm::CubicAt a = rtm::cubic_at(ctx->cub_coefs.data() + j * RT_NCOEF, rtm::D3{fa.origin[0], fa.origin[1], fa.origin[2]});
        const rtm::CubicAbs ab = rtm::cubic_abs(ctx->cub_coefs.data() + j * RT_NCOEF); // what cubic_guarded's error bounds follow from (same function as on the device)
        const double v[RT_CUB_REC] = {a.f, a.gx, a.gy, a.gz, a.hxx, a.hyy, a.hzz, a.hxy, a.hxz, a.hyz};
        const double va[4] = {ab.a3, ab.a2, ab.a1, ab.a0};
        std::memcpy(fa.cub_rec[j], v, sizeof(v));
        std::memcpy(fa.cub_abs[j], va, sizeof(va));
    }
    { // tile pyramids of the early-out test (FrameArgs::tile_nt): inverse transpose of the camera's 3x3 part
        const double a = cam[0], b = cam[4], c = cam[8], d = cam[1], e = cam[5], f = cam[9], g = cam[2], h = cam[6], i = cam[10];
        const double co00 = e * i - f * h, co01 = -(d * i - f * g), co02 = d * h - e * g;
        const double co10 = -(b * i - c * h), co11 = a * i - c * g, co12 = -(a * h - b * g);
        const double co20 = b * f - c * e, co21 = -(a * f - c * d), co22 = a * e - b * d;
        const double det = a * co00 + b * co01 + c * co02;
        const double amax = std::fabs(a) + std::fabs(b) + std::fabs(c) + std::fabs(d) + std::fabs(e) + std::fabs(f) + std::fabs(g) + std::fabs(h) + std::fabs(i);
        fa.tile_planes_ok = (std::isfinite(det) && std::isfinite(amax) && std::fabs(det) > 1e-9 * amax * amax * amax) ? 1u : 0u;
        if (fa.tile_planes_ok) { // (M^-1)^T = cofactor matrix / det; stored column-major: element (row r, col k) at [3 * k + r]
            const double inv = 1.0 / det;
            const double nt[9] = {co00 * inv, co10 * inv, co20 * inv, co01 * inv, co11 * inv, co21 * inv, co02 * inv, co12 * inv, co22 * inv};
            for (int k = 0; k < 9; k++) fa.tile_nt[k] = nt[k];
        }
        fa.cx_a = 2.0 * fa.aspect * fa.tan_half_fov / (double) fa.width;
        fa.cx_b = (1.0 / (double) fa.width - 1.0) * fa.aspect * fa.tan_half_fov;
        fa.cy_a = 2.0 * fa.tan_half_fov / (double) fa.height;
        fa.cy_b = (1.0 / (double) fa.height - 1.0) * fa.tan_half_fov;
        if (!(fa.cx_a > 0.0) || !(fa.cy_a > 0.0) || !std::isfinite(fa.cx_a) || !std::isfinite(fa.cy_a)) fa.tile_planes_ok = 0;
    }
    int cur = -1;
    RT_HIP(hipGetDevice(&cur));
    if (cur != ctx->device) RT_HIP(hipSetDevice(ctx->device));

    // A context's frames depend on each other on the device (launch-order generations: read k, append k + 1, clear k + 2; tile words
    // tagged per frame), so they must run in the order they were issued.  On one stream they do; when the caller switches streams,
    // the new stream first waits for the previous frame.
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (stream) RT_HIP(hipStreamIsCapturing(stream, &cap));
    const bool capturing = cap != hipStreamCaptureStatusNone;
    if (ctx->rendered && stream != ctx->last_stream) {
        if (ctx->captured || capturing)
            return fail(RT_ERR_INVALID, "rt_render: a context whose frames were captured into a graph on one stream must stay on that stream (frames of a context are "
                                        "ordered on the device, and a capture cannot be ordered against another stream through an event)");
        RT_HIP(hipStreamWaitEvent(stream, ctx->ev_done, 0));
    }
    void *fb = dev_fb ? dev_fb : ctx->d_fb;
    if (sparse) RT_HIP(hipMemsetAsync(fb, 0, 16, stream)); // message header: count, overflow
    const int count = (ctx->cfg.flags & RT_FLAG_COUNT) ? 1 : 0;
    const int rgba8 = ctx->cfg.format == RT_FMT_RGBA8;
    if (count || ctx->zero_counters) {
        RT_HIP(hipMemsetAsync(ctx->d_counters, 0, sizeof(unsigned long long) * 28, stream)); // (word 31 holds the stamp rows' address)
        RT_HIP(hipMemsetAsync(ctx->d_counters + 32, 0, sizeof(unsigned long long) * 32, stream));
    }
    if (ctx->d_stamps) {
        RT_HIP(hipMemsetAsync(ctx->d_stamps, 0, ctx->n_stamp_rows * 16 * sizeof(uint64_t), stream)); // rows of this frame only
        const unsigned long long ptr = (unsigned long long) (uintptr_t) ctx->d_stamps;
        RT_HIP(hipMemcpyAsync(ctx->d_counters + 31, &ptr, sizeof(ptr), hipMemcpyHostToDevice, stream));
        RT_HIP(hipStreamSynchronize(stream));
    }
    fa.n_scan = 0;
    if (fa.tile_state && fa.tile_planes_ok && !(ctx->cfg.flags & RT_FLAG_NOSCAN)) {
        if (ctx->tag >= 0x1FFFFFF0u) { // the tag is stored shifted by three bits: start over with clean words (once in 2^29 frames)
            RT_HIP(hipMemsetAsync(fa.tile_state, 0, sizeof(uint32_t) * fa.n_tiles, stream));
            ctx->tag = 0;
        }
        fa.frame_tag = ++ctx->tag;
        fa.n_scan = (fa.n_tiles + RT_SCAN_TILES - 1) / RT_SCAN_TILES;
    }
    if (fa.order_state) { // rotate the launch-order generations: read k, write k+1, clear k+2
        fa.ord_read = (uint32_t) (ctx->frame % 3u);
        fa.ord_write = (uint32_t) ((ctx->frame + 1u) % 3u);
        fa.ord_zero = (uint32_t) ((ctx->frame + 2u) % 3u);
        // list slots of this launch: what an earlier frame reported (the host runs ahead of the device, so the words
        // are a few frames old) plus a quarter and 64; too few only means that the surplus tiles start in index order.
        // The ordering is switched off while the census says that >= 25 % of the tiles have hits (back on below 20 %).
        uint32_t cap = fa.n_tiles;
        if (ctx->h_listed) {
            const uint32_t seen = ((volatile uint32_t *) ctx->h_listed)[0], census = ((volatile uint32_t *) ctx->h_listed)[1];
            const uint64_t want = (uint64_t) seen + seen / 4u + 64u;
            if (want < cap) cap = (uint32_t) want;
            const uint64_t with_hits = (uint64_t) census * 16u;
            // ... and while so many tiles have hits that the launch is many rounds of workgroups deep anyway: there the order buys nothing any
            // more and the lists only cost their upkeep -- the decode in front of every list slot.  Measured with the lean schedule (index
            // order against lists): 2 rounds (4K orbit pose 5) 81 -> 69 us with the lists, 3.7 rounds (4K pose 19) 117 -> 109, 4.3 rounds (4K pose
            // 16) 119 / 120, 7.5 - 8 rounds (8K poses 5 / 6) 206 -> 243 and 197 -> 233, 12.7 rounds (8K start pose) 288 / 293.  Off from 16 / 3
            // rounds, back on below 4.
            const uint64_t slots = ctx->wg_slots ? ctx->wg_slots : 1536u;
            const bool too_many = ctx->ord_on ? with_hits * 3u >= slots * 16u : with_hits >= slots * 4u;
            const bool too_dense = ctx->ord_on ? with_hits * 4u >= fa.n_tiles : with_hits * 5u >= fa.n_tiles;
            ctx->ord_on = !(too_many || too_dense);
        }
        fa.ord_cap = cap;
        fa.ord_on = ctx->ord_on ? 1u : 0u;
        fa.ord_split = (fa.sparse || fa.lean) ? 0u : ctx->ord_split; // (a sparse message has one slot per tile; the lean instantiation's waves are independent:
                                                                     // a second workgroup per tile would shorten nothing)
        // this frame's number for the per-tile "entered by" words of split tiles: 1 .. 0xFFFFFFF0, never 0 (the words start out 0).  The
        // election is an atomicMax, so when the number starts over (every 2^32 - 16 frames) the words are cleared first -- the same
        // guard the tile-word tag has above.
        fa.ord_frame = (uint32_t) (ctx->frame % 0xFFFFFFF0ull) + 1u;
        if (fa.ord_frame == 1u && ctx->frame != 0u)
            RT_HIP(hipMemsetAsync(fa.order_state + 3u * (size_t) fa.ord_stride, 0, sizeof(uint32_t) * fa.n_tiles, stream));
        ctx->frame++;
    }
    if (ms) RT_HIP(hipEventRecord(ctx->ev0, stream));
    const bool fast = (ctx->cfg.flags & RT_FLAG_FAST) != 0;
    hipError_t e;
    if (ctx->cfg.flags & RT_FLAG_SIMPLE)
        e = fast ? rt_launch_trace_fast(&fa, ctx->d_obj, ctx->d_light, fb, ctx->d_counters, rgba8, count, stream)
                 : rt_launch_trace_strict(&fa, ctx->d_obj, ctx->d_light, fb, ctx->d_counters, rgba8, count, stream);
    else
        e = fast ? rt_launch_wavefront_fast(&fa, ctx->d_obj, ctx->d_light, fb, ctx->d_counters, count, ctx->d_camx, ctx->d_camy, stream)
                 : rt_launch_wavefront_strict(&fa, ctx->d_obj, ctx->d_light, fb, ctx->d_counters, count, ctx->d_camx, ctx->d_camy, stream);
    if (e != hipSuccess) return fail(RT_ERR_DEVICE, "kernel launch failed: %s", hipGetErrorString(e));
    ctx->counted = count != 0;
    if (!capturing) RT_HIP(hipEventRecord(ctx->ev_done, stream));
    ctx->captured = capturing;
    ctx->last_stream = stream;
    ctx->rendered = true;
    if (ms) {
        RT_HIP(hipEventRecord(ctx->ev1, stream));
        RT_HIP(hipEventSynchronize(ctx->ev1));
        RT_HIP(hipEventElapsedTime(ms, ctx->ev0, ctx->ev1));
    }
    return RT_OK;
}

extern "C" int rt_render(rt_ctx *ctx, const double cam[16], void *dev_fb, void *stream, float *ms)
{
    if (!ctx || !cam) return fail(RT_ERR_INVALID, "rt_render: null argument");
    return render_impl(ctx, cam, dev_fb, stream, ms, false, 0);
}

extern "C" int rt_render_sparse(rt_ctx *ctx, const double cam[16], void *dev_msg, uint32_t capacity_tiles, void *stream, float *ms)
{
    if (!ctx || !cam || !dev_msg) return fail(RT_ERR_INVALID, "rt_render_sparse: null argument");
    if (ctx->cfg.format != RT_FMT_RGBA8) return fail(RT_ERR_INVALID, "rt_render_sparse: the context does not render RGBA8");
    if (ctx->cfg.flags & RT_FLAG_SIMPLE) return fail(RT_ERR_INVALID, "rt_render_sparse: not available with RT_FLAG_SIMPLE");
    return render_impl(ctx, cam, dev_msg, stream, ms, true, capacity_tiles);
}

extern "C" int rt_local_rows(const rt_ctx *ctx, uint32_t *n_rows)
{
    if (!ctx || !n_rows) return fail(RT_ERR_INVALID, "rt_local_rows: null argument");
    *n_rows = ctx->local_rows;
    return RT_OK;
}

extern "C" int rt_max_local_rows(const rt_ctx *ctx, uint32_t *n_rows)
{
    if (!ctx || !n_rows) return fail(RT_ERR_INVALID, "rt_max_local_rows: null argument");
    *n_rows = ctx->max_local_rows;
    return RT_OK;
}

extern "C" int rt_row_map(const rt_ctx *ctx, uint32_t *rows)
{
    if (!ctx || !rows) return fail(RT_ERR_INVALID, "rt_row_map: null argument");
    const uint32_t B = ctx->cfg.band_rows, W = ctx->cfg.world, R = ctx->cfg.rank;
    for (uint32_t lr = 0; lr < ctx->local_rows; lr++) {
        uint32_t b = lr / B;
        rows[lr] = (b * W + R) * B + (lr - b * B);
    }
    return RT_OK;
}

extern "C" size_t rt_pixel_bytes(const rt_ctx *ctx) { return ctx ? ctx->pixel_bytes : 0; }

extern "C" void *rt_device_fb(rt_ctx *ctx) { return ctx ? ctx->d_fb : nullptr; }

extern "C" int rt_download(rt_ctx *ctx, void *host_dst, size_t bytes)
{
    if (!ctx || !host_dst) return fail(RT_ERR_INVALID, "rt_download: null argument");
    const size_t have = (size_t) ctx->local_rows * ctx->fa.width * ctx->pixel_bytes;
    if (bytes > have) return fail(RT_ERR_INVALID, "rt_download: %zu bytes requested, framebuffer holds %zu", bytes, have);
    RT_HIP(hipSetDevice(ctx->device));
    RT_HIP(hipDeviceSynchronize());
    RT_HIP(hipMemcpy(host_dst, ctx->d_fb, bytes, hipMemcpyDeviceToHost));
    return RT_OK;
}

extern "C" int rt_assemble(rt_ctx *ctx, const void *gathered, void *full, void *stream)
{
    if (!ctx || !gathered || !full) return fail(RT_ERR_INVALID, "rt_assemble: null argument");
    hipError_t e = rt_launch_assemble_strict(gathered, full, ctx->fa.width, ctx->fa.height, ctx->cfg.world, ctx->cfg.band_rows,
                                             ctx->max_local_rows, ctx->cfg.format == RT_FMT_RGBA8, (hipStream_t) stream);
    if (e != hipSuccess) return fail(RT_ERR_DEVICE, "assemble launch failed: %s", hipGetErrorString(e));
    return RT_OK;
}

// background colour as the RGBA8 kernels store it (iround(c * 255), alpha 255), little-endian r | g << 8 | b << 16 | a << 24
static uint32_t bg_rgba8(const FrameArgs &fa)
{
    const uint32_t r = (uint32_t) (unsigned char) (int) (fa.bg[0] * 255.0f + 0.5f), g = (uint32_t) (unsigned char) (int) (fa.bg[1] * 255.0f + 0.5f),
                   b = (uint32_t) (unsigned char) (int) (fa.bg[2] * 255.0f + 0.5f);
    return r | (g << 8) | (b << 16) | (255u << 24);
}

extern "C" size_t rt_sparse_bytes(uint32_t capacity_tiles)
{
    return ((size_t) ((4u + capacity_tiles + 3u) & ~3u) + (size_t) capacity_tiles * 256u) * sizeof(uint32_t);
}

extern "C" int rt_pack_sparse(rt_ctx *ctx, const void *dev_fb, void *dev_msg, uint32_t capacity_tiles, void *stream)
{
    if (!ctx || !dev_msg) return fail(RT_ERR_INVALID, "rt_pack_sparse: null argument");
    if (ctx->cfg.format != RT_FMT_RGBA8) return fail(RT_ERR_INVALID, "rt_pack_sparse: the context does not render RGBA8");
    hipError_t e = rt_launch_pack_sparse_strict(dev_fb ? dev_fb : ctx->d_fb, dev_msg, ctx->fa.width, ctx->local_rows, bg_rgba8(ctx->fa), capacity_tiles,
                                                (hipStream_t) stream);
    if (e != hipSuccess) return fail(RT_ERR_DEVICE, "pack launch failed: %s", hipGetErrorString(e));
    return RT_OK;
}

extern "C" int rt_assemble_sparse(rt_ctx *ctx, const void *gathered, uint32_t capacity_tiles, void *full, void *stream)
{
    if (!ctx || !gathered || !full) return fail(RT_ERR_INVALID, "rt_assemble_sparse: null argument");
    if (ctx->cfg.format != RT_FMT_RGBA8) return fail(RT_ERR_INVALID, "rt_assemble_sparse: the context does not render RGBA8");
    hipError_t e = rt_launch_assemble_sparse_strict(gathered, full, ctx->fa.width, ctx->fa.height, ctx->cfg.world, ctx->cfg.band_rows, bg_rgba8(ctx->fa),
                                                    capacity_tiles, nullptr, 0, 0, (hipStream_t) stream);
    if (e != hipSuccess) return fail(RT_ERR_DEVICE, "sparse assemble launch failed: %s", hipGetErrorString(e));
    return RT_OK;
}

extern "C" size_t rt_sparse_stamp_bytes(rt_ctx *ctx)
{
    if (!ctx) return 0;
    const size_t max_tiles = (size_t) ((ctx->fa.width + 15u) / 16u) * ((ctx->max_local_rows + 15u) / 16u);
    return sizeof(uint32_t) * (size_t) ctx->cfg.world * max_tiles;
}

extern "C" int rt_assemble_sparse_incremental(rt_ctx *ctx, const void *gathered, uint32_t capacity_tiles, void *full, void *stamps, uint32_t frame_tag,
                                              void *stream)
{
    if (!ctx || !gathered || !full || !stamps) return fail(RT_ERR_INVALID, "rt_assemble_sparse_incremental: null argument");
    if (ctx->cfg.format != RT_FMT_RGBA8) return fail(RT_ERR_INVALID, "rt_assemble_sparse_incremental: the context does not render RGBA8");
    const uint32_t max_tiles = ((ctx->fa.width + 15u) / 16u) * ((ctx->max_local_rows + 15u) / 16u);
    hipError_t e = rt_launch_assemble_sparse_strict(gathered, full, ctx->fa.width, ctx->fa.height, ctx->cfg.world, ctx->cfg.band_rows, bg_rgba8(ctx->fa),
                                                    capacity_tiles, stamps, max_tiles, frame_tag, (hipStream_t) stream);
    if (e != hipSuccess) return fail(RT_ERR_DEVICE, "sparse assemble launch failed: %s", hipGetErrorString(e));
    return RT_OK;
}

extern "C" int rt_get_counters(rt_ctx *ctx, rt_counters *out)
{
    if (!ctx || !out) return fail(RT_ERR_INVALID, "rt_get_counters: null argument");
    if (!ctx->counted) return fail(RT_ERR_INVALID, "rt_get_counters: the last render was not done with RT_FLAG_COUNT");
    unsigned long long h[8];
    RT_HIP(hipSetDevice(ctx->device));
    RT_HIP(hipDeviceSynchronize());
    RT_HIP(hipMemcpy(h, ctx->d_counters, sizeof(h), hipMemcpyDeviceToHost));
    out->primary_rays = h[0];
    out->shadow_rays = h[1];
    out->reflect_rays = h[2];
    out->tests = h[3];
    out->hits = h[4];
    out->solves = h[5];
    out->tests_executed = h[6];
    out->cull_evals = h[7];
    return RT_OK;
}

extern "C" int rt_get_counters_detail(rt_ctx *ctx, rt_counters_detail *out)
{
    if (!ctx || !out) return fail(RT_ERR_INVALID, "rt_get_counters_detail: null argument");
    if (!ctx->counted) return fail(RT_ERR_INVALID, "rt_get_counters_detail: the last render was not done with RT_FLAG_COUNT");
    if (ctx->cfg.flags & RT_FLAG_SIMPLE) return fail(RT_ERR_INVALID, "rt_get_counters_detail: the simple kernel does not split its counters");
    unsigned long long h[21];
    RT_HIP(hipSetDevice(ctx->device));
    RT_HIP(hipDeviceSynchronize());
    RT_HIP(hipMemcpy(h, ctx->d_counters + 32, sizeof(h), hipMemcpyDeviceToHost));
    for (int i = 0; i < 4; i++) out->tests_executed[i] = h[i];
    for (int i = 0; i < 3; i++) out->solves[i] = h[4 + i];
    for (int i = 0; i < 5; i++) out->cull_evals[i] = h[7 + i];
    for (int i = 0; i < 4; i++) out->cubic_branch[i] = h[12 + i];
    out->shadow_rays_traced = h[16];
    out->hit_lights_shaded = h[17];
    out->primary_rays_formed = h[18];
    out->cubic_points = h[19];
    out->cubic_refused = h[20];
    return RT_OK;
}

extern "C" int rt_debug_counters(rt_ctx *ctx, uint64_t out[32])
{
    if (!ctx || !out) return fail(RT_ERR_INVALID, "rt_debug_counters: null argument");
    RT_HIP(hipSetDevice(ctx->device));
    RT_HIP(hipDeviceSynchronize());
    RT_HIP(hipMemcpy(out, ctx->d_counters, sizeof(uint64_t) * 32, hipMemcpyDeviceToHost));
    if (ctx->d_stamps) { // diagnostic build: sum the per-wave stamp rows into words 8..19
        std::vector<uint64_t> rows(ctx->n_stamp_rows * 16);
        RT_HIP(hipMemcpy(rows.data(), ctx->d_stamps, rows.size() * sizeof(uint64_t), hipMemcpyDeviceToHost));
        for (int i = 0; i < 12; i++) out[8 + i] = 0;
        for (size_t r = 0; r < ctx->n_stamp_rows; r++)
            for (int i = 0; i < 12; i++) out[8 + i] += rows[r * 16 + i];
    }
    return RT_OK;
}

extern "C" int rt_debug_stamp_rows(rt_ctx *ctx, uint64_t *out, size_t max_rows, size_t *n_rows)
{
    if (!ctx || !n_rows) return fail(RT_ERR_INVALID, "rt_debug_stamp_rows: null argument");
    *n_rows = ctx->d_stamps ? ctx->n_stamp_rows : 0;
    if (!out || !ctx->d_stamps) return RT_OK;
    const size_t n = max_rows < ctx->n_stamp_rows ? max_rows : ctx->n_stamp_rows;
    RT_HIP(hipSetDevice(ctx->device));
    RT_HIP(hipDeviceSynchronize());
    RT_HIP(hipMemcpy(out, ctx->d_stamps, n * 16 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return RT_OK;
}

extern "C" int rt_destroy(rt_ctx *ctx)
{
    if (!ctx) return RT_OK;
    (void) hipSetDevice(ctx->device);
    if (ctx->d_obj) (void) hipFree(ctx->d_obj);
    if (ctx->d_light) (void) hipFree(ctx->d_light);
    if (ctx->d_fb) (void) hipFree(ctx->d_fb);
    if (ctx->d_counters) (void) hipFree(ctx->d_counters);
    if (ctx->d_stamps) (void) hipFree(ctx->d_stamps);
    if (ctx->d_camx) (void) hipFree(ctx->d_camx);
    if (ctx->d_camy) (void) hipFree(ctx->d_camy);
    if (ctx->fa.order_state) (void) hipFree(ctx->fa.order_state);
    if (ctx->fa.tile_state) (void) hipFree(ctx->fa.tile_state);
    if (ctx->h_listed) (void) hipHostFree(ctx->h_listed);
    if (ctx->ev0) (void) hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void) hipEventDestroy(ctx->ev1);
    if (ctx->ev_done) (void) hipEventDestroy(ctx->ev_done);

    delete ctx;
    return RT_OK;
}
