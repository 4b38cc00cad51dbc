// update-hip.cpp -- the MI355X back end behind the reference's update.h contract.
//
// Defines the three symbols a render back end must provide (reference include/update.h:6-8) by
// adapting them onto the C ABI of libmi355rt.so (include/mi355rt.h).  It takes the place of
// src/update-cuda.cu in the reference's link line (src/CMakeLists.txt:22-23); INTEGRATION.md shows the
// build change.  Like the reference back ends it keeps its state in file scope: the host calls
// init_update once, update once per frame and cleanup_update once (src/ray-tracer.cpp:215,226,245).
//
// Errors: the reference's CUDA back end prints the failure and exits (include/helper_cuda_opengl.h:13-24);
// this one does the same through die().
#include "update.h"

#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "mi355rt.h"

// Environment (read by init_update; the reference's back ends take their configuration at build time):
//   MI355RT_DEVICES=0,1,...   render on these GPUs (rows band-cyclic, RCCL gather to the first: libmi355rt_multi.so,
//                             loaded on demand so that a one-GPU host never needs librccl); default: the current device
//   MI355RT_PARTS=n           contexts per device in that mode (overlap of transfer and rendering), default 2
//   MI355RT_BAND_ROWS=n       rows per band in that mode, default 16
//   MI355RT_FORMAT=rgba8      framebuffer = iround(c*255) RGBA8, the format the reference's CUDA back end writes to its display
//                             surface (src/update-cuda.cu:149-156); default rgba32f, the CPU back end's floats (src/update-cpu.cpp:128-131)
namespace {

rt_ctx *g_ctx = nullptr;
rt_multi *g_multi = nullptr;
void *g_multi_lib = nullptr;
struct MultiApi {
    int (*create)(rt_multi **, const rt_scene_desc *, const int *, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t) = nullptr;
    int (*render)(rt_multi *, const double *, void *, float *) = nullptr;
    int (*download)(rt_multi *, void *, size_t) = nullptr;
    int (*destroy)(rt_multi *) = nullptr;
} g_mapi;
unsigned int g_texture = 0;
unsigned int g_width = 0, g_height = 0;
uint32_t g_format = RT_FMT_RGBA32F;
// Optional presentation hooks: an interactive host that wants the frame in its GL texture installs a function that
// receives the rows (bottom row first) -- RGBA32F floats (what the CPU back end hands to glTexImage2D, src/update-cpu.cpp:136-137)
// or, with MI355RT_FORMAT=rgba8, RGBA8 bytes.  Headless use leaves them null and reads the device buffer.
void (*g_present)(unsigned int texture, unsigned int width, unsigned int height, const float *rgba) = nullptr;
void (*g_present8)(unsigned int texture, unsigned int width, unsigned int height, const unsigned char *rgba) = nullptr;
std::vector<unsigned char> g_staging;

[[noreturn]] void die(const char *what)
{
    std::fprintf(stderr, "mi355rt: %s: %s\n", what, rt_last_error());
    std::exit(EXIT_FAILURE);
}

[[noreturn]] void die_text(const char *what, const char *why)
{
    std::fprintf(stderr, "mi355rt: %s: %s\n", what, why);
    std::exit(EXIT_FAILURE);
}

std::vector<int> device_list()
{
    std::vector<int> out;
    const char *env = std::getenv("MI355RT_DEVICES");
    if (!env || !*env) return out;
    const char *p = env;
    while (*p) {
        char *end = nullptr;
        long v = std::strtol(p, &end, 10);
        if (end == p || v < 0) die_text("MI355RT_DEVICES", "expected a comma-separated list of device ordinals");
        out.push_back((int) v);
        p = end;
        if (*p == ',') p++;
        else if (*p) die_text("MI355RT_DEVICES", "expected a comma-separated list of device ordinals");
    }
    return out;
}

uint32_t env_u32(const char *name, uint32_t dflt)
{
    const char *e = std::getenv(name);
    return (e && *e) ? (uint32_t) std::strtoul(e, nullptr, 10) : dflt;
}

void load_multi()
{
    if (g_multi_lib) return;
    // next to this library first (the build puts both in one directory), then the loader's search path
    Dl_info info{};
    std::string path = "libmi355rt_multi.so";
    if (dladdr((void *) &load_multi, &info) && info.dli_fname) {
        std::string here = info.dli_fname;
        const size_t slash = here.rfind('/');
        if (slash != std::string::npos) path = here.substr(0, slash + 1) + "libmi355rt_multi.so";
    }
    g_multi_lib = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!g_multi_lib) g_multi_lib = dlopen("libmi355rt_multi.so", RTLD_NOW | RTLD_LOCAL);
    if (!g_multi_lib) die_text("MI355RT_DEVICES needs libmi355rt_multi.so", dlerror());
    g_mapi.create = (decltype(g_mapi.create)) dlsym(g_multi_lib, "rt_create_multi");
    g_mapi.render = (decltype(g_mapi.render)) dlsym(g_multi_lib, "rt_render_multi");
    g_mapi.download = (decltype(g_mapi.download)) dlsym(g_multi_lib, "rt_multi_download");
    g_mapi.destroy = (decltype(g_mapi.destroy)) dlsym(g_multi_lib, "rt_multi_destroy");
    if (!g_mapi.create || !g_mapi.render || !g_mapi.download || !g_mapi.destroy) die_text("libmi355rt_multi.so", "missing entry points");
}

} // namespace

// Exported so a host can install / query without new headers.
extern "C" void mi355rt_set_presenter(void (*fn)(unsigned int, unsigned int, unsigned int, const float *)) { g_present = fn; }
extern "C" void mi355rt_set_presenter_rgba8(void (*fn)(unsigned int, unsigned int, unsigned int, const unsigned char *)) { g_present8 = fn; }
extern "C" rt_ctx *mi355rt_update_context(void) { return g_ctx; }
// Blocking copy of the last frame ([height][width] pixels of the configured format), whichever mode is active.
extern "C" int mi355rt_update_download(void *host_dst, size_t bytes)
{
    if (g_multi) return g_mapi.download(g_multi, host_dst, bytes);
    if (g_ctx) return rt_download(g_ctx, host_dst, bytes);
    return RT_ERR_INVALID;
}
extern "C" unsigned int mi355rt_update_format(void) { return g_format; }

void init_update(unsigned int texture, const Scene &scene)
{
    if (g_ctx || g_multi) cleanup_update();
    g_texture = texture;
    g_width = scene.px_width;
    g_height = scene.px_height;
    g_format = RT_FMT_RGBA32F;
    if (const char *f = std::getenv("MI355RT_FORMAT")) {
        if (!std::strcmp(f, "rgba8")) g_format = RT_FMT_RGBA8;
        else if (std::strcmp(f, "rgba32f") && *f) die_text("MI355RT_FORMAT", "expected rgba32f or rgba8");
    }

    // flatten the Scene into the ABI's descriptor (arrays are borrowed only for the call)
    const size_t no = scene.objects.size(), nl = scene.lights.size();
    std::vector<double> coefs(no * RT_NCOEF), light_p(nl * 3);
    std::vector<float> refl(no), albedo(no * 3), light_c(nl * 3);
    std::vector<uint8_t> kind(nl);
    for (size_t i = 0; i < no; i++) {
        const Object &o = scene.objects[i];
        const double *c = &o.surface.x3; // 20 packed doubles in SurfaceCoefs order (include/surface.h:12-14)
        for (int k = 0; k < RT_NCOEF; k++) coefs[i * RT_NCOEF + k] = c[k];
        refl[i] = o.reflection_ratio;
        for (int k = 0; k < 3; k++) albedo[3 * i + k] = o.color[k];
    }
    for (size_t i = 0; i < nl; i++) {
        const LightSource &l = scene.lights[i];
        kind[i] = l.is_spherical ? 1 : 0;
        for (int k = 0; k < 3; k++) {
            light_p[3 * i + k] = l.p[k];
            light_c[3 * i + k] = l.light_color[k];
        }
    }
    rt_scene_desc sd{};
    sd.width = scene.px_width;
    sd.height = scene.px_height;
    sd.vertical_fov = scene.vertical_fov;
    for (int k = 0; k < 3; k++) sd.bg_color[k] = scene.bg_color[k];
    sd.max_reflections = scene.max_reflections;
    sd.n_objects = (uint32_t) no;
    sd.n_lights = (uint32_t) nl;
    sd.coefs = coefs.data();
    sd.reflection = refl.data();
    sd.albedo = albedo.data();
    sd.light_is_spherical = kind.data();
    sd.light_p = light_p.data();
    sd.light_color = light_c.data();

    const std::vector<int> devs = device_list();
    if (devs.size() > 1 || (devs.size() == 1 && std::getenv("MI355RT_MULTI_SELF"))) {
        load_multi();
        const uint32_t flags = RT_FLAG_STRICT | (std::getenv("MI355RT_MULTI_SELF") ? RT_MULTI_SELF_EXCHANGE : 0u) |
                               (std::getenv("MI355RT_MULTI_BANDWISE") ? RT_MULTI_BANDWISE : 0u); // (rows band by band into their place in the frame: no reassembly pass)
        if (g_mapi.create(&g_multi, &sd, devs.data(), (uint32_t) devs.size(), env_u32("MI355RT_BAND_ROWS", 16), env_u32("MI355RT_PARTS", 2), flags, g_format) != RT_OK)
            die("init_update (MI355RT_DEVICES)");
        return;
    }
    rt_config cfg{};
    cfg.device = devs.empty() ? -1 : devs[0];
    cfg.world = 1;
    cfg.flags = RT_FLAG_STRICT;
    cfg.format = g_format;
    if (rt_create(&g_ctx, &sd, &cfg) != RT_OK) die("init_update");
}

float update(const glm::dmat4 &camera_matrix)
{
    if (!g_ctx && !g_multi) {
        std::fprintf(stderr, "mi355rt: update() called before init_update()\n");
        std::exit(EXIT_FAILURE);
    }
    double cam[16];
    for (int c = 0; c < 4; c++)
        for (int r = 0; r < 4; r++) cam[c * 4 + r] = camera_matrix[c][r];
    float ms = 0.0f;
    if (g_multi) {
        if (g_mapi.render(g_multi, cam, nullptr, &ms) != RT_OK) die("update");
    } else if (rt_render(g_ctx, cam, nullptr, nullptr, &ms) != RT_OK) {
        die("update");
    }
    if ((g_format == RT_FMT_RGBA8 && g_present8) || (g_format == RT_FMT_RGBA32F && g_present)) {
        g_staging.resize((size_t) g_width * g_height * (g_format == RT_FMT_RGBA8 ? 4 : 16));
        if (mi355rt_update_download(g_staging.data(), g_staging.size()) != RT_OK) die("update (download)");
        if (g_format == RT_FMT_RGBA8) g_present8(g_texture, g_width, g_height, g_staging.data());
        else g_present(g_texture, g_width, g_height, reinterpret_cast<const float *>(g_staging.data()));
    }
    return ms; // device time of the frame, like src/update-cuda.cu:187-189 (several GPUs: transfers and reassembly included)
}

void cleanup_update()
{
    if (g_ctx) rt_destroy(g_ctx);
    g_ctx = nullptr;
    if (g_multi) g_mapi.destroy(g_multi);
    g_multi = nullptr;
}
