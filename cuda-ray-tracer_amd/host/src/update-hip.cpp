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

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "mi355rt.h"

namespace {

rt_ctx *g_ctx = nullptr;
unsigned int g_texture = 0;
unsigned int g_width = 0, g_height = 0;
// Optional presentation hook: an interactive host that wants the frame in its GL texture installs a
// function that receives the RGBA32F rows (bottom row first).  Headless use leaves it null and reads the
// device buffer through rt_device_fb().
void (*g_present)(unsigned int texture, unsigned int width, unsigned int height, const float *rgba) = nullptr;
std::vector<float> g_staging;

[[noreturn]] void die(const char *what)
{
    std::fprintf(stderr, "mi355rt: %s: %s\n", what, rt_last_error());
    std::exit(EXIT_FAILURE);
}

} // namespace

// Exported so a host can install / query without new headers.
extern "C" void mi355rt_set_presenter(void (*fn)(unsigned int, unsigned int, unsigned int, const float *)) { g_present = fn; }
extern "C" rt_ctx *mi355rt_update_context(void) { return g_ctx; }

void init_update(unsigned int texture, const Scene &scene)
{
    if (g_ctx) cleanup_update();
    g_texture = texture;
    g_width = scene.px_width;
    g_height = scene.px_height;

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

    rt_config cfg{};
    cfg.device = -1;
    cfg.world = 1;
    cfg.flags = RT_FLAG_STRICT;
    cfg.format = RT_FMT_RGBA32F;
    if (rt_create(&g_ctx, &sd, &cfg) != RT_OK) die("init_update");
}

float update(const glm::dmat4 &camera_matrix)
{
    if (!g_ctx) {
        std::fprintf(stderr, "mi355rt: update() called before init_update()\n");
        std::exit(EXIT_FAILURE);
    }
    double cam[16];
    for (int c = 0; c < 4; c++)
        for (int r = 0; r < 4; r++) cam[c * 4 + r] = camera_matrix[c][r];
    float ms = 0.0f;
    if (rt_render(g_ctx, cam, nullptr, nullptr, &ms) != RT_OK) die("update");
    if (g_present) {
        g_staging.resize((size_t) g_width * g_height * 4);
        if (rt_download(g_ctx, g_staging.data(), g_staging.size() * sizeof(float)) != RT_OK) die("update (download)");
        g_present(g_texture, g_width, g_height, g_staging.data());
    }
    return ms; // device time of the render kernel, like src/update-cuda.cu:187-189
}

void cleanup_update()
{
    if (g_ctx) rt_destroy(g_ctx);
    g_ctx = nullptr;
}
