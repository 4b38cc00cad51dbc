// update_driver.cpp -- a minimal headless "host" written only against the reference's interface
// (update.h / scene.h, reference src/ray-tracer.cpp:152,215,226,245 call sequence), linked against
// libmi355rt_update.so.  Used by the GPU tests to prove the drop-in boundary end to end:
//   update_driver <scene.yml> <width> <height> <max_reflections|-1> <out> [16 camera doubles] [--present] [--ppm file.ppm] [--frames n]
// writes the frame (bottom row first) to <out>: width*height*4 floats (RGBA32F) or, with MI355RT_FORMAT=rgba8 in the
// environment, width*height*4 bytes; prints the ms that update() returned.
//   --present   install a presenter (the hook an interactive host uses to get the frame into its GL texture,
//               src/ray-tracer.cpp:209-233) and write what IT received to <out>.present
//   --ppm       also write a binary P6 image, top row first (from RGBA8 as is; from floats by iround(c*255), the
//               quantisation of src/update-cuda.cu:149-156)
// MI355RT_DEVICES / MI355RT_PARTS / MI355RT_FORMAT select GPUs and format (host/src/update-hip.cpp).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "mi355rt.h"
#include "scene-exception.h"
#include "update.h"

extern "C" int mi355rt_update_download(void *host_dst, size_t bytes);
extern "C" unsigned int mi355rt_update_format(void);
extern "C" void mi355rt_set_presenter(void (*fn)(unsigned int, unsigned int, unsigned int, const float *));
extern "C" void mi355rt_set_presenter_rgba8(void (*fn)(unsigned int, unsigned int, unsigned int, const unsigned char *));

static std::vector<unsigned char> g_presented;
static unsigned g_presented_texture = 0, g_present_calls = 0;

static void present_f32(unsigned texture, unsigned w, unsigned h, const float *rgba)
{
    g_presented.assign((const unsigned char *) rgba, (const unsigned char *) rgba + (size_t) w * h * 16);
    g_presented_texture = texture;
    g_present_calls++;
}
static void present_u8(unsigned texture, unsigned w, unsigned h, const unsigned char *rgba)
{
    g_presented.assign(rgba, rgba + (size_t) w * h * 4);
    g_presented_texture = texture;
    g_present_calls++;
}

int main(int argc, char **argv)
{
    if (argc < 6) {
        std::fprintf(stderr, "usage: %s scene.yml W H max_refl out [cam x16] [--present] [--ppm file] [--frames n]\n", argv[0]);
        return 2;
    }
    bool present = false;
    const char *ppm = nullptr;
    int frames = 1;
    std::vector<const char *> pos;
    for (int i = 1; i < argc; i++) {
        if (!std::strcmp(argv[i], "--present")) present = true;
        else if (!std::strcmp(argv[i], "--ppm") && i + 1 < argc) ppm = argv[++i];
        else if (!std::strcmp(argv[i], "--frames") && i + 1 < argc) frames = std::atoi(argv[++i]);
        else pos.push_back(argv[i]);
    }
    if (pos.size() < 5) return 2;
    Scene scene;
    try {
        scene = Scene::load_from_file(pos[0]);
    } catch (const SceneException &e) {
        std::fprintf(stderr, "%s\n", e.what());
        return 1;
    }
    scene.px_width = (unsigned) std::atoi(pos[1]);
    scene.px_height = (unsigned) std::atoi(pos[2]);
    if (std::atoi(pos[3]) >= 0) scene.max_reflections = (unsigned) std::atoi(pos[3]);
    glm::dmat4 cam(1.0);
    if (pos.size() >= 5 + 16)
        for (int c = 0; c < 4; c++)
            for (int r = 0; r < 4; r++) cam[c][r] = std::atof(pos[5 + c * 4 + r]);

    if (present) {
        mi355rt_set_presenter(present_f32);
        mi355rt_set_presenter_rgba8(present_u8);
    }
    init_update(42, scene);
    float ms = 0.0f;
    for (int i = 0; i < frames; i++) ms = update(cam);
    const bool u8 = mi355rt_update_format() == RT_FMT_RGBA8;
    const size_t W = scene.px_width, H = scene.px_height;
    std::vector<unsigned char> px(W * H * (u8 ? 4 : 16));
    if (mi355rt_update_download(px.data(), px.size()) != RT_OK) {
        std::fprintf(stderr, "download failed: %s\n", rt_last_error());
        return 1;
    }
    cleanup_update();
    FILE *f = std::fopen(pos[4], "wb");
    if (!f) return 1;
    std::fwrite(px.data(), 1, px.size(), f);
    std::fclose(f);
    if (present) {
        if (g_present_calls != (unsigned) frames || g_presented_texture != 42) {
            std::fprintf(stderr, "presenter called %u times with texture %u\n", g_present_calls, g_presented_texture);
            return 1;
        }
        const std::string pf = std::string(pos[4]) + ".present";
        f = std::fopen(pf.c_str(), "wb");
        if (!f) return 1;
        std::fwrite(g_presented.data(), 1, g_presented.size(), f);
        std::fclose(f);
    }
    if (ppm) {
        f = std::fopen(ppm, "wb");
        if (!f) return 1;
        std::fprintf(f, "P6\n%zu %zu\n255\n", W, H);
        std::vector<unsigned char> row(W * 3);
        for (size_t y = 0; y < H; y++) { // PPM is top row first; the frame is bottom row first (row 0 = bottom, src/update-cpu.cpp:125-131)
            const size_t src = H - 1 - y;
            for (size_t x = 0; x < W; x++)
                for (int k = 0; k < 3; k++) {
                    if (u8) row[3 * x + k] = px[(src * W + x) * 4 + k];
                    else row[3 * x + k] = (unsigned char) (int) (reinterpret_cast<const float *>(px.data())[(src * W + x) * 4 + k] * 255.0f + 0.5f);
                }
            std::fwrite(row.data(), 1, row.size(), f);
        }
        std::fclose(f);
    }
    std::printf("%f\n", ms);
    return 0;
}
