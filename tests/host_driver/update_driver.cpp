// update_driver.cpp -- a minimal headless "host" written only against the reference's interface
// (update.h / scene.h, reference src/ray-tracer.cpp:152,215,226,245 call sequence), linked against
// libmi355rt_update.so.  Used by tests/test_gpu_parity.py to prove the drop-in boundary end to end:
//   update_driver <scene.yml> <width> <height> <max_reflections|-1> <out.f32> [16 camera doubles]
// writes width*height*4 floats (RGBA32F, bottom row first) and prints the ms that update() returned.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "mi355rt.h"
#include "scene-exception.h"
#include "update.h"

extern "C" rt_ctx *mi355rt_update_context(void);

int main(int argc, char **argv)
{
    if (argc < 6) {
        std::fprintf(stderr, "usage: %s scene.yml W H max_refl out.f32 [cam x16]\n", argv[0]);
        return 2;
    }
    Scene scene;
    try {
        scene = Scene::load_from_file(argv[1]);
    } catch (const SceneException &e) {
        std::fprintf(stderr, "%s\n", e.what());
        return 1;
    }
    scene.px_width = (unsigned) std::atoi(argv[2]);
    scene.px_height = (unsigned) std::atoi(argv[3]);
    if (std::atoi(argv[4]) >= 0) scene.max_reflections = (unsigned) std::atoi(argv[4]);
    glm::dmat4 cam(1.0);
    if (argc >= 6 + 16)
        for (int c = 0; c < 4; c++)
            for (int r = 0; r < 4; r++) cam[c][r] = std::atof(argv[6 + c * 4 + r]);

    init_update(0, scene);
    float ms = update(cam);
    std::vector<float> px((size_t) scene.px_width * scene.px_height * 4);
    if (rt_download(mi355rt_update_context(), px.data(), px.size() * sizeof(float)) != RT_OK) {
        std::fprintf(stderr, "download failed: %s\n", rt_last_error());
        return 1;
    }
    cleanup_update();
    FILE *f = std::fopen(argv[5], "wb");
    if (!f) return 1;
    std::fwrite(px.data(), sizeof(float), px.size(), f);
    std::fclose(f);
    std::printf("%f\n", ms);
    return 0;
}
