// scene.h -- scene model handed to the render back end.
//
// API mirror of the reference's include/scene.h:8-36: Object and Scene with the same public members,
// constructors and Scene::load_from_file.  The loader behind load_from_file is this repo's own
// YAML-subset parser (../src/scene.cpp, ../src/yaml_subset.*); yaml-cpp is not needed.
#pragma once

#include <vector>

#include "light.h"
#include "surface.h"

struct Object {
    SurfaceCoefs surface;
    float reflection_ratio; // 0 = matte; > 1e-7 makes the object a (partial) mirror
    glm::vec3 color;        // albedo, channels in [0, 1]

    // validates reflection_ratio >= 0 and the colour (reference src/scene.cpp:9-14)
    Object(SurfaceCoefs surface, float reflection_ratio, const glm::vec3 &color);
};

struct Scene {
    unsigned int px_width, px_height; // render resolution (independent of any window)
    double vertical_fov;              // RADIANS (the constructor converts from degrees)
    glm::vec3 bg_color;
    unsigned int max_reflections;

    std::vector<Object> objects;
    std::vector<LightSource> lights;

    Scene() = default;
    // reference src/scene.cpp:16-22
    Scene(unsigned int px_width, unsigned int px_height, double vertical_fov_deg, unsigned int max_reflections,
          const glm::vec3 &bg_color = glm::vec3(0.0f));

    double aspect_ratio() const { return (double) px_width / px_height; }

    // Parses a YAML scene description; throws SceneException (reference src/scene.cpp:154-203).
    static Scene load_from_file(const char *path);
};
