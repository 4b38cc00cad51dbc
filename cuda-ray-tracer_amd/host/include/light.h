// light.h -- light sources of the scene model.
//
// API mirror of the reference's include/light.h:6-13 (type name, factory names, the three public
// members and their order).  Implementation: ../src/light.cpp.
#pragma once

#include <glm/glm.hpp>

struct LightSource {
    // Distant light shining along `dir`.  Stores p = -normalize(dir) (the direction TOWARDS the light) and
    // light_color = intensity * color.                         (reference src/light.cpp:4-14)
    static LightSource directional(float intensity, const glm::dvec3 &dir, const glm::vec3 &color);
    // Point light at `pos`; inverse-square falloff is applied at shading time.
    //                                                           (reference src/light.cpp:16-26)
    static LightSource spherical(float intensity, const glm::dvec3 &pos, const glm::vec3 &color);

    bool is_spherical;
    glm::dvec3 p;
    glm::vec3 light_color;
};
