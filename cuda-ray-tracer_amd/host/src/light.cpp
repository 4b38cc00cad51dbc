// light.cpp -- light factories of the scene model (reference src/light.cpp:4-26).
#include "light.h"

#include "scene-exception.h"

namespace {
LightSource make_light(bool spherical, float intensity, const glm::dvec3 &p, const glm::vec3 &color)
{
    validate_positive("light intensity", intensity);
    validate_color(color);
    LightSource l{};
    l.is_spherical = spherical;
    l.light_color = intensity * color; // intensity folded into the colour once, at load time
    l.p = p;
    return l;
}
} // namespace

LightSource LightSource::directional(float intensity, const glm::dvec3 &dir, const glm::vec3 &color)
{
    // stored vector points TOWARDS the light: -normalize(dir), computed in FP64
    validate_positive("light intensity", intensity);
    validate_color(color);
    return make_light(false, intensity, -glm::normalize(dir), color);
}

LightSource LightSource::spherical(float intensity, const glm::dvec3 &pos, const glm::vec3 &color)
{
    return make_light(true, intensity, pos, color);
}
