// surface.cpp -- surface factories of the scene model (host side, load time).
//
// Same coefficient values as the reference's factories (src/surface.cpp:4-60); the arithmetic that
// produces them is kept in the reference's order because the values are parity-critical inputs of the
// kernels.  Build with -ffp-contract=off.
#include "surface.h"

#include "scene-exception.h"

namespace {
SurfaceCoefs zero_surface()
{
    SurfaceCoefs s{};
    return s;
}
} // namespace

SurfaceCoefs SurfaceCoefs::sphere(const glm::dvec3 &center, double radius)
{
    validate_positive("sphere radius", radius);
    SurfaceCoefs s = zero_surface();
    s.x2 = 1.0;
    s.y2 = 1.0;
    s.z2 = 1.0;
    s.x = -2.0 * center.x;
    s.y = -2.0 * center.y;
    s.z = -2.0 * center.z;
    s.c = glm::dot(center, center) - radius * radius;
    return s;
}

SurfaceCoefs SurfaceCoefs::plane(const glm::dvec3 &origin, const glm::dvec3 &nv)
{
    SurfaceCoefs s = zero_surface();
    s.x = nv.x;
    s.y = nv.y;
    s.z = nv.z;
    s.c = -glm::dot(origin, nv);
    return s;
}

SurfaceCoefs SurfaceCoefs::dingDong(const glm::dvec3 &origin)
{
    SurfaceCoefs s = zero_surface();
    s.x2 = 1.0;
    s.z2 = 1.0;
    s.y3 = 1.0;
    s.y2 = -1.0 - 3.0 * origin.y;
    s.x = -2.0 * origin.x;
    s.z = -2.0 * origin.z;
    s.y = (2.0 + 3.0 * origin.y) * origin.y;
    s.c = glm::pow(origin.x, 2) + glm::pow(origin.z, 2) - glm::pow(origin.y, 2) * (1.0 + origin.y);
    return s;
}

SurfaceCoefs SurfaceCoefs::clebsch()
{
    // The reference's table sets x3 and y3 to 81 and never sets z3 (src/surface.cpp:44), so the surface it
    // renders has z3 = 0.  Kept: the scenes and the parity oracle are defined by that behaviour.
    SurfaceCoefs s = zero_surface();
    s.x3 = 81.0;
    s.y3 = 81.0;
    s.z3 = 0.0;
    s.x2y = s.x2z = s.xy2 = s.y2z = s.xz2 = s.yz2 = -189.0;
    s.xyz = 54.0;
    s.xy = s.yz = s.xz = 126.0;
    s.x2 = s.y2 = s.z2 = -9.0;
    s.x = s.y = s.z = 9.0;
    s.c = 1.0;
    return s;
}

SurfaceCoefs SurfaceCoefs::cayley()
{
    SurfaceCoefs s = zero_surface();
    s.x2y = s.x2z = s.xy2 = s.y2z = s.xz2 = s.yz2 = -5.0;
    s.xy = s.yz = s.xz = 2.0;
    return s;
}
