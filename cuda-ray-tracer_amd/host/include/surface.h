// surface.h -- implicit algebraic surface of degree <= 3 (host-side scene model).
//
// API mirror of the reference's include/surface.h:10-22 so that hosts written against the
// reference compile unchanged: same type name, same 20 public double members in the same order
// (the kernels and the C ABI index them 0..19 in this order), same static factory names.
// Implementation: ../src/surface.cpp.
#pragma once

#include <glm/glm.hpp>

struct SurfaceCoefs {
    // F(x,y,z) = sum of coefficient * monomial; a member is named after its monomial (x2y = x*x*y).
    // degree 3
    double x3, y3, z3;
    double x2y, xy2;
    double x2z, xz2;
    double y2z, yz2;
    double xyz;
    // degree 2
    double x2, y2, z2;
    double xy, xz, yz;
    // degree 1 and constant
    double x, y, z;
    double c;

    // |p - center|^2 = radius^2                      (reference src/surface.cpp:4-15)
    static SurfaceCoefs sphere(const glm::dvec3 &center, double radius);
    // dot(p - origin, nv) = 0                         (reference src/surface.cpp:17-25)
    static SurfaceCoefs plane(const glm::dvec3 &origin, const glm::dvec3 &nv);
    // x^2 + z^2 = y^2 (1 - y) translated to origin    (reference src/surface.cpp:27-39)
    static SurfaceCoefs dingDong(const glm::dvec3 &origin);
    // Clebsch diagonal cubic as the reference defines it, z3 == 0 included (src/surface.cpp:41-52)
    static SurfaceCoefs clebsch();
    // Cayley nodal cubic                              (reference src/surface.cpp:54-60)
    static SurfaceCoefs cayley();

    double *data() { return &x3; }
    const double *data() const { return &x3; }
};
static_assert(sizeof(SurfaceCoefs) == 20 * sizeof(double), "SurfaceCoefs must be 20 packed doubles");
