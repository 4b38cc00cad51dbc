// rt_math.hpp -- per-ray device math of the MI355X ray-tracing path (gfx950, wave64).
//
// Included by rt_kernels.hip, which is compiled twice: once with -ffp-contract=off (variant "strict",
// the parity mode: every FP64/FP32 operation is a separately rounded IEEE operation in the order of the
// reference's expressions) and once with -ffp-contract=fast (variant "fast": the same expression trees,
// FMA contraction allowed).
//
// What is computed follows the reference (cited per function); HOW is this repo's own:
//   * per-ray monomials of (origin, dir) are formed once per ray and shared by all objects
//     (the reference re-derives them inside every intersect_ray call);
//   * objects are dispatched on a precomputed class (rt_scene_dev.h) so that absent groups of
//     coefficients cost nothing -- a wave-uniform branch;
//   * sqrt / division of the quadratic solve are only executed for lanes whose discriminant is
//     non-negative.
#pragma once

#include <hip/hip_runtime.h>
#include "rt_scene_dev.h"

namespace rtm {

// include/surface_impl.h:16-19
constexpr double EPS = 1e-7;
constexpr double TWO_THIRD_PI = 3.14159265358979323846 * 2.0 / 3.0;
constexpr double SHADOW_BIAS = 1e-2;
constexpr double MAX_T = 1e6;
// (float) M_PIf32, include/light_impl.h:38,43
constexpr float PI_F = 3.14159274101257324219f;

struct D3 {
    double x, y, z;
};
struct F3 {
    float x, y, z;
};

// glm::dot(dvec3): (a.x*b.x + a.y*b.y) + a.z*b.z
__device__ __forceinline__ double dot3(const D3 &a, const D3 &b)
{
    return (a.x * b.x + a.y * b.y) + a.z * b.z;
}

// glm::normalize(v) = v * (1 / sqrt(dot(v, v)))
__device__ __forceinline__ D3 normalize3(const D3 &v)
{
    double inv = 1.0 / sqrt(dot3(v, v));
    return D3{v.x * inv, v.y * inv, v.z * inv};
}

// Monomials of one ray that the degree <= 2 part of F(o + t d) needs.  Names follow the factor each
// one multiplies in include/surface_impl.h:25-41 (COEF_2, COEF_1_2, COEF_1_11, COEF_0_2).
struct Mono {
    D3 o, d;
    double dxx, dyy, dzz, dxy, dxz, dyz; // d_a d_b
    double sx, sy, sz;                   // 2 o_a d_a
    double cxy, cxz, cyz;                // o_a d_b + d_a o_b
    double oxx, oyy, ozz, oxy, oxz, oyz; // o_a o_b
    double u2, u1, u0;                   // squared part of t2, t1, t0 for x2 = y2 = z2 = 1
};

__device__ __forceinline__ void make_mono(Mono &m, const D3 &o, const D3 &d)
{
    m.o = o;
    m.d = d;
    m.dxx = d.x * d.x;
    m.dyy = d.y * d.y;
    m.dzz = d.z * d.z;
    m.dxy = d.x * d.y;
    m.dxz = d.x * d.z;
    m.dyz = d.y * d.z;
    m.sx = 2.0 * o.x * d.x;
    m.sy = 2.0 * o.y * d.y;
    m.sz = 2.0 * o.z * d.z;
    m.cxy = o.x * d.y + d.x * o.y;
    m.cxz = o.x * d.z + d.x * o.z;
    m.cyz = o.y * d.z + d.y * o.z;
    m.oxx = o.x * o.x;
    m.oyy = o.y * o.y;
    m.ozz = o.z * o.z;
    m.oxy = o.x * o.y;
    m.oxz = o.x * o.z;
    m.oyz = o.y * o.z;
    m.u2 = (m.dxx + m.dyy) + m.dzz;
    m.u1 = (m.sx + m.sy) + m.sz;
    m.u0 = (m.oxx + m.oyy) + m.ozz;
}

// t2, t1, t0 of a surface without degree-3 terms (t3 is an exact zero), include/surface_impl.h:54-103
// with the zero groups skipped.  `c` may point to global memory (wave-uniform index -> scalar loads) or
// to LDS (per-lane index).
__device__ __forceinline__ void quadric_poly(const double *__restrict__ c, uint32_t cls, const Mono &m,
                                             double &t2, double &t1, double &t0)
{
    if (cls & RT_CLS_UNITSQ) {
        t2 = m.u2;
        t1 = m.u1;
        t0 = m.u0;
    } else if (cls & RT_CLS_SQUARE) {
        const double x2 = c[K_X2], y2 = c[K_Y2], z2 = c[K_Z2];
        t2 = (x2 * m.dxx + y2 * m.dyy) + z2 * m.dzz;
        t1 = (x2 * m.sx + y2 * m.sy) + z2 * m.sz;
        t0 = (x2 * m.oxx + y2 * m.oyy) + z2 * m.ozz;
    } else {
        t2 = 0.0;
        t1 = 0.0;
        t0 = 0.0;
    }
    if (cls & RT_CLS_CROSS) {
        const double xy = c[K_XY], xz = c[K_XZ], yz = c[K_YZ];
        t2 = ((t2 + xy * m.dxy) + xz * m.dxz) + yz * m.dyz;
        t1 = ((t1 + xy * m.cxy) + xz * m.cxz) + yz * m.cyz;
        t0 = ((t0 + xy * m.oxy) + xz * m.oxz) + yz * m.oyz;
    }
    const double kx = c[K_X], ky = c[K_Y], kz = c[K_Z];
    t1 = ((t1 + kx * m.d.x) + ky * m.d.y) + kz * m.d.z;
    t0 = (((t0 + kx * m.o.x) + ky * m.o.y) + kz * m.o.z) + c[K_C];
}

// Root selection for degree <= 2, include/surface_impl.h:138-154.
__device__ __forceinline__ double solve_quadlin(double t2, double t1, double t0)
{
    if (fabs(t2) > EPS) {
        double delta = t1 * t1 - 4.0 * t2 * t0;
        if (delta < 0) return -1.0;
        delta = sqrt(delta);
        double x = (-t1 - delta) / (2.0 * t2);
        if (x >= EPS) return x;
        return (-t1 + delta) / (2.0 * t2);
    }
    if (fabs(t1) > EPS) return -t0 / t1;
    return -1.0;
}

// Dense expansion for surfaces with degree-3 terms, include/surface_impl.h:44-103, all 20 terms in the
// reference's order.  Factor helpers mirror the reference's parenthesisation (argument order matters
// for rounding).
__device__ __forceinline__ double tri(double a, double b, double c) { return a * b * c; }
__device__ __forceinline__ double cube2(double o, double d) { return 3.0 * o * d * d; }
__device__ __forceinline__ double cube1(double o, double d) { return 3.0 * o * o * d; }
__device__ __forceinline__ double sqlin2(double op, double dp, double oq, double dq)
{
    return dp * (dp * oq + 2.0 * op * dq);
}
__device__ __forceinline__ double sqlin1(double op, double dp, double oq, double dq)
{
    return op * (op * dq + 2.0 * dp * oq);
}

__device__ __forceinline__ void cubic_poly(const double *__restrict__ c, const Mono &m, double &t3, double &t2,
                                           double &t1, double &t0)
{
    const double ox = m.o.x, oy = m.o.y, oz = m.o.z, dx = m.d.x, dy = m.d.y, dz = m.d.z;
    const double x3 = c[K_X3], y3 = c[K_Y3], z3 = c[K_Z3], x2y = c[K_X2Y], xy2 = c[K_XY2], x2z = c[K_X2Z],
                 xz2 = c[K_XZ2], y2z = c[K_Y2Z], yz2 = c[K_YZ2], xyz = c[K_XYZ];
    const double x2 = c[K_X2], y2 = c[K_Y2], z2 = c[K_Z2], xy = c[K_XY], xz = c[K_XZ], yz = c[K_YZ];
    double a;
    a = x3 * tri(dx, dx, dx);
    a += y3 * tri(dy, dy, dy);
    a += z3 * tri(dz, dz, dz);
    a += x2y * tri(dx, dx, dy);
    a += xy2 * tri(dx, dy, dy);
    a += x2z * tri(dx, dx, dz);
    a += xz2 * tri(dx, dz, dz);
    a += y2z * tri(dy, dy, dz);
    a += yz2 * tri(dy, dz, dz);
    a += xyz * tri(dx, dy, dz);
    t3 = a;
    a = x3 * cube2(ox, dx);
    a += y3 * cube2(oy, dy);
    a += z3 * cube2(oz, dz);
    a += x2y * sqlin2(ox, dx, oy, dy);
    a += xy2 * sqlin2(oy, dy, ox, dx);
    a += x2z * sqlin2(ox, dx, oz, dz);
    a += xz2 * sqlin2(oz, dz, ox, dx);
    a += y2z * sqlin2(oy, dy, oz, dz);
    a += yz2 * sqlin2(oz, dz, oy, dy);
    a += xyz * (dx * dy * oz + dx * oy * dz + ox * dy * dz);
    a += x2 * m.dxx;
    a += y2 * m.dyy;
    a += z2 * m.dzz;
    a += xy * m.dxy;
    a += xz * m.dxz;
    a += yz * m.dyz;
    t2 = a;
    a = x3 * cube1(ox, dx);
    a += y3 * cube1(oy, dy);
    a += z3 * cube1(oz, dz);
    a += x2y * sqlin1(ox, dx, oy, dy);
    a += xy2 * sqlin1(oy, dy, ox, dx);
    a += x2z * sqlin1(ox, dx, oz, dz);
    a += xz2 * sqlin1(oz, dz, ox, dx);
    a += y2z * sqlin1(oy, dy, oz, dz);
    a += yz2 * sqlin1(oz, dz, oy, dy);
    a += xyz * (dx * oy * oz + ox * dy * oz + ox * oy * dz);
    a += x2 * m.sx;
    a += y2 * m.sy;
    a += z2 * m.sz;
    a += xy * m.cxy;
    a += xz * m.cxz;
    a += yz * m.cyz;
    a += c[K_X] * dx;
    a += c[K_Y] * dy;
    a += c[K_Z] * dz;
    t1 = a;
    a = x3 * tri(ox, ox, ox);
    a += y3 * tri(oy, oy, oy);
    a += z3 * tri(oz, oz, oz);
    a += x2y * tri(ox, ox, oy);
    a += xy2 * tri(ox, oy, oy);
    a += x2z * tri(ox, ox, oz);
    a += xz2 * tri(ox, oz, oz);
    a += y2z * tri(oy, oy, oz);
    a += yz2 * tri(oy, oz, oz);
    a += xyz * tri(ox, oy, oz);
    a += x2 * m.oxx;
    a += y2 * m.oyy;
    a += z2 * m.ozz;
    a += xy * m.oxy;
    a += xz * m.oxz;
    a += yz * m.oyz;
    a += c[K_X] * ox;
    a += c[K_Y] * oy;
    a += c[K_Z] * oz;
    a += c[K_C];
    t0 = a;
}

// Cubic root selection, include/surface_impl.h:106-136: Cardano's single real root
// (unfiltered) or the smallest acceptable of the three trigonometric roots.
__device__ __forceinline__ double solve_cubic(double t3, double t2, double t1, double t0)
{
    t2 /= t3;
    t1 /= t3;
    t0 /= t3;
    double q = (3.0 * t1 - t2 * t2) / 9.0;
    double r = (9.0 * t2 * t1 - 27.0 * t0 - 2.0 * t2 * t2 * t2) / 54.0;
    double delta = q * q * q + r * r;
    if (delta > 0) {
        delta = sqrt(delta);
        q = cbrt(r + delta);
        r = cbrt(r - delta);
        return q + r - t2 / 3.0;
    }
    double theta = acos(r / sqrt(-q * q * q)) / 3.0;
    double c = 2.0 * sqrt(-q);
    double x = c * cos(theta) - t2 / 3.0;
    double x1 = c * cos(theta + TWO_THIRD_PI) - t2 / 3.0;
    if (x1 >= EPS && x1 < x) x = x1;
    x1 = c * cos(theta + 2.0 * TWO_THIRD_PI) - t2 / 3.0;
    if (x1 >= EPS && x1 < x) x = x1;
    return x;
}

// Surfaces with degree-3 terms: dense expansion + full solver.  Kept out of line (one copy per kernel)
// so that the common quadric path stays small in registers; it re-derives the degree <= 2 monomials from
// (o, d), which is noise next to the ~300 flops and the cbrt/acos/cos of this path.
__device__ __forceinline__ double intersect_cubic_inl(const double *c, double ox, double oy, double oz, double dx, double dy, double dz)
{
    Mono m;
    make_mono(m, D3{ox, oy, oz}, D3{dx, dy, dz});
    double t3, t2, t1, t0;
    cubic_poly(c, m, t3, t2, t1, t0);
    if (fabs(t3) > EPS) return solve_cubic(t3, t2, t1, t0);
    return solve_quadlin(t2, t1, t0);
}
__device__ __noinline__ double intersect_cubic(const double *c, double ox, double oy, double oz, double dx, double dy, double dz)
{
    return intersect_cubic_inl(c, ox, oy, oz, dx, dy, dz);
}

// The same value, inlined, also naming the solver branch that produced it (counting builds only):
// 0 Cardano, 1 trigonometric, 2 quadratic, 3 linear / constant.
__device__ __forceinline__ double intersect_cubic_branch(const double *c, double ox, double oy, double oz, double dx, double dy, double dz, int &branch)
{
    Mono m;
    make_mono(m, D3{ox, oy, oz}, D3{dx, dy, dz});
    double t3, t2, t1, t0;
    cubic_poly(c, m, t3, t2, t1, t0);
    if (fabs(t3) > EPS) {
        const double a2 = t2 / t3, a1 = t1 / t3, a0 = t0 / t3; // the first lines of solve_cubic, to see which way it goes
        const double q = (3.0 * a1 - a2 * a2) / 9.0;
        const double r = (9.0 * a2 * a1 - 27.0 * a0 - 2.0 * a2 * a2 * a2) / 54.0;
        branch = (q * q * q + r * r > 0) ? 0 : 1;
        return solve_cubic(t3, t2, t1, t0);
    }
    branch = fabs(t2) > EPS ? 2 : 3;
    return solve_quadlin(t2, t1, t0);
}

// ---- degree-3 surfaces: F(o + t d) as a Taylor polynomial around the ray origin ----
// The reference expands F(o + t d) term by term for every (ray, object) pair (include/surface_impl.h:44-103: 286 operations).
// The same polynomial in t is
//     t0 = F(o),   t1 = grad F(o) . d,   t2 = 1/2 d^T H(o) d,   t3 = C(d)   (C = the cubic form of the degree-3 terms),
// and most of that does not depend on the ray at hand: F, grad F and H at the camera origin are constants of the frame for all primary
// rays; at a hit's shadow-ray origin they are the same for every light.  What is left per test is a dot product, a quadratic form and
// (where the direction differs per lane) the cubic form: ~50 operations instead of 286, ~16 for a directional light.
// Degree-3 scenes are held to 1e-5 relative, not to bit-identity (device cbrt / acos / cos already differ from glibc's in the last
// place), so the re-association is admissible; the flip statistics of tests/ and tests/tools/fuzz_cubic.py bound its effect.
struct CubicAt {
    double f;                          // F(o)
    double gx, gy, gz;                 // grad F(o)
    double hxx, hyy, hzz, hxy, hxz, hyz; // 1/2 d2F/dx2 ..., d2F/dxdy ...: t2 = hxx dx^2 + hyy dy^2 + hzz dz^2 + hxy dx dy + hxz dx dz + hyz dy dz
};

__host__ __device__ __forceinline__ CubicAt cubic_at(const double *__restrict__ c, const D3 &o)
{
    const double x = o.x, y = o.y, z = o.z;
    CubicAt a;
    // second derivatives: linear in o
    a.hxx = ((3.0 * c[K_X3]) * x + c[K_X2Y] * y) + (c[K_X2Z] * z + c[K_X2]);
    a.hyy = ((3.0 * c[K_Y3]) * y + c[K_XY2] * x) + (c[K_Y2Z] * z + c[K_Y2]);
    a.hzz = ((3.0 * c[K_Z3]) * z + c[K_XZ2] * x) + (c[K_YZ2] * y + c[K_Z2]);
    a.hxy = ((2.0 * c[K_X2Y]) * x + (2.0 * c[K_XY2]) * y) + (c[K_XYZ] * z + c[K_XY]);
    a.hxz = ((2.0 * c[K_X2Z]) * x + (2.0 * c[K_XZ2]) * z) + (c[K_XYZ] * y + c[K_XZ]);
    a.hyz = ((2.0 * c[K_Y2Z]) * y + (2.0 * c[K_YZ2]) * z) + (c[K_XYZ] * x + c[K_YZ]);
    // gradient: dF/dx = x (3 x3 x + 2 x2y y + 2 x2z z + 2 x2) + y (xy2 y + xyz z + xy) + z (xz2 z + xz) + kx, and cyclically
    a.gx = (x * (((3.0 * c[K_X3]) * x + (2.0 * c[K_X2Y]) * y) + ((2.0 * c[K_X2Z]) * z + 2.0 * c[K_X2])) + y * ((c[K_XY2] * y + c[K_XYZ] * z) + c[K_XY])) + (z * (c[K_XZ2] * z + c[K_XZ]) + c[K_X]);
    a.gy = (y * (((3.0 * c[K_Y3]) * y + (2.0 * c[K_XY2]) * x) + ((2.0 * c[K_Y2Z]) * z + 2.0 * c[K_Y2])) + x * ((c[K_X2Y] * x + c[K_XYZ] * z) + c[K_XY])) + (z * (c[K_YZ2] * z + c[K_YZ]) + c[K_Y]);
    a.gz = (z * (((3.0 * c[K_Z3]) * z + (2.0 * c[K_XZ2]) * x) + ((2.0 * c[K_YZ2]) * y + 2.0 * c[K_Z2])) + x * ((c[K_X2Z] * x + c[K_XYZ] * y) + c[K_XZ])) + (y * (c[K_Y2Z] * y + c[K_YZ]) + c[K_Z]);
    // F(o), grouped by the leading variable
    const double fx = x * (x * ((c[K_X3] * x + c[K_X2Y] * y) + (c[K_X2Z] * z + c[K_X2])) + ((y * ((c[K_XY2] * y + c[K_XYZ] * z) + c[K_XY]) + z * (c[K_XZ2] * z + c[K_XZ])) + c[K_X]));
    const double fy = y * (y * ((c[K_Y3] * y + c[K_Y2Z] * z) + c[K_Y2]) + (z * (c[K_YZ2] * z + c[K_YZ]) + c[K_Y]));
    const double fz = z * (z * (c[K_Z3] * z + c[K_Z2]) + c[K_Z]);
    a.f = (fx + fy) + (fz + c[K_C]);
    return a;
}

// t3 .. t0 of the ray (o, d) from the surface's Taylor data at o and the direction.
__device__ __forceinline__ void cubic_coefs(const double *__restrict__ c, const CubicAt &a, const D3 &d, double &t3, double &t2, double &t1, double &t0)
{
    const double dxx = d.x * d.x, dyy = d.y * d.y, dzz = d.z * d.z, dxy = d.x * d.y, dxz = d.x * d.z, dyz = d.y * d.z;
    t0 = a.f;
    t1 = (a.gx * d.x + a.gy * d.y) + a.gz * d.z;
    t2 = ((a.hxx * dxx + a.hyy * dyy) + (a.hzz * dzz + a.hxy * dxy)) + (a.hxz * dxz + a.hyz * dyz);
    t3 = (((c[K_X3] * dxx + c[K_XY2] * dyy) + (c[K_XZ2] * dzz + c[K_XYZ] * dyz)) * d.x + ((c[K_Y3] * dyy + c[K_X2Y] * dxx) + c[K_YZ2] * dzz) * d.y) +
         ((c[K_Z3] * dzz + c[K_X2Z] * dxx) + c[K_Y2Z] * dyy) * d.z;
}

// Root of the reference's solver for a degree-3 surface, from the Taylor data (kept out of line like intersect_cubic; the data by value,
// in registers: a pointer to it would put it in scratch memory).
__device__ __noinline__ double intersect_cubic_at(const double *c, double f, double gx, double gy, double gz, double hxx, double hyy, double hzz, double hxy, double hxz,
                                                  double hyz, double dx, double dy, double dz)
{
    const CubicAt a{f, gx, gy, gz, hxx, hyy, hzz, hxy, hxz, hyz};
    double t3, t2, t1, t0;
    cubic_coefs(c, a, D3{dx, dy, dz}, t3, t2, t1, t0);
    if (fabs(t3) > EPS) return solve_cubic(t3, t2, t1, t0);
    return solve_quadlin(t2, t1, t0);
}
__device__ __forceinline__ double intersect_cubic_at(const double *c, const CubicAt &a, const D3 &d)
{
    return intersect_cubic_at(c, a.f, a.gx, a.gy, a.gz, a.hxx, a.hyy, a.hzz, a.hxy, a.hxz, a.hyz, d.x, d.y, d.z);
}

// The same value, inlined, also naming the solver branch that produced it (counting builds only): 0 Cardano, 1 trigonometric, 2 quadratic, 3 linear / constant.
__device__ __forceinline__ double intersect_cubic_at_branch(const double *c, const CubicAt &a, const D3 &d, int &branch)
{
    double t3, t2, t1, t0;
    cubic_coefs(c, a, d, t3, t2, t1, t0);
    if (fabs(t3) > EPS) {
        const double a2 = t2 / t3, a1 = t1 / t3, a0 = t0 / t3; // the first lines of solve_cubic, to see which way it goes
        const double q = (3.0 * a1 - a2 * a2) / 9.0;
        const double r = (9.0 * a2 * a1 - 27.0 * a0 - 2.0 * a2 * a2 * a2) / 54.0;
        branch = (q * q * q + r * r > 0) ? 0 : 1;
        return solve_cubic(t3, t2, t1, t0);
    }
    branch = fabs(t2) > EPS ? 2 : 3;
    return solve_quadlin(t2, t1, t0);
}

// ---- degree-3 surfaces, strict build: the Taylor polynomial behind a GUARD ----
// The reference's solver (solve_cubic above, include/surface_impl.h:106-136) is not a continuous function of t3 .. t0: the sign of its
// discriminant picks Cardano's formula or the trigonometric one, the trigonometric roots are filtered with `>= EPS`, callers compare
// the result with EPS and max_t -- and where |t3| is small against the other coefficients, or q^3 << r^2, the formulas as written lose
// most of their digits, so even their rounding noise depends on the exact bits of the coefficients.  The Taylor coefficients equal the
// dense expansion's up to rounding only (both evaluate the same polynomial in o and d; each is off by some 30 roundings of its largest
// term), so using them blindly moves pixels across those discontinuities (scenes/cayley.yml from the origin: F(0) = 0, a double root at
// t = 0 in EVERY primary ray, Cardano-or-trigonometric decided by the last bit).  cubic_guarded therefore carries a running bound of
// how far its result can be from what the reference's expressions give on the dense coefficients -- the coefficients' own uncertainty
// (CubicMag: E x the sum of the absolute values of a coefficient's terms) pushed through every step of the solver, plus the roundings of
// the steps themselves -- and refuses (returns false: the caller takes the dense expansion and the reference's solver, as before)
// whenever a decision (|t3| > EPS, the discriminant's sign, a root against EPS or max_t) is closer to its threshold than CUB_K times
// that bound, or an accepted root is uncertain by more than CUB_TOL of its value.  What it returns otherwise is the reference's result up
// to CUB_TOL -- three orders of magnitude inside the 1e-5 the north star allows degree-3 scenes -- computed with one division instead
// of nine and one sincos instead of three cosines (it need not mimic the reference's operation order: that is the guard's job).
struct CubicMag {
    double m3, m2, m1, m0; // uncertainty of t3 .. t0: CUB_E x an upper bound of the sum of the coefficient's |terms|
};
struct CubicAbs {        // per object: CUB_E x the sums of |coefficients| by degree
    double a3, a2, a1, a0;
};
constexpr double CUB_E = 0x1p-46;   // relative uncertainty of a coefficient against the sum of its |terms| (dense vs Taylor evaluation: some 30 roundings each)
__host__ __device__ __forceinline__ CubicAbs cubic_abs(const double *__restrict__ c)
{
    CubicAbs a;
    a.a3 = 0.0;
    for (int k = K_X3; k <= K_XYZ; k++) a.a3 += fabs(c[k]);
    a.a2 = 0.0;
    for (int k = K_X2; k <= K_YZ; k++) a.a2 += fabs(c[k]);
    a.a1 = (fabs(c[K_X]) + fabs(c[K_Y])) + fabs(c[K_Z]);
    a.a0 = fabs(c[K_C]);
    a.a3 *= CUB_E; a.a2 *= CUB_E; a.a1 *= CUB_E; a.a0 *= CUB_E;
    return a;
}
// the origin's part of the bounds (once per origin) ...
__host__ __device__ __forceinline__ CubicMag cubic_mag_origin(const CubicAbs &a, const D3 &o)
{
    const double s = fmax(fmax(fabs(o.x), fabs(o.y)), fabs(o.z));
    CubicMag m;
    m.m3 = a.a3;
    m.m2 = 3.0 * a.a3 * s + a.a2;
    m.m1 = (3.0 * a.a3 * s + 2.0 * a.a2) * s + a.a1;
    m.m0 = ((a.a3 * s + a.a2) * s + a.a1) * s + a.a0;
    return m;
}
// ... and the direction's (dmax = the largest |component| of d)
__host__ __device__ __forceinline__ CubicMag cubic_mag_dir(const CubicMag &m, double dmax)
{
    const double d2 = dmax * dmax;
    return CubicMag{m.m3 * d2 * dmax, m.m2 * d2, m.m1 * dmax, m.m0};
}

constexpr double CUB_U = 0x1p-52;
constexpr double CUB_K = 16.0;      // safety factor between a decision's distance from its threshold and the bound
constexpr double CUB_TOL = 1e-8;    // accepted relative uncertainty of a root that is used

#ifdef RT_CUB_LAB // (tests/tools/cubic_guard_lab.cpp: which check sent a test back to the dense path)
static int g_cub_why = 0;
#define CUB_REFUSE(n) (g_cub_why = (n), false)
#else
#define CUB_REFUSE(n) false
#endif

// Reciprocals.  cub_rcpa: a few good digits are enough (error bounds, Newton corrections) -- on the device the bare v_rcp_f64, one
// instruction where a division is a dozen.  Measured on gfx950 (scratch probe, 2^24 arguments over 16 binades): v_rcp_f64 and v_rsq_f64 are
// good to 2^-24.4 / 2^-24.2; the code trusts them to 2^-20 (the host stand-in is made 2^-21 wrong on purpose when RT_CUB_LAB is set, so
// the CPU experiment covers it).  cub_rcp: one Newton step on top -- 2^-40 at worst by that assumption, 2^-48 as measured; what is left
// is part of the relative uncertainty the callers book for it (r3 below).
__host__ __device__ __forceinline__ double cub_rcpa(double x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcp(x);
#elif defined(RT_CUB_LAB)
    return (1.0 / x) * (1.0 + 0x1p-21);
#else
    return 1.0 / x;
#endif
}
__host__ __device__ __forceinline__ double cub_rcp(double x)
{
    const double y = cub_rcpa(x);
    return fma(y, fma(-x, y, 1.0), y);
}

// Square roots of this function's own intermediate values: v_rsq_f64, a coupled Newton step and a correction (no scaling for denormal or huge
// arguments -- an argument out there ends in inf / NaN, which every check below refuses).  cub_rsqa: a few good digits (error bounds).
__host__ __device__ __forceinline__ double cub_rsqa(double x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rsq(x);
#elif defined(RT_CUB_LAB)
    return (1.0 / sqrt(x)) * (1.0 + 0x1p-21);
#else
    return 1.0 / sqrt(x);
#endif
}
__host__ __device__ __forceinline__ double cub_sqrt(double x)
{
    const double y = cub_rsqa(x);
    double g = x * y, h = 0.5 * y;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    return fma(fma(-g, g, x), h, g); // (one coupled step, one more correction of g alone: 2^-20 -> 2^-39 -> below 2^-50)
}

// false: no verdict (take the reference's path).  true: `t` is what the reference's solver returns for this ray up to CUB_TOL, and
// every comparison the callers make with it (t >= EPS, t > EPS, t < max_t) comes out as with the reference's value.
// Error bounds: e_i = mg.m_i >= 64 u |t_i| by construction (CubicMag), so the roundings of the steps below and of the reference's own
// evaluation -- a few u of each term -- are an order of magnitude inside the propagated bounds and live in CUB_K's slack.
// decide: the caller only asks "EPS < t < max_t?" (a shadow ray): a root that is used need not be sharp, and `t` need not be the root the
// reference selects as long as that question gets the reference's answer.
__host__ __device__ __forceinline__ bool cubic_guarded(double t3, double t2, double t1, double t0, const CubicMag &mg, double max_t, bool decide, double &t)
{
    const double at3 = fabs(t3);
    const double e3 = mg.m3;
    double root, eroot;
    if (at3 > EPS + (EPS + CUB_K * e3)) { // ---- solve_cubic, include/surface_impl.h:106-136 ----
        const double inv = cub_rcp(t3), ainv = fabs(inv);
        const double a2 = t2 * inv, a1 = t1 * inv, a0 = t0 * inv;
        const double r3 = e3 * ainv + 0x1p-39;      // relative uncertainty of 1 / t3 (its own rounding: cub_rcp)
        const double ea2 = fma(mg.m2, ainv, fabs(a2) * r3);
        const double ea1 = fma(mg.m1, ainv, fabs(a1) * r3);
        const double ea0 = fma(mg.m0, ainv, fabs(a0) * r3);
        const double a2s = a2 * a2;
        const double q = fma(3.0, a1, -a2s) * (1.0 / 9.0);
        const double eq = fma(3.0, ea1, 2.0 * fabs(a2) * ea2) * (1.0 / 9.0);
        const double r = fma(fma(9.0, a1, -2.0 * a2s), a2, -27.0 * a0) * (1.0 / 54.0);
        const double er = fma(9.0, fma(fabs(a2), ea1, fabs(a1) * ea2), fma(6.0 * a2s, ea2, 27.0 * ea0)) * (1.0 / 54.0);
        const double q2 = q * q, q3 = q2 * q, r2 = r * r;
        const double delta = q3 + r2;
        const double ed = fma(3.0 * q2, eq, 2.0 * fabs(r) * er);
        if (!(fabs(delta) > CUB_K * ed)) return CUB_REFUSE(2); // Cardano or trigonometric: not for this function to say
        const double third_a2 = a2 * (1.0 / 3.0);
        if (delta > 0.0) {
            // one real root: cbrt(r + s) + cbrt(r - s) - a2 / 3.  Here: the cube root of the term that does not cancel, the other one from
            // their product (-q); what the reference's cancelling term can be off by is in the bound (eab over the small root's square)
            const double s = cub_sqrt(delta);
            const double es = ed * cub_rcpa(s); // (>= ed / (2 s))
            const double eab = er + es;
            const double ca = cbrt(r + copysign(s, r));
            const double ra = cub_rcp(ca);
            const double cb = -q * ra;
            const double rb = cub_rcpa(cb); // (cb = 0: infinite -- refused below)
            root = (ca + cb) - third_a2;
            eroot = fma(eab * (1.0 / 3.0), fma(ra, ra, rb * rb), fma(eq, fabs(ra), ea2 * (1.0 / 3.0))) + 0x1p-39 * fabs(cb); // (the last term: ra's own rounding)
        } else {
            // three real roots 2 m cos(theta + 2 k pi / 3) - a2 / 3, cos(3 theta) = r / m^3.  No acos / cos: c = cos(theta) is the root of
            // 4 c^3 - 3 c = arg in [1/2, 1], four Newton steps from 1/2 + sqrt((1 + arg) / 8) (good to 0.016: then 8e-4, 2e-6, 2e-11, 1e-16);
            // the size of the last correction -- the error before it -- is taken as the error after it.  sin(theta) >= 0 from c.
            const double m2 = -q; // > 0: delta < 0 means q^3 < -r^2 <= 0
            const double m = cub_sqrt(m2);
            const double rm = cub_rcp(m), rm3 = rm * rm * rm;
            const double em = 0.5 * eq * rm;
            const double arg = r * rm3;
            const double earg = fma(fma(3.0 * fabs(arg) * m2, em, er), rm3, 0x1p-38 * fabs(arg)); // (the last term: rm's own rounding, three times)
            const double w2 = -delta * (rm3 * rm3); // 1 - arg^2, without the cancellation
            double c = 0.5 + (double) sqrtf((float) fma(0.125, arg, 0.125));
            double dc = 0.0;
#pragma unroll
            for (int it = 0; it < 4; it++) {
                if (it == 3 && decide) break; // (a decision does with three steps: the bands below just get wider)
                const double c2 = c * c;
                dc = fma(fma(4.0, c2, -3.0), c, -arg) * cub_rcpa(fma(12.0, c2, -3.0));
                c -= dc;
            }
            const double eth = earg * (1.0 / 3.0) * cub_rsqa(w2); // of theta
            const double ec = fabs(dc) + 4.0 * CUB_U;            // of c, on top of theta's
            const double two_m = 2.0 * m;
            const double x0 = fma(two_m, c, -third_a2); // the largest root
            const double e0 = fma(two_m, eth + ec, fma(2.0, em, ea2 * (1.0 / 3.0))); // (|d cos| <= |d theta|)
            bool done = false;
            if (decide) {
                // the reference returns the smallest root >= EPS, or x0 when only x0 or none is: with x0 < EPS that is x0 itself (no), with
                // EPS < x0 < max_t whatever it returns lies in [EPS, x0] (yes; `t > EPS` is `t >= EPS` away from the band) -- the other two
                // roots matter only when x0 is beyond max_t
                const double band0 = CUB_K * e0;
                if (!(fabs(x0 - EPS) > band0 && fabs(x0 - max_t) > band0)) return CUB_REFUSE(7);
                done = x0 < max_t;
                root = x0;
                eroot = e0;
            }
            if (!done) {
                const double sn = cub_sqrt(fmax(fma(-c, c, 1.0), 0.0));
                const double hs = 0.86602540378443864676 * sn; // cos(theta +- 2 pi / 3) = -cos(theta) / 2 -+ sqrt(3) / 2 sin(theta)
                const double x1 = fma(two_m, fma(-0.5, c, -hs), -third_a2); // the smallest root
                const double x2 = fma(two_m, fma(-0.5, c, hs), -third_a2);  // the middle one
                eroot = fma(two_m, eth + ec * (1.0 + c * cub_rcpa(fmax(sn, 0x1p-500))), fma(2.0, em, ea2 * (1.0 / 3.0)));
                const double band = CUB_K * eroot;
                if (!(fabs(x1 - EPS) > band && fabs(x2 - EPS) > band)) return CUB_REFUSE(3); // the reference filters these two with `>= EPS`
                root = x0;
                if (x1 >= EPS && x1 < root) root = x1;
                if (x2 >= EPS && x2 < root) root = x2;
            }
        }
    } else if (at3 + CUB_K * e3 < EPS) { // ---- solve_quadlin, include/surface_impl.h:138-154 ----
        const double at2 = fabs(t2), at1 = fabs(t1);
        const double e2 = mg.m2, e1 = mg.m1, e0 = mg.m0;
        if (at2 > EPS + (EPS + CUB_K * e2)) {
            const double t20 = t2 * t0;
            const double delta = fma(t1, t1, -4.0 * t20);
            const double ed = fma(2.0 * at1, e1, 4.0 * fma(at2, e0, fabs(t0) * e2));
            if (!(fabs(delta) > CUB_K * ed)) return CUB_REFUSE(4);
            if (delta < 0.0) {
                t = -1.0;
                return true;
            }
            const double s = cub_sqrt(delta), es = ed * cub_rcpa(s);
            const double h = 0.5 * cub_rcp(t2), ah = fabs(h);
            const double rel2 = 2.0 * e2 * ah + 0x1p-39;
            const double en = e1 + es; // of the numerators -t1 -+ s
            const double xa = (-t1 - s) * h;
            const double exa = fma(en, ah, fabs(xa) * rel2);
            if (!(fabs(xa - EPS) > CUB_K * exa)) return CUB_REFUSE(5); // `if (x >= EPS) return x;`
            if (xa >= EPS) {
                root = xa;
                eroot = exa;
            } else {
                root = (-t1 + s) * h;
                eroot = fma(en, ah, fabs(root) * rel2);
            }
        } else if (at2 + CUB_K * e2 < EPS) {
            if (at1 > EPS + (EPS + CUB_K * e1)) {
                const double r1 = cub_rcp(t1);
                root = -t0 * r1;
                eroot = fma(fabs(root), e1, e0) * fabs(r1) + 0x1p-39 * fabs(root);
            } else if (at1 + CUB_K * e1 < EPS) {
                t = -1.0;
                return true;
            } else {
                return CUB_REFUSE(6);
            }
        } else {
            return CUB_REFUSE(6);
        }
    } else {
        return CUB_REFUSE(1); // |t3| too close to EPS (NaN lands here too)
    }
    t = root;
    const double band = CUB_K * eroot;
    if (!(fabs(root - EPS) > band && fabs(root - max_t) > band)) return CUB_REFUSE(7); // the callers' comparisons
    if (!decide && !(root < EPS || root >= max_t || eroot <= CUB_TOL * root)) return CUB_REFUSE(8); // a root that is used must be sharp
    return true;
}

// A degree-3 object against the ray (o, d), as both kernels do it: the Taylor coefficients from the surface's data at o (`ca`, and `mo` =
// cubic_mag_origin there) through cubic_guarded; where the guard refuses, the dense expansion and the reference's solver.  max_t: what the
// caller compares the result with (MAX_T for a nearest-hit search).  `refused` reports which way it went (work counters).
template <bool DENSE_INLINE = false> // (DENSE_INLINE: the caller is itself an out-of-line function -- a call from there would need a stack frame)
__device__ __forceinline__ double intersect_cubic_taylor(const double *__restrict__ c, const CubicAt &ca, const CubicMag &mo, const D3 &o, const D3 &d, double max_t,
                                                         bool decide, bool &refused)
{
    double t3, t2, t1, t0, t;
    cubic_coefs(c, ca, d, t3, t2, t1, t0);
    const CubicMag mg = cubic_mag_dir(mo, fmax(fmax(fabs(d.x), fabs(d.y)), fabs(d.z)));
    refused = !cubic_guarded(t3, t2, t1, t0, mg, max_t, decide, t);
    if (refused) t = DENSE_INLINE ? intersect_cubic_inl(c, o.x, o.y, o.z, d.x, d.y, d.z) : intersect_cubic(c, o.x, o.y, o.z, d.x, d.y, d.z);
    return t;
}

// intersect_ray, include/surface_impl.h:21-155: parameter of the root the reference would return (degree <= 2: exactly; degree 3: see
// cubic_guarded).  max_t, decide as above.
__device__ __forceinline__ double intersect(const double *__restrict__ c, uint32_t cls, const Mono &m, double max_t, bool decide)
{
    if (cls & RT_CLS_CUBIC) {
        bool refused;
        return intersect_cubic_taylor(c, cubic_at(c, m.o), cubic_mag_origin(cubic_abs(c), m.o), m.o, m.d, max_t, decide, refused);
    }
    double t2, t1, t0;
    quadric_poly(c, cls, m, t2, t1, t0);
    return solve_quadlin(t2, t1, t0);
}

// normal_vector, include/surface_impl.h:157-172: normalised gradient, never flipped.
__device__ __forceinline__ D3 normal_vector(const double *__restrict__ c, const D3 &p)
{
    D3 g;
    g.x = ((3.0 * c[K_X3]) * p.x) * p.x + (2.0 * c[K_X2]) * p.x + c[K_X];
    g.y = ((3.0 * c[K_Y3]) * p.y) * p.y + (2.0 * c[K_Y2]) * p.y + c[K_Y];
    g.z = ((3.0 * c[K_Z3]) * p.z) * p.z + (2.0 * c[K_Z2]) * p.z + c[K_Z];
    g.x += 2.0 * p.x * (c[K_X2Y] * p.y + c[K_X2Z] * p.z) + p.y * (c[K_XY2] * p.y + c[K_XYZ] * p.z + c[K_XY])
           + p.z * (c[K_XZ2] * p.z + c[K_XZ]);
    g.y += 2.0 * p.y * (c[K_XY2] * p.x + c[K_Y2Z] * p.z) + p.x * (c[K_X2Y] * p.x + c[K_XYZ] * p.z + c[K_XY])
           + p.z * (c[K_YZ2] * p.z + c[K_YZ]);
    g.z += 2.0 * p.z * (c[K_XZ2] * p.x + c[K_YZ2] * p.y) + p.x * (c[K_X2Z] * p.x + c[K_XYZ] * p.y + c[K_XZ])
           + p.y * (c[K_Y2Z] * p.y + c[K_YZ]);
    return normalize3(g);
}

// shadow_ray, include/light_impl.h:17-27: the direction passes through FP32.
__device__ __forceinline__ D3 shadow_dir(const double *__restrict__ lp, bool spherical, const D3 &sp, double &max_t)
{
    float fx, fy, fz;
    if (spherical) {
        max_t = 1.0;
        fx = (float) (lp[0] - sp.x);
        fy = (float) (lp[1] - sp.y);
        fz = (float) (lp[2] - sp.z);
    } else {
        max_t = 1e6;
        fx = (float) lp[0];
        fy = (float) lp[1];
        fz = (float) lp[2];
    }
    return D3{(double) fx, (double) fy, (double) fz};
}

// surface_color, include/light_impl.h:29-44
__device__ __forceinline__ F3 surface_color(const double *__restrict__ lp, const float *__restrict__ lc,
                                            bool spherical, const D3 &p, const D3 &n, const F3 &albedo)
{
    D3 dir;
    F3 col;
    if (spherical) {
        dir = D3{lp[0] - p.x, lp[1] - p.y, lp[2] - p.z};
        float denom = 4.0f * PI_F * (float) dot3(dir, dir);
        col = F3{lc[0] / denom, lc[1] / denom, lc[2] / denom};
        dir = normalize3(dir);
    } else {
        dir = D3{lp[0], lp[1], lp[2]};
        col = F3{lc[0], lc[1], lc[2]};
    }
    float lam = (float) dot3(n, dir);
    float mx = (0.0f < lam) ? lam : 0.0f; // glm::max(0.0f, lam)
    return F3{albedo.x / PI_F * col.x * mx, albedo.y / PI_F * col.y * mx, albedo.z / PI_F * col.z * mx};
}

// surface_color with the per-hit factor object_color / pi formed once by the caller: the reference evaluates
// ((object_color / pi) * color) * max(0, n.l) left to right (include/light_impl.h:43), so hoisting the first
// division out of the light loop does not change a bit -- it removes three FP32 divisions per (hit, light).
__device__ __forceinline__ F3 surface_color_pre(const double *__restrict__ lp, const float *__restrict__ lc, bool spherical,
                                                const D3 &p, const D3 &n, const F3 &albedo_over_pi)
{
    D3 dir;
    F3 col;
    if (spherical) {
        dir = D3{lp[0] - p.x, lp[1] - p.y, lp[2] - p.z};
        float denom = 4.0f * PI_F * (float) dot3(dir, dir);
        col = F3{lc[0] / denom, lc[1] / denom, lc[2] / denom};
        dir = normalize3(dir);
    } else {
        dir = D3{lp[0], lp[1], lp[2]};
        col = F3{lc[0], lc[1], lc[2]};
    }
    float lam = (float) dot3(n, dir);
    float mx = (0.0f < lam) ? lam : 0.0f;
    return F3{albedo_over_pi.x * col.x * mx, albedo_over_pi.y * col.y * mx, albedo_over_pi.z * col.z * mx};
}

// reflect_ray, include/light_impl.h:46-49
__device__ __forceinline__ D3 reflect_ray(const D3 &d, const D3 &n)
{
    double s = 2.0 * dot3(d, n);
    return D3{d.x - s * n.x, d.y - s * n.y, d.z - s * n.z};
}

// Primary-ray direction of pixel (x, y), src/update-cpu.cpp:84-89.
__device__ __forceinline__ D3 primary_dir(const FrameArgs &fa, int x, int y)
{
    double ndc_x = (x + 0.5) / (int) fa.width;
    double ndc_y = (y + 0.5) / (int) fa.height;
    double cx = (2.0 * ndc_x - 1.0) * fa.aspect * fa.tan_half_fov;
    double cy = (2.0 * ndc_y - 1.0) * fa.tan_half_fov;
    const double *m = fa.cam;
    // dmat4 * dvec4(cx, cy, 1, 1): (m0*v.x + m1*v.y) + (m2*v.z + m3*v.w)
    D3 w;
    w.x = (m[0] * cx + m[4] * cy) + (m[8] * 1.0 + m[12] * 1.0);
    w.y = (m[1] * cx + m[5] * cy) + (m[9] * 1.0 + m[13] * 1.0);
    w.z = (m[2] * cx + m[6] * cy) + (m[10] * 1.0 + m[14] * 1.0);
    D3 rel{w.x - fa.origin[0], w.y - fa.origin[1], w.z - fa.origin[2]};
    return normalize3(rel);
}

// The same direction from per-column / per-row tables: camera_x and camera_y above depend only on the pixel
// column / row and on (width, height, fov), so rt_create evaluates them once on the host with the same IEEE
// operations (two FP64 divisions per pixel saved); the per-frame part is the camera matrix and the normalisation.
__device__ __forceinline__ D3 primary_dir_tab(const FrameArgs &fa, double cx, double cy)
{
    const double *m = fa.cam;
    D3 w;
    w.x = (m[0] * cx + m[4] * cy) + (m[8] * 1.0 + m[12] * 1.0);
    w.y = (m[1] * cx + m[5] * cy) + (m[9] * 1.0 + m[13] * 1.0);
    w.z = (m[2] * cx + m[6] * cy) + (m[10] * 1.0 + m[14] * 1.0);
    D3 rel{w.x - fa.origin[0], w.y - fa.origin[1], w.z - fa.origin[2]};
    return normalize3(rel);
}

// Global image row of local row `lr` under band-cyclic ownership: band b = lr / B of this rank is global
// band b * world + rank.
__device__ __forceinline__ uint32_t global_row(const FrameArgs &fa, uint32_t lr)
{
    uint32_t b = lr / fa.band_rows;
    return (b * fa.world + fa.rank) * fa.band_rows + (lr - b * fa.band_rows);
}

} // namespace rtm
