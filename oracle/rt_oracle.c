/*
 * rt_oracle.c -- CPU ORACLE (test infrastructure, NOT product code; see rt_oracle.h for the rules and
 * for the "parity unpinned" statement).
 *
 * Plain-C restatement of the reference CPU back end.  Arithmetic type and operation ORDER follow the
 * reference exactly (FP64 geometry, FP32 colour, one FP64->FP32->FP64 round trip on the shadow
 * direction); the code structure is this repo's own (arrays + small inline helpers instead of glm
 * types and macros).  Build with -ffp-contract=off and without -ffast-math.
 */
#define _GNU_SOURCE
#include "rt_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* include/surface_impl.h:16-19 */
static const double K_EPS = 1e-7;
static const double K_TWO_THIRD_PI = M_PI * 2.0 / 3.0;
static const double K_SHADOW_BIAS = 1e-2;
static const double K_MAX_T = 1e6;
/* (float) M_PIf32, include/light_impl.h:38,43 */
static const float K_PI_F = 3.14159274101257324219f;

/* ------------------------------------------------------------------------------------------------
 * glm operations the reference relies on, written out (glm is an unpinned dependency;
 * this operation order defines parity at that boundary).
 * ---------------------------------------------------------------------------------------------- */
static inline double dot3(const double a[3], const double b[3])
{
    /* glm::dot(vec3): tmp = a*b; (tmp.x + tmp.y) + tmp.z */
    return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2];
}

static inline void normalize3(const double v[3], double out[3])
{
    /* glm::normalize(v) = v * inversesqrt(dot(v,v)), inversesqrt(x) = 1/sqrt(x) */
    double inv = 1.0 / sqrt(dot3(v, v));
    out[0] = v[0] * inv;
    out[1] = v[1] * inv;
    out[2] = v[2] * inv;
}

static inline void cross3(const double a[3], const double b[3], double out[3])
{
    /* glm::cross */
    out[0] = a[1] * b[2] - b[1] * a[2];
    out[1] = a[2] * b[0] - b[2] * a[0];
    out[2] = a[0] * b[1] - b[0] * a[1];
}

/* glm dmat4 * dvec4 (column-major m[col*4+row]): (m0*v.x + m1*v.y) + (m2*v.z + m3*v.w) */
static inline void mat4_mul_vec4(const double m[16], const double v[4], double out[4])
{
    for (int r = 0; r < 4; r++) {
        out[r] = (m[0 + r] * v[0] + m[4 + r] * v[1]) + (m[8 + r] * v[2] + m[12 + r] * v[3]);
    }
}

/* glm::radians, used by Scene::Scene, src/scene.cpp:20 */
double orc_radians(double deg)
{
    return deg * 0.01745329251994329576923690768489;
}

/* ------------------------------------------------------------------------------------------------
 * Factories -- src/surface.cpp, src/light.cpp
 * ---------------------------------------------------------------------------------------------- */
/* SurfaceCoefs::sphere, src/surface.cpp:4-15 (validate_positive is the loader's business) */
void orc_surface_sphere(const double center[3], double radius, double out[ORC_NCOEF])
{
    memset(out, 0, sizeof(double) * ORC_NCOEF);
    out[ORC_X2] = out[ORC_Y2] = out[ORC_Z2] = 1.0;
    out[ORC_X] = -2.0 * center[0];
    out[ORC_Y] = -2.0 * center[1];
    out[ORC_Z] = -2.0 * center[2];
    out[ORC_C] = dot3(center, center) - radius * radius;
}

/* SurfaceCoefs::plane, src/surface.cpp:17-25 */
void orc_surface_plane(const double origin[3], const double nv[3], double out[ORC_NCOEF])
{
    memset(out, 0, sizeof(double) * ORC_NCOEF);
    out[ORC_X] = nv[0];
    out[ORC_Y] = nv[1];
    out[ORC_Z] = nv[2];
    out[ORC_C] = -dot3(origin, nv);
}

/* SurfaceCoefs::dingDong, src/surface.cpp:27-39 */
void orc_surface_dingdong(const double origin[3], double out[ORC_NCOEF])
{
    memset(out, 0, sizeof(double) * ORC_NCOEF);
    out[ORC_X2] = out[ORC_Y3] = out[ORC_Z2] = 1.0;
    out[ORC_Y2] = -1.0 - 3.0 * origin[1];
    out[ORC_X] = -2.0 * origin[0];
    out[ORC_Z] = -2.0 * origin[2];
    out[ORC_Y] = (2.0 + 3.0 * origin[1]) * origin[1];
    out[ORC_C] = pow(origin[0], 2) + pow(origin[2], 2) - pow(origin[1], 2) * (1.0 + origin[1]);
}

/* SurfaceCoefs::clebsch, src/surface.cpp:41-52.  The reference assigns x3 twice and never assigns
 * z3, which therefore stays 0 -- reproduced, not fixed. */
void orc_surface_clebsch(double out[ORC_NCOEF])
{
    memset(out, 0, sizeof(double) * ORC_NCOEF);
    out[ORC_X3] = out[ORC_Y3] = 81.0; /* z3 deliberately left 0 */
    out[ORC_X2Y] = out[ORC_X2Z] = out[ORC_XY2] = out[ORC_Y2Z] = out[ORC_XZ2] = out[ORC_YZ2] = -189.0;
    out[ORC_XYZ] = 54.0;
    out[ORC_XY] = out[ORC_YZ] = out[ORC_XZ] = 126.0;
    out[ORC_X2] = out[ORC_Y2] = out[ORC_Z2] = -9.0;
    out[ORC_X] = out[ORC_Y] = out[ORC_Z] = 9.0;
    out[ORC_C] = 1.0;
}

/* SurfaceCoefs::cayley, src/surface.cpp:54-60 */
void orc_surface_cayley(double out[ORC_NCOEF])
{
    memset(out, 0, sizeof(double) * ORC_NCOEF);
    out[ORC_X2Y] = out[ORC_X2Z] = out[ORC_XY2] = out[ORC_Y2Z] = out[ORC_XZ2] = out[ORC_YZ2] = -5.0;
    out[ORC_XY] = out[ORC_YZ] = out[ORC_XZ] = 2.0;
}

/* LightSource::directional, src/light.cpp:4-14: colour pre-multiplied, direction normalised in FP64
 * and NEGATED */
void orc_light_directional(float intensity, const double dir[3], const float color[3], orc_light *out)
{
    double n[3];
    memset(out, 0, sizeof(*out));
    out->is_spherical = 0;
    for (int i = 0; i < 3; i++) out->color[i] = intensity * color[i];
    normalize3(dir, n);
    for (int i = 0; i < 3; i++) out->p[i] = -n[i];
}

/* LightSource::spherical, src/light.cpp:16-26 */
void orc_light_spherical(float intensity, const double pos[3], const float color[3], orc_light *out)
{
    memset(out, 0, sizeof(*out));
    out->is_spherical = 1;
    for (int i = 0; i < 3; i++) out->color[i] = intensity * color[i];
    for (int i = 0; i < 3; i++) out->p[i] = pos[i];
}

/* ------------------------------------------------------------------------------------------------
 * intersect_ray -- include/surface_impl.h:21-155
 *
 * F(o + t d) = t3 t^3 + t2 t^2 + t1 t + t0.  The four coefficients are 20-term sums taken in the
 * order of SurfaceCoefs, each term "coef * (parenthesised monomial factor)" (surface_impl.h:25-41
 * define the factors, :44-103 the sums).  The helpers below are those factors; argument order is
 * significant for rounding and follows the reference's macro arguments literally.
 * ---------------------------------------------------------------------------------------------- */
#define X 0
#define Y 1
#define Z 2
/* d_a d_b d_c  (COEF_3) and o_a o_b o_c (COEF_0_3): (a*b)*c */
static inline double tri(const double v[3], int a, int b, int c) { return v[a] * v[b] * v[c]; }
/* d_a d_b (COEF_2), o_a o_b (COEF_0_2) */
static inline double duo(const double v[3], int a, int b) { return v[a] * v[b]; }
/* t^2 factor of (o_p + t d_p)^3 : 3 o d d  (COEF_2_3) */
static inline double cube_t2(const double o[3], const double d[3], int p) { return 3.0 * o[p] * d[p] * d[p]; }
/* t^1 factor of (o_p + t d_p)^3 : 3 o o d  (COEF_1_3) */
static inline double cube_t1(const double o[3], const double d[3], int p) { return 3.0 * o[p] * o[p] * d[p]; }
/* t^2 factor of (o_p + t d_p)^2 (o_q + t d_q)  (COEF_2_21) */
static inline double sqlin_t2(const double o[3], const double d[3], int p, int q)
{
    return d[p] * (d[p] * o[q] + 2.0 * o[p] * d[q]);
}
/* t^1 factor of (o_p + t d_p)^2 (o_q + t d_q)  (COEF_1_21) */
static inline double sqlin_t1(const double o[3], const double d[3], int p, int q)
{
    return o[p] * (o[p] * d[q] + 2.0 * d[p] * o[q]);
}
/* t^2 / t^1 factors of (o_x + t d_x)(o_y + t d_y)(o_z + t d_z)  (COEF_2_111 / COEF_1_111) */
static inline double xyz_t2(const double o[3], const double d[3])
{
    return d[X] * d[Y] * o[Z] + d[X] * o[Y] * d[Z] + o[X] * d[Y] * d[Z];
}
static inline double xyz_t1(const double o[3], const double d[3])
{
    return d[X] * o[Y] * o[Z] + o[X] * d[Y] * o[Z] + o[X] * o[Y] * d[Z];
}
/* t^1 factor of (o_p + t d_p)^2 : 2 o d  (COEF_1_2) */
static inline double sq_t1(const double o[3], const double d[3], int p) { return 2.0 * o[p] * d[p]; }
/* t^1 factor of (o_a + t d_a)(o_b + t d_b) : o_a d_b + d_a o_b  (COEF_1_11) */
static inline double cross_t1(const double o[3], const double d[3], int a, int b)
{
    return o[a] * d[b] + d[a] * o[b];
}

static void ray_poly(const double k[ORC_NCOEF], const double o[3], const double d[3], double t[4])
{
    /* surface_impl.h:44-53 */
    double t3 = k[ORC_X3] * tri(d, X, X, X);
    t3 += k[ORC_Y3] * tri(d, Y, Y, Y);
    t3 += k[ORC_Z3] * tri(d, Z, Z, Z);
    t3 += k[ORC_X2Y] * tri(d, X, X, Y);
    t3 += k[ORC_XY2] * tri(d, X, Y, Y);
    t3 += k[ORC_X2Z] * tri(d, X, X, Z);
    t3 += k[ORC_XZ2] * tri(d, X, Z, Z);
    t3 += k[ORC_Y2Z] * tri(d, Y, Y, Z);
    t3 += k[ORC_YZ2] * tri(d, Y, Z, Z);
    t3 += k[ORC_XYZ] * tri(d, X, Y, Z);
    /* surface_impl.h:54-69 */
    double t2 = k[ORC_X3] * cube_t2(o, d, X);
    t2 += k[ORC_Y3] * cube_t2(o, d, Y);
    t2 += k[ORC_Z3] * cube_t2(o, d, Z);
    t2 += k[ORC_X2Y] * sqlin_t2(o, d, X, Y);
    t2 += k[ORC_XY2] * sqlin_t2(o, d, Y, X);
    t2 += k[ORC_X2Z] * sqlin_t2(o, d, X, Z);
    t2 += k[ORC_XZ2] * sqlin_t2(o, d, Z, X);
    t2 += k[ORC_Y2Z] * sqlin_t2(o, d, Y, Z);
    t2 += k[ORC_YZ2] * sqlin_t2(o, d, Z, Y);
    t2 += k[ORC_XYZ] * xyz_t2(o, d);
    t2 += k[ORC_X2] * duo(d, X, X);
    t2 += k[ORC_Y2] * duo(d, Y, Y);
    t2 += k[ORC_Z2] * duo(d, Z, Z);
    t2 += k[ORC_XY] * duo(d, X, Y);
    t2 += k[ORC_XZ] * duo(d, X, Z);
    t2 += k[ORC_YZ] * duo(d, Y, Z);
    /* surface_impl.h:70-86 */
    double t1 = k[ORC_X3] * cube_t1(o, d, X);
    t1 += k[ORC_Y3] * cube_t1(o, d, Y);
    t1 += k[ORC_Z3] * cube_t1(o, d, Z);
    t1 += k[ORC_X2Y] * sqlin_t1(o, d, X, Y);
    t1 += k[ORC_XY2] * sqlin_t1(o, d, Y, X);
    t1 += k[ORC_X2Z] * sqlin_t1(o, d, X, Z);
    t1 += k[ORC_XZ2] * sqlin_t1(o, d, Z, X);
    t1 += k[ORC_Y2Z] * sqlin_t1(o, d, Y, Z);
    t1 += k[ORC_YZ2] * sqlin_t1(o, d, Z, Y);
    t1 += k[ORC_XYZ] * xyz_t1(o, d);
    t1 += k[ORC_X2] * sq_t1(o, d, X);
    t1 += k[ORC_Y2] * sq_t1(o, d, Y);
    t1 += k[ORC_Z2] * sq_t1(o, d, Z);
    t1 += k[ORC_XY] * cross_t1(o, d, X, Y);
    t1 += k[ORC_XZ] * cross_t1(o, d, X, Z);
    t1 += k[ORC_YZ] * cross_t1(o, d, Y, Z);
    t1 += k[ORC_X] * d[X];
    t1 += k[ORC_Y] * d[Y];
    t1 += k[ORC_Z] * d[Z];
    /* surface_impl.h:87-103 */
    double t0 = k[ORC_X3] * tri(o, X, X, X);
    t0 += k[ORC_Y3] * tri(o, Y, Y, Y);
    t0 += k[ORC_Z3] * tri(o, Z, Z, Z);
    t0 += k[ORC_X2Y] * tri(o, X, X, Y);
    t0 += k[ORC_XY2] * tri(o, X, Y, Y);
    t0 += k[ORC_X2Z] * tri(o, X, X, Z);
    t0 += k[ORC_XZ2] * tri(o, X, Z, Z);
    t0 += k[ORC_Y2Z] * tri(o, Y, Y, Z);
    t0 += k[ORC_YZ2] * tri(o, Y, Z, Z);
    t0 += k[ORC_XYZ] * tri(o, X, Y, Z);
    t0 += k[ORC_X2] * duo(o, X, X);
    t0 += k[ORC_Y2] * duo(o, Y, Y);
    t0 += k[ORC_Z2] * duo(o, Z, Z);
    t0 += k[ORC_XY] * duo(o, X, Y);
    t0 += k[ORC_XZ] * duo(o, X, Z);
    t0 += k[ORC_YZ] * duo(o, Y, Z);
    t0 += k[ORC_X] * o[X];
    t0 += k[ORC_Y] * o[Y];
    t0 += k[ORC_Z] * o[Z];
    t0 += k[ORC_C];
    t[3] = t3;
    t[2] = t2;
    t[1] = t1;
    t[0] = t0;
}
#undef X
#undef Y
#undef Z

/* root selection, surface_impl.h:105-155 */
static double solve_poly(double t3, double t2, double t1, double t0, int *branch)
{
    if (fabs(t3) > K_EPS) {
        /* cubic, :106-136 */
        t2 /= t3;
        t1 /= t3;
        t0 /= t3;
        double q = (3.0 * t1 - t2 * t2) / 9.0;
        double r = (9.0 * t2 * t1 - 27.0 * t0 - 2.0 * t2 * t2 * t2) / 54.0;
        double delta = q * q * q + r * r;
        if (delta > 0) {
            /* one real root: Cardano, :113-118 -- not filtered, may be negative */
            *branch = 4;
            delta = sqrt(delta);
            q = cbrt(r + delta);
            r = cbrt(r - delta);
            return q + r - t2 / 3.0;
        }
        /* three real roots: trigonometric form, :120-133 */
        *branch = 5;
        double theta = acos(r / sqrt(-q * q * q)) / 3.0;
        double c = 2.0 * sqrt(-q);
        double x = c * cos(theta) - t2 / 3.0;
        double x1 = c * cos(theta + K_TWO_THIRD_PI) - t2 / 3.0;
        if (x1 >= K_EPS && x1 < x) x = x1;
        x1 = c * cos(theta + 2.0 * K_TWO_THIRD_PI) - t2 / 3.0;
        if (x1 >= K_EPS && x1 < x) x = x1;
        return x;
    }
    if (fabs(t2) > K_EPS) {
        /* quadratic, :138-149 -- first candidate is (-t1 - sqrt)/(2 t2) whatever the sign of t2 */
        double delta = t1 * t1 - 4.0 * t2 * t0;
        if (delta < 0) {
            *branch = 2;
            return -1.0;
        }
        *branch = 3;
        delta = sqrt(delta);
        double x = (-t1 - delta) / (2.0 * t2);
        if (x >= K_EPS) return x;
        return (-t1 + delta) / (2.0 * t2);
    }
    if (fabs(t1) > K_EPS) {
        /* linear, :150-153 */
        *branch = 1;
        return -t0 / t1;
    }
    *branch = 0;
    return -1.0;
}

double orc_intersect_ray_ex(const double coef[ORC_NCOEF], const double origin[3], const double dir[3],
                            double tcoef[4], int *branch)
{
    double t[4];
    int br = 0;
    ray_poly(coef, origin, dir, t);
    if (tcoef) memcpy(tcoef, t, sizeof(t));
    double r = solve_poly(t[3], t[2], t[1], t[0], &br);
    if (branch) *branch = br;
    return r;
}

double orc_intersect_ray(const double coef[ORC_NCOEF], const double origin[3], const double dir[3])
{
    return orc_intersect_ray_ex(coef, origin, dir, NULL, NULL);
}

/* normal_vector, include/surface_impl.h:157-172: normalised gradient, never flipped (Q8) */
void orc_normal_vector(const double k[ORC_NCOEF], const double p[3], double out[3])
{
    double g[3];
    const double k3[3] = {k[ORC_X3], k[ORC_Y3], k[ORC_Z3]};
    const double k2[3] = {k[ORC_X2], k[ORC_Y2], k[ORC_Z2]};
    const double k1[3] = {k[ORC_X], k[ORC_Y], k[ORC_Z]};
    /* 3.0 * vec(k3) * p * p + 2.0 * vec(k2) * p + vec(k1), component-wise, left to right */
    for (int i = 0; i < 3; i++) {
        g[i] = ((3.0 * k3[i]) * p[i]) * p[i] + (2.0 * k2[i]) * p[i] + k1[i];
    }
    g[0] += 2.0 * p[0] * (k[ORC_X2Y] * p[1] + k[ORC_X2Z] * p[2])
            + p[1] * (k[ORC_XY2] * p[1] + k[ORC_XYZ] * p[2] + k[ORC_XY])
            + p[2] * (k[ORC_XZ2] * p[2] + k[ORC_XZ]);
    g[1] += 2.0 * p[1] * (k[ORC_XY2] * p[0] + k[ORC_Y2Z] * p[2])
            + p[0] * (k[ORC_X2Y] * p[0] + k[ORC_XYZ] * p[2] + k[ORC_XY])
            + p[2] * (k[ORC_YZ2] * p[2] + k[ORC_YZ]);
    g[2] += 2.0 * p[2] * (k[ORC_XZ2] * p[0] + k[ORC_YZ2] * p[1])
            + p[0] * (k[ORC_X2Z] * p[0] + k[ORC_XYZ] * p[1] + k[ORC_XZ])
            + p[1] * (k[ORC_Y2Z] * p[1] + k[ORC_YZ]);
    normalize3(g, out);
}

/* shadow_ray, include/light_impl.h:17-27.  Returns a FLOAT vector (Q10): the direction is rounded to
 * FP32 here and widened again by the caller. */
void orc_shadow_ray(const orc_light *light, const double sp[3], float out_dir[3], double *max_t)
{
    if (light->is_spherical) {
        *max_t = 1.0f;
        for (int i = 0; i < 3; i++) out_dir[i] = (float) (light->p[i] - sp[i]);
    } else {
        *max_t = 1e6;
        for (int i = 0; i < 3; i++) out_dir[i] = (float) light->p[i];
    }
}

/* surface_color, include/light_impl.h:29-44 (Q11): Lambert, inverse-square for point lights */
void orc_surface_color(const orc_light *light, const double point[3], const double norm[3],
                       const float object_color[3], float out[3])
{
    double dir[3];
    float color[3];
    if (light->is_spherical) {
        for (int i = 0; i < 3; i++) dir[i] = light->p[i] - point[i];
        float denom = 4.0f * K_PI_F * (float) dot3(dir, dir); /* glm::length2 */
        for (int i = 0; i < 3; i++) color[i] = light->color[i] / denom;
        double n[3];
        normalize3(dir, n);
        for (int i = 0; i < 3; i++) dir[i] = n[i];
    } else {
        for (int i = 0; i < 3; i++) dir[i] = light->p[i];
        for (int i = 0; i < 3; i++) color[i] = light->color[i];
    }
    float lambert = (float) dot3(norm, dir);
    /* glm::max(0.0f, x) = (0.0f < x) ? x : 0.0f */
    float m = (0.0f < lambert) ? lambert : 0.0f;
    for (int i = 0; i < 3; i++) out[i] = object_color[i] / K_PI_F * color[i] * m;
}

/* reflect_ray, include/light_impl.h:46-49 */
void orc_reflect_ray(const double dir[3], const double normal[3], double out[3])
{
    double s = 2.0 * dot3(dir, normal);
    for (int i = 0; i < 3; i++) out[i] = dir[i] - s * normal[i];
}

/* ------------------------------------------------------------------------------------------------
 * Frame -- src/update-cpu.cpp
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    const orc_scene *scene;
    const double *cam;
    double aspect, tan_half_fov; /* init_update, update-cpu.cpp:27-28 */
    double ray_origin[3];        /* update, update-cpu.cpp:123 */
} frame_ctx;

static void frame_setup(frame_ctx *f, const orc_scene *scene, const double cam[16])
{
    static const double RAY_ORIGIN[4] = {0.0, 0.0, 0.0, 1.0};
    double o4[4];
    f->scene = scene;
    f->cam = cam;
    f->aspect = (double) scene->px_width / scene->px_height; /* Scene::aspect_ratio, scene.h:32-33 */
    f->tan_half_fov = tan(0.5 * scene->vertical_fov);
    mat4_mul_vec4(cam, RAY_ORIGIN, o4);
    f->ray_origin[0] = o4[0];
    f->ray_origin[1] = o4[1];
    f->ray_origin[2] = o4[2];
}

static inline void count_branch(orc_counters *cnt, int br)
{
    cnt->tests++;
    switch (br) {
    case 0: cnt->br_none++; break;
    case 1: cnt->br_linear++; break;
    case 2: cnt->br_quad_miss++; break;
    case 3: cnt->br_quad_hit++; break;
    case 4: cnt->br_cardano++; break;
    default: cnt->br_trig++; break;
    }
}

static inline double hit_test(const orc_object *obj, const double o[3], const double d[3], orc_counters *cnt)
{
    if (cnt) {
        int br;
        double t = orc_intersect_ray_ex(obj->c, o, d, NULL, &br);
        count_branch(cnt, br);
        return t;
    }
    return orc_intersect_ray(obj->c, o, d);
}

/* get_color_and_object, update-cpu.cpp:45-80 (Q7, Q12) */
static int trace(const frame_ctx *f, const double origin[3], const double dir[3], float result[3],
                 double sp[3], double sn[3], orc_counters *cnt)
{
    const orc_scene *s = f->scene;
    int best = -1;
    double best_t = INFINITY;
    for (int i = 0; i < (int) s->n_objects; i++) {
        double t = hit_test(&s->objects[i], origin, dir, cnt);
        if (t >= K_EPS && t < K_MAX_T && t < best_t) {
            best_t = t;
            best = i;
        }
    }
    if (best < 0) return -1;

    result[0] = result[1] = result[2] = 0.0f;
    for (int i = 0; i < 3; i++) sp[i] = origin[i] + best_t * dir[i];
    orc_normal_vector(s->objects[best].c, sp, sn);
    if (cnt) cnt->normals++;
    const float *albedo = s->objects[best].color;
    for (uint32_t l = 0; l < s->n_lights; l++) {
        const orc_light *light = &s->lights[l];
        double max_t = 0;
        float sdir_f[3];
        orc_shadow_ray(light, sp, sdir_f, &max_t);
        if (cnt) cnt->shadow_rays++;
        const double sdir[3] = {sdir_f[0], sdir_f[1], sdir_f[2]};
        double so[3];
        for (int i = 0; i < 3; i++) so[i] = sp[i] + K_SHADOW_BIAS * sn[i];
        int in_shadow = 0;
        for (uint32_t k = 0; k < s->n_objects; k++) {
            double t = hit_test(&s->objects[k], so, sdir, cnt);
            if (t > K_EPS && t < max_t) {
                in_shadow = 1;
                break;
            }
        }
        if (!in_shadow) {
            float c[3];
            orc_surface_color(light, sp, sn, albedo, c);
            if (cnt) cnt->surface_colors++;
            for (int i = 0; i < 3; i++) result[i] += c[i];
        }
    }
    /* glm::min(vec3(1.0f), result) = (result < 1) ? result : 1 */
    for (int i = 0; i < 3; i++) result[i] = (result[i] < 1.0f) ? result[i] : 1.0f;
    return best;
}

static void primary_dir(const frame_ctx *f, int x, int y, double dir[3])
{
    /* update-cpu.cpp:84-89 (Q1) */
    const orc_scene *s = f->scene;
    double ndc_x = (x + 0.5) / (int) s->px_width;
    double ndc_y = (y + 0.5) / (int) s->px_height;
    double camera_x = (2.0 * ndc_x - 1.0) * f->aspect * f->tan_half_fov;
    double camera_y = (2.0 * ndc_y - 1.0) * f->tan_half_fov;
    const double v[4] = {camera_x, camera_y, 1.0, 1.0};
    double w[4], rel[3];
    mat4_mul_vec4(f->cam, v, w);
    for (int i = 0; i < 3; i++) rel[i] = w[i] - f->ray_origin[i];
    normalize3(rel, dir);
}

static void render_pixel(const frame_ctx *f, int x, int y, float out[3], orc_counters *cnt)
{
    const orc_scene *s = f->scene;
    double dir[3], sp[3], sn[3];
    float oc[3], res[3];
    primary_dir(f, x, y, dir);
    if (cnt) cnt->primary_rays++;
    int idx = trace(f, f->ray_origin, dir, oc, sp, sn, cnt);
    if (idx < 0) {
        for (int i = 0; i < 3; i++) out[i] = s->bg_color[i];
        return;
    }
    for (int i = 0; i < 3; i++) res[i] = oc[i];
    /* reflection loop, update-cpu.cpp:96-117 (Q13) */
    float cur_ratio = 1.0f;
    int cur_reflections = 0;
    while (s->objects[idx].reflection_ratio > K_EPS) {
        cur_ratio *= s->objects[idx].reflection_ratio;
        if (cur_reflections == (int) s->max_reflections) {
            for (int i = 0; i < 3; i++) res[i] = (1.0f - cur_ratio) * res[i] + cur_ratio * s->bg_color[i];
            break;
        }
        cur_reflections++;
        double nd[3], no[3];
        orc_reflect_ray(dir, sn, nd);
        if (cnt) cnt->reflect_rays++;
        for (int i = 0; i < 3; i++) dir[i] = nd[i];
        for (int i = 0; i < 3; i++) no[i] = sp[i] + K_SHADOW_BIAS * sn[i];
        idx = trace(f, no, dir, oc, sp, sn, cnt);
        if (idx < 0) {
            for (int i = 0; i < 3; i++) res[i] = (1.0f - cur_ratio) * res[i] + cur_ratio * s->bg_color[i];
            break;
        }
        for (int i = 0; i < 3; i++) res[i] = (1.0f - cur_ratio) * res[i] + cur_ratio * oc[i];
    }
    for (int i = 0; i < 3; i++) out[i] = res[i];
}

void orc_primary_dir(const orc_scene *scene, const double cam[16], int x, int y, double out_dir[3])
{
    frame_ctx f;
    frame_setup(&f, scene, cam);
    primary_dir(&f, x, y, out_dir);
}

void orc_render_pixel(const orc_scene *scene, const double cam[16], int x, int y, float out_rgb[3],
                      orc_counters *cnt)
{
    frame_ctx f;
    frame_setup(&f, scene, cam);
    render_pixel(&f, x, y, out_rgb, cnt);
}

typedef struct {
    const frame_ctx *f;
    const uint32_t *rows;
    uint32_t n_rows, first, step;
    float *out;
    orc_counters cnt;
    int want_cnt;
} row_job;

static void *row_worker(void *arg)
{
    row_job *j = (row_job *) arg;
    const uint32_t w = j->f->scene->px_width;
    for (uint32_t r = j->first; r < j->n_rows; r += j->step) {
        int y = j->rows ? (int) j->rows[r] : (int) r;
        float *dst = j->out + (size_t) r * w * 3;
        for (uint32_t x = 0; x < w; x++) render_pixel(j->f, (int) x, y, dst + 3 * (size_t) x, j->want_cnt ? &j->cnt : NULL);
    }
    return NULL;
}

static void add_counters(orc_counters *a, const orc_counters *b)
{
    uint64_t *pa = (uint64_t *) a;
    const uint64_t *pb = (const uint64_t *) b;
    for (size_t i = 0; i < sizeof(orc_counters) / sizeof(uint64_t); i++) pa[i] += pb[i];
}

/* update, update-cpu.cpp:121-133: serial y/x loop; nthreads > 1 interleaves rows over threads with
 * the same per-pixel arithmetic (for the all-cores baseline only). */
void orc_render_rows(const orc_scene *scene, const double cam[16], const uint32_t *rows, uint32_t n_rows,
                     float *out_rgb, orc_counters *cnt, int nthreads)
{
    frame_ctx f;
    frame_setup(&f, scene, cam);
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    row_job *jobs = (row_job *) calloc((size_t) nthreads, sizeof(row_job));
    pthread_t *th = (pthread_t *) calloc((size_t) nthreads, sizeof(pthread_t));
    for (int i = 0; i < nthreads; i++) {
        jobs[i].f = &f;
        jobs[i].rows = rows;
        jobs[i].n_rows = n_rows;
        jobs[i].first = (uint32_t) i;
        jobs[i].step = (uint32_t) nthreads;
        jobs[i].out = out_rgb;
        jobs[i].want_cnt = cnt != NULL;
    }
    if (nthreads == 1) {
        row_worker(&jobs[0]);
    } else {
        for (int i = 0; i < nthreads; i++) pthread_create(&th[i], NULL, row_worker, &jobs[i]);
        for (int i = 0; i < nthreads; i++) pthread_join(th[i], NULL);
    }
    if (cnt) {
        memset(cnt, 0, sizeof(*cnt));
        for (int i = 0; i < nthreads; i++) add_counters(cnt, &jobs[i].cnt);
    }
    free(jobs);
    free(th);
}

double orc_checksum(const float *rgb, uint64_t n_floats)
{
    double s = 0.0;
    for (uint64_t i = 0; i < n_floats; i++) s += rgb[i];
    return s;
}

/* ------------------------------------------------------------------------------------------------
 * Host camera -- src/ray-tracer.cpp:44-58: inverse(lookAt(position, position - direction, up)).
 * lookAt = glm::lookAtRH, inverse = glm's cofactor-expansion 4x4 inverse (restated from the published
 * glm algorithm; glm is not vendored by the reference).  Only produces INPUTS for update().
 * ---------------------------------------------------------------------------------------------- */
static void mat4_inverse(const double m[16], double out[16])
{
#define M(c, r) m[(c) * 4 + (r)]
    double c00 = M(2, 2) * M(3, 3) - M(3, 2) * M(2, 3);
    double c02 = M(1, 2) * M(3, 3) - M(3, 2) * M(1, 3);
    double c03 = M(1, 2) * M(2, 3) - M(2, 2) * M(1, 3);
    double c04 = M(2, 1) * M(3, 3) - M(3, 1) * M(2, 3);
    double c06 = M(1, 1) * M(3, 3) - M(3, 1) * M(1, 3);
    double c07 = M(1, 1) * M(2, 3) - M(2, 1) * M(1, 3);
    double c08 = M(2, 1) * M(3, 2) - M(3, 1) * M(2, 2);
    double c10 = M(1, 1) * M(3, 2) - M(3, 1) * M(1, 2);
    double c11 = M(1, 1) * M(2, 2) - M(2, 1) * M(1, 2);
    double c12 = M(2, 0) * M(3, 3) - M(3, 0) * M(2, 3);
    double c14 = M(1, 0) * M(3, 3) - M(3, 0) * M(1, 3);
    double c15 = M(1, 0) * M(2, 3) - M(2, 0) * M(1, 3);
    double c16 = M(2, 0) * M(3, 2) - M(3, 0) * M(2, 2);
    double c18 = M(1, 0) * M(3, 2) - M(3, 0) * M(1, 2);
    double c19 = M(1, 0) * M(2, 2) - M(2, 0) * M(1, 2);
    double c20 = M(2, 0) * M(3, 1) - M(3, 0) * M(2, 1);
    double c22 = M(1, 0) * M(3, 1) - M(3, 0) * M(1, 1);
    double c23 = M(1, 0) * M(2, 1) - M(2, 0) * M(1, 1);
    const double f0[4] = {c00, c00, c02, c03}, f1[4] = {c04, c04, c06, c07}, f2[4] = {c08, c08, c10, c11};
    const double f3[4] = {c12, c12, c14, c15}, f4[4] = {c16, c16, c18, c19}, f5[4] = {c20, c20, c22, c23};
    const double v0[4] = {M(1, 0), M(0, 0), M(0, 0), M(0, 0)}, v1[4] = {M(1, 1), M(0, 1), M(0, 1), M(0, 1)};
    const double v2[4] = {M(1, 2), M(0, 2), M(0, 2), M(0, 2)}, v3[4] = {M(1, 3), M(0, 3), M(0, 3), M(0, 3)};
    static const double sa[4] = {+1, -1, +1, -1}, sb[4] = {-1, +1, -1, +1};
    double inv[16];
    for (int i = 0; i < 4; i++) {
        inv[0 + i] = (v1[i] * f0[i] - v2[i] * f1[i] + v3[i] * f2[i]) * sa[i];
        inv[4 + i] = (v0[i] * f0[i] - v2[i] * f3[i] + v3[i] * f4[i]) * sb[i];
        inv[8 + i] = (v0[i] * f1[i] - v1[i] * f3[i] + v3[i] * f5[i]) * sa[i];
        inv[12 + i] = (v0[i] * f2[i] - v1[i] * f4[i] + v2[i] * f5[i]) * sb[i];
    }
    double d0 = M(0, 0) * inv[0], d1 = M(0, 1) * inv[4], d2 = M(0, 2) * inv[8], d3 = M(0, 3) * inv[12];
    double one_over_det = 1.0 / ((d0 + d1) + (d2 + d3));
    for (int i = 0; i < 16; i++) out[i] = inv[i] * one_over_det;
#undef M
}

void orc_camera_matrix(const double pos[3], double yaw_deg, double pitch_deg, double out_cam[16])
{
    /* update_direction, src/ray-tracer.cpp:44-52 */
    double dir[3] = {cos(orc_radians(yaw_deg)) * cos(orc_radians(pitch_deg)), sin(orc_radians(pitch_deg)),
                     sin(orc_radians(yaw_deg)) * cos(orc_radians(pitch_deg))};
    const double up[3] = {0.0, 1.0, 0.0};
    /* camera_matrix, src/ray-tracer.cpp:54-58 */
    double center[3], fwd[3], tmp[3], s[3], u[3], view[16];
    for (int i = 0; i < 3; i++) center[i] = pos[i] - dir[i];
    for (int i = 0; i < 3; i++) tmp[i] = center[i] - pos[i];
    normalize3(tmp, fwd);
    cross3(fwd, up, tmp);
    normalize3(tmp, s);
    cross3(s, fwd, u);
    memset(view, 0, sizeof(view));
    view[15] = 1.0;
    view[0] = s[0]; view[4] = s[1]; view[8] = s[2];
    view[1] = u[0]; view[5] = u[1]; view[9] = u[2];
    view[2] = -fwd[0]; view[6] = -fwd[1]; view[10] = -fwd[2];
    view[12] = -dot3(s, pos);
    view[13] = -dot3(u, pos);
    view[14] = dot3(fwd, pos);
    mat4_inverse(view, out_cam);
}
