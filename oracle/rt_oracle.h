/*
 * rt_oracle.h -- CPU ORACLE for the per-pixel ray-tracing hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may build, load or call it, and only as
 * the checker / the timed CPU baseline.  The product path (cuda-ray-tracer_amd/) never links it.
 *
 * What it is: a plain-C restatement of the reference's CPU back end
 *     /root/reference/src/update-cpu.cpp            (frame loop, nearest hit, shadows, reflection loop)
 *     /root/reference/include/surface_impl.h        (intersect_ray, normal_vector)
 *     /root/reference/include/light_impl.h          (shadow_ray, surface_color, reflect_ray)
 *     /root/reference/src/surface.cpp, src/light.cpp (surface / light factories)
 * with the glm vector operations it relies on (dot, normalize, mat4*vec4, min, max, radians) written
 * out explicitly.  Every function cites the reference file:line it follows.
 *
 * PARITY UNPINNED (formally): the reference ships no tests, golden vectors or rendered outputs, and
 * its CPU path cannot be built in this image (glm, GLFW and yaml-cpp are absent and un-vendored; a
 * build against stand-in headers is not allowed).  The only cross-checks available are the
 * survey-time anchors recorded in SURVEY.md section 8(c)/8 work table (frame checksums, sample pixels, ray and
 * intersection-test counts); tests/test_oracle_anchors.py asserts the oracle reproduces all of them.
 * glm itself is an unpinned third-party dependency of the reference: the explicit operation order
 * used here ("third-party arithmetic") DEFINES parity at that boundary.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off, no -ffast-math: the reference's x86-64
 * default has no FMA contraction).
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* index of each coefficient inside orc_object.c[] -- order of SurfaceCoefs, include/surface.h:10-15 */
enum {
    ORC_X3 = 0, ORC_Y3, ORC_Z3, ORC_X2Y, ORC_XY2, ORC_X2Z, ORC_XZ2, ORC_Y2Z, ORC_YZ2, ORC_XYZ,
    ORC_X2, ORC_Y2, ORC_Z2, ORC_XY, ORC_XZ, ORC_YZ,
    ORC_X, ORC_Y, ORC_Z, ORC_C,
    ORC_NCOEF
};

/* Object, include/scene.h:8-15 */
typedef struct {
    double c[ORC_NCOEF];
    float reflection_ratio;
    float color[3];
} orc_object;

/* LightSource, include/light.h:6-13 (light_color already multiplied by intensity, src/light.cpp:11,23) */
typedef struct {
    int32_t is_spherical;
    int32_t pad_;
    double p[3];
    float color[3];
    float pad2_;
} orc_light;

/* Scene, include/scene.h:17-36 -- vertical_fov in RADIANS as in the reference */
typedef struct {
    uint32_t px_width, px_height;
    double vertical_fov;
    float bg_color[3];
    uint32_t max_reflections;
    uint32_t n_objects, n_lights;
    const orc_object *objects;
    const orc_light *lights;
} orc_scene;

/* Work counters (SURVEY.md section 8 work table / 8(d) ray definition). */
typedef struct {
    uint64_t primary_rays, shadow_rays, reflect_rays;
    uint64_t tests;                 /* intersect_ray calls */
    uint64_t br_cardano, br_trig, br_quad_hit, br_quad_miss, br_linear, br_none;
    uint64_t normals, surface_colors;
} orc_counters;

/* ---- factories: src/surface.cpp, src/light.cpp, src/scene.cpp:20 ---- */
double orc_radians(double deg);
void orc_surface_sphere(const double center[3], double radius, double out[ORC_NCOEF]);
void orc_surface_plane(const double origin[3], const double nv[3], double out[ORC_NCOEF]);
void orc_surface_dingdong(const double origin[3], double out[ORC_NCOEF]);
void orc_surface_clebsch(double out[ORC_NCOEF]);
void orc_surface_cayley(double out[ORC_NCOEF]);
void orc_light_directional(float intensity, const double dir[3], const float color[3], orc_light *out);
void orc_light_spherical(float intensity, const double pos[3], const float color[3], orc_light *out);

/* ---- per-ray math: include/surface_impl.h, include/light_impl.h ---- */
double orc_intersect_ray(const double coef[ORC_NCOEF], const double origin[3], const double dir[3]);
/* as above, also reports t3..t0 before the solver touches them and the branch taken (0 none,1 linear,
 * 2 quad miss,3 quad hit,4 cardano,5 trig) -- used by the unit-tier tests */
double orc_intersect_ray_ex(const double coef[ORC_NCOEF], const double origin[3], const double dir[3],
                            double tcoef[4], int *branch);
void orc_normal_vector(const double coef[ORC_NCOEF], const double pos[3], double out[3]);
void orc_shadow_ray(const orc_light *light, const double surface_point[3], float out_dir[3], double *max_t);
void orc_surface_color(const orc_light *light, const double point[3], const double norm[3],
                       const float object_color[3], float out[3]);
void orc_reflect_ray(const double dir[3], const double normal[3], double out[3]);

/* ---- frame: src/update-cpu.cpp ---- */
/* Primary-ray direction of pixel (x, y), update-cpu.cpp:82-89 (camera column-major, 16 doubles). */
void orc_primary_dir(const orc_scene *scene, const double cam[16], int x, int y, double out_dir[3]);
/* render_pixel, update-cpu.cpp:82-119 */
void orc_render_pixel(const orc_scene *scene, const double cam[16], int x, int y, float out_rgb[3],
                      orc_counters *cnt);
/* Rows rows[0..n_rows) (global y, row 0 = bottom) -> out_rgb[n_rows][W][3] float, update-cpu.cpp:121-133.
 * rows == NULL means rows 0..n_rows-1.  cnt may be NULL.  nthreads <= 1: serial like the reference. */
void orc_render_rows(const orc_scene *scene, const double cam[16], const uint32_t *rows, uint32_t n_rows,
                     float *out_rgb, orc_counters *cnt, int nthreads);
/* Sum of all channels accumulated in double (SURVEY.md section 8 "checksum"). */
double orc_checksum(const float *rgb, uint64_t n_floats);

/* ---- host camera (src/ray-tracer.cpp:44-58): inverse(lookAt(pos, pos - dir, up)) ---- */
void orc_camera_matrix(const double pos[3], double yaw_deg, double pitch_deg, double out_cam[16]);

#ifdef __cplusplus
}
#endif
#endif /* RT_ORACLE_H */
