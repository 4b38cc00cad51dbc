// rt_scene_dev.h -- scene records as they sit in HBM / LDS, shared by the host packer (rt_capi.cpp)
// and the kernels (rt_kernels.hip).
//
// The reference keeps an array of Object (176 B: SurfaceCoefs + reflection_ratio + color,
// include/scene.h:8-15) and LightSource (48 B, include/light.h:6-13) in global memory
// (src/update-cuda.cu:42-48).  Here each record is padded to a multiple of 16 bytes so that the whole
// scene block can be staged into LDS with 16-byte copies and gathered per lane with ds_read_b128 /
// ds_read_b64, and every object carries a class word computed once at rt_create.
#ifndef RT_SCENE_DEV_H
#define RT_SCENE_DEV_H

#include <stdint.h>

// coefficient indices, order of SurfaceCoefs (include/surface.h:10-15)
enum {
    K_X3 = 0, K_Y3, K_Z3, K_X2Y, K_XY2, K_X2Z, K_XZ2, K_Y2Z, K_YZ2, K_XYZ,
    K_X2, K_Y2, K_Z2, K_XY, K_XZ, K_YZ,
    K_X, K_Y, K_Z, K_C
};

// Object classes.  A coefficient that is exactly 0 contributes an exact (signed) zero to the
// reference's 20-term sums (include/surface_impl.h:44-103), so leaving its term out does not change
// t3..t0 except possibly the sign of a zero, which no comparison of the solver can see.  The class
// says which groups of terms are present; the branch on it is wave-uniform.
#define RT_CLS_CUBIC 1u   // some degree-3 coefficient != 0  -> dense 20-term expansion + full solver
#define RT_CLS_SQUARE 2u  // some of x2, y2, z2 != 0
#define RT_CLS_CROSS 4u   // some of xy, xz, yz != 0
#define RT_CLS_UNITSQ 8u  // x2 == y2 == z2 == 1 and no cross terms (a sphere): 1.0 * m == m exactly, so the
                          // squared part of t2, t1, t0 is one per-ray sum shared by all such objects

struct alignas(16) DevObject {
    double c[20];     // 160 B
    float albedo[3];  // Object::color
    float refl;       // Object::reflection_ratio
    uint32_t cls;     // RT_CLS_*
    uint32_t pad[3];
};                    // 192 B
static_assert(sizeof(DevObject) == 192, "DevObject layout");

struct alignas(16) DevLight {
    double p[3];        // LightSource::p
    float color[3];     // LightSource::light_color
    uint32_t spherical; // LightSource::is_spherical
    uint32_t pad[2];
};                      // 48 B
static_assert(sizeof(DevLight) == 48, "DevLight layout");

// Per-launch constants (kernel argument, lands in SGPRs).
struct FrameArgs {
    double cam[16];      // camera-to-world, column-major
    double origin[3];    // vec3(cam * (0,0,0,1)), src/update-cpu.cpp:123
    double aspect;       // (double) W / H, include/scene.h:32-33
    double tan_half_fov; // tan(0.5 * vertical_fov), src/update-cpu.cpp:28
    float bg[4];
    uint32_t width, height;
    uint32_t n_obj, n_lights;
    uint32_t max_refl;
    uint32_t rank, world, band_rows;
    uint32_t local_rows;
    uint32_t tiles_x;    // number of 16-pixel tile columns
};

#define RT_TILE 16        // a workgroup renders a 16 x 16 pixel tile: 4 waves of 8 x 8
#define RT_MAX_LDS_SCENE (96u * 1024u)

#endif
