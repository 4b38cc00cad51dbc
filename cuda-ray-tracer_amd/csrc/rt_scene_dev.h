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

#include <stddef.h>
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
    // Bounding sphere for the conservative shadow-ray culling of the wavefront kernel (rt_wavefront.hip):
    // centre and radius of the sphere itself for RT_CLS_UNITSQ objects with a real radius, radius = +inf
    // ("always test") for everything else.  Derived data, never used to compute a pixel.
    double bs_center[3];
    double bs_radius;
};                    // 224 B
static_assert(sizeof(DevObject) == 224, "DevObject layout");

// Per-class packed tables for the wavefront kernel's wave-uniform loops (built by rt_create next to the
// DevObject array, staged into LDS with it).  One entry = exactly the coefficients that class's test reads,
// so a loop iteration is: one unconditional 16-byte-aligned LDS broadcast read, then arithmetic -- no class
// word to wait for, no conditional loads.  `orig` is the object's index in the scene (tie-breaking, hit
// records, counters).
struct alignas(16) UsEntry {  // RT_CLS_UNITSQ: x2 = y2 = z2 = 1, no cross terms (spheres)
    double kx, ky, kz, c;     // K_X, K_Y, K_Z, K_C
    double r, inv_r;          // bounding radius (+inf: do not cull) and 1 / r  -- culling only
    uint32_t orig;
    // Own-sphere rule of the lean path (rt_wavefront.hip, own_sphere_skippable): a shadow ray that leaves this sphere towards a
    // directional light in front of the surface cannot be blocked by this very sphere while the reference's own t0 of that test
    // lies in (own_lo, own_hi).  own_lo = +inf: never skipped.  Derived data, set by rt_create.
    float own_lo, own_hi;
    uint32_t pad;
};                            // 64 B
struct alignas(16) LinEntry { // no degree-2 terms (planes)
    double kx, ky, kz, c;
    uint32_t orig, pad[3];
};                            // 48 B
struct alignas(16) GqEntry {  // any other surface of degree <= 2
    double x2, y2, z2, xy, xz, yz, kx, ky, kz, c;
    uint32_t orig, pad[3];
};                            // 96 B
struct alignas(16) MatEntry { // per object, scene order: what shading and the bounce decision read
    float albedo[3];          // Object::color
    float refl;               // Object::reflection_ratio
};                            // 16 B
static_assert(sizeof(UsEntry) == 64 && sizeof(LinEntry) == 48 && sizeof(GqEntry) == 96 && sizeof(MatEntry) == 16, "table layout");

struct alignas(16) DevLight {
    double p[3];        // LightSource::p
    float color[3];     // LightSource::light_color
    uint32_t spherical; // LightSource::is_spherical
    // Derived at rt_create for the wavefront kernel.  A directional light's shadow rays all use the SAME
    // direction, the FP32-rounded p (include/light_impl.h:17-27), so its monomials are per-light constants.
    double sdir[3];                      // (double) (float) p
    double dxx, dyy, dzz, dxy, dxz, dyz; // products of sdir components (COEF_2 of surface_impl.h:28)
    double u2;                           // (dxx + dyy) + dzz
    double inv_uu, len_u;                // 1 / |sdir|^2 and 1.001 |sdir|: culling only
    // 1: a hit that faces away from this (directional) light may skip it altogether, shadow test and shading: its term is
    // ((albedo / pi) * colour) * max(0, n.l) = finite * 0 = +0, and accumulating +0 changes nothing (include/light_impl.h:43,
    // src/update-cpu.cpp:74).  Set by rt_create when this light's colour and every albedo of the scene are finite.
    uint32_t backface_exact;
    uint32_t pad_;
};                                       // 144 B
static_assert(sizeof(DevLight) == 144, "DevLight layout");

// One light as the lean path of the wavefront kernel reads it: through the constant address space, all of it in one batch of scalar
// loads, so that a directional light's constants live in SGPRs and never in a vector register.  The table follows the DevLight
// array in the same allocation ([DevLight x n][LightK x n]).
struct alignas(64) LightK {
    double p[3];        // LightSource::p
    double sdir[3];     // (double) (float) p: a directional light's shadow-ray direction (include/light_impl.h:23-25)
    double u2;          // |sdir|^2 as the reference sums it
    double inv_uu, len_u; // culling only (DevLight)
    double four_u2;     // 4.0 * u2: the factor of t0 in the discriminant (include/surface_impl.h:139)
    double s_yz, s_xz, s_xy; // |sdir.y| + |sdir.z|, |sdir.x| + |sdir.z|, |sdir.x| + |sdir.y|: the box stage of the culling
    float color[3];     // LightSource::light_color
    uint32_t flags;     // 1 spherical   2 backface_exact (DevLight)   4 |sdir|^2 > EPS (the reference solves a quadratic for this light's rays)
    uint32_t pad[2];
};                      // 128 B: two 64-byte scalar loads
static_assert(sizeof(LightK) == 128, "LightK layout");

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
    uint32_t cull;       // wavefront kernel: 1 = cull shadow tests against per-chunk bounding volumes
    // scene blob = [DevObject x n_obj][UsEntry x n_us][GqEntry x n_gq][LinEntry x n_lin][uint32 x n_cub][MatEntry x n_obj]
    // Only the part from off_us on (class tables + materials: 64-96 B per object) is ever staged into LDS; the
    // 224-byte object records stay in global memory and are read per hit (normals) only.
    uint32_t n_us, n_gq, n_lin, n_cub;
    uint32_t off_us, off_gq, off_lin, off_cub, off_mat; // byte offsets into the blob
    uint32_t scene_bytes;                                // blob size, multiple of 16
    uint32_t stage_bytes;                                // scene_bytes - off_us
    uint32_t n_tiles;                           // tiles of this rank's rows (tiles_x * tiles_y)
    uint32_t rgba8;                             // 1: store iround(c*255) RGBA8 instead of RGBA32F
    uint32_t has_mirror;                        // some object has reflection_ratio > EPS
    uint32_t all_cullable;                      // every object is a unit sphere with a finite bounding radius
    // Tile-level early-out (all_cullable scenes): the rays of a tile are t * M3 * (cx, cy, 1), cx / cy between the tile's
    // first and last pixel, so they lie inside the pyramid of the planes (M3^-T n) . p >= 0 with n = (1, 0, -cx0),
    // (-1, 0, cx1), (0, 1, -cy0), (0, -1, cy1), (0, 0, 1).  Set per frame by rt_render:
    double tile_nt[9];   // M3^-T, column-major (M3 = upper-left 3x3 of cam)
    double cx_a, cx_b;   // camera-plane x of pixel column x  ~=  cx_a * x + cx_b   (cx_a > 0)
    double cy_a, cy_b;   // camera-plane y of image row y     ~=  cy_a * y + cy_b   (cy_a > 0)
    uint32_t tile_planes_ok; // 0: M3 is singular / not finite -- no tile is declared empty this frame
    // Sparse output (rt_render_sparse, RGBA8): the framebuffer argument is a message in rt_pack_sparse's layout; tiles with
    // round-0 hits take a slot each, background tiles are not stored.
    uint32_t sparse, sparse_cap;
    // Launch-order feedback (wavefront kernel): three generations, ord_stride words apart, of
    //   { count[16] (cost classes, 0 = the costliest), census, largest cost, pad[14], word[n_tiles], list[16][n_tiles] }     (uint32)
    // word[t] = (position in its class list << 5) | (class + 1), written only by tiles with hits (a stale word is harmless: the
    // reader checks that the list entry it points at names tile t).  Frame k reads what frame k-1 wrote, writes its
    // own generation and clears the counters frame k+1 will append to.  NULL = tiles run in index order.
    uint32_t *order_state;
    uint32_t *ord_host;   // host-mapped {listed tiles, census} of the previous frame: sizes / switches later launches
    uint32_t ord_stride, ord_read, ord_write, ord_zero;
    uint32_t ord_cap;     // list slots in this launch (the grid is n_scan + ord_cap + n_tiles workgroups)
    uint32_t ord_on;      // 0: index order this frame, census only
    uint32_t ord_plain;   // 1: RT_FLAG_PLAIN_ORDER (A/B): list slot q renders list entry q instead of the balanced (snake) assignment
    uint32_t ord_frame;   // this frame's number (never 0): order_state[3 * ord_stride + t] remembers the last frame in which a half-tile workgroup
                          // entered tile t into the lists, so that exactly one of the two does
    uint32_t ord_split;   // the tiles of this many of the costliest cost classes (0..15) are rendered by two workgroups, one per half tile
                          // (rt_wavefront.hip, "half tiles"); 0 with sparse output (a tile is one message slot) and RT_FLAG_NOSPLIT
    // Tile words (all_cullable scenes; rt_wavefront.hip, "tile words"): the first n_scan workgroups of the grid classify
    // RT_SCAN_TILES tiles each and publish tile_state[t] = (frame_tag << 3) | EMPTY / NONEMPTY / COVERED, paint workgroups paint the
    // EMPTY ones, and the workgroup that gets tile t in index order leaves at once when the word says EMPTY.
    uint32_t *tile_state; // [n_tiles], NULL = no scan workgroups
    uint32_t frame_tag;   // unique per frame of this context, never 0, < 2^29
    uint32_t n_scan;      // classifying workgroups in this launch
    uint32_t lean;        // 1: all-sphere scene without mirrors, dense output: the wave-per-block instantiation renders it (rt_wavefront.hip, "the lean path")
    // Degree-3 surfaces: the record (rtm::CubicAt -- F, grad F, half Hessian: ten doubles) of the first RT_CUB_AT_MAX of them at the ray origin of the frame -- constants of all primary rays, set per frame by rt_render
    // (further objects and other origins are evaluated by the kernel) -- and their rtm::CubicAbs (per object)
    double cub_rec[4][10];
    double cub_abs[4][4];   // (directly behind cub_rec: the kernel copies both to LDS in one sweep)
    uint32_t lights_plain; // lean path: every directional light has flags 2 and 4 of LightK (finite colours, |sdir|^2 > EPS): the specialised light loop applies
    uint32_t pt_mask[2];  // lean path: bit l = light l is a point light (at most 64 lights there; scenes with more take the general instantiation)
};

#define RT_CUB_AT_MAX 4    // FrameArgs::cub_rec
#define RT_CUB_REC 10      // doubles per record (rtm::CubicAt)
static_assert(offsetof(FrameArgs, cub_abs) == offsetof(FrameArgs, cub_rec) + sizeof(double) * 4 * RT_CUB_REC, "cub_abs directly behind cub_rec");
#define RT_TILE 16        // a workgroup renders a 16 x 16 pixel tile
#define RT_SCAN_TILES 64  // tiles classified by one classifying workgroup: sixteen per wave
#define RT_PAINT_TILES 16 // tiles one paint workgroup is responsible for: four per wave
#define RT_ORD_SPLIT_CLASSES 8u /* default FrameArgs::ord_split: tiles that cost more than half of the costliest one */
#define RT_ORD_HDR 32      // words before word[] in one generation of FrameArgs::order_state
#define RT_ORD_MAX_TILES 262144u // 8K frames (129 600 tiles) included: the order still pays there (measured), the state is 20 B per tile and generation

#endif
