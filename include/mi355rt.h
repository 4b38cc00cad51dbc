/*
 * mi355rt.h -- C ABI of the MI355X-native per-pixel ray-tracing path (libmi355rt.so).
 *
 * This is the drop-in boundary underneath the reference's C++ back-end interface.  The reference
 * selects a back end at link time by defining three functions (include/update.h:6-8) on top of the
 * scene model (include/scene.h:8-36) filled by its YAML loader (src/scene.cpp:154-203).  Each entry
 * point below names the reference interface it replaces; cuda-ray-tracer_amd/host/src/update-hip.cpp
 * is the adapter that implements update.h on this ABI, INTEGRATION.md shows how a maintainer of
 * the reference links it.
 *
 * Conventions: plain pointers and sizes only, no C++/torch types; every function returns RT_OK (0) or a
 * negative rt_status and never throws; rt_last_error() gives the message of the calling thread's last
 * failure.  A context belongs to one thread at a time (the reference back ends keep their state in
 * file-scope globals, src/update-cpu.cpp:10-19; here it is an explicit object).
 */
#ifndef MI355RT_H
#define MI355RT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_ABI_VERSION 3
#define RT_ABI_DIAGNOSTIC 0x4000 /* set in rt_abi_version() of a library built with STAMPS / DEBUG_EXITS / SPILLS_OK / ...: for
                                    measurements only (it may spill registers to scratch, which the product never does) */

typedef enum rt_status {
    RT_OK = 0,
    RT_ERR_INVALID = -1,   /* bad argument */
    RT_ERR_SCENE = -2,     /* scene description rejected (what SceneException reports in the reference) */
    RT_ERR_DEVICE = -3,    /* HIP runtime error */
    RT_ERR_NO_DEVICE = -4, /* no usable GPU: the product has no CPU fallback */
    RT_ERR_NOMEM = -5
} rt_status;

/* Number of doubles per surface: SurfaceCoefs, include/surface.h:10-15, same order
 * x3 y3 z3 x2y xy2 x2z xz2 y2z yz2 xyz x2 y2 z2 xy xz yz x y z c */
#define RT_NCOEF 20

/* Flat view of a Scene (include/scene.h:17-36).  Arrays are borrowed for the duration of the call
 * that receives the descriptor (the reference back ends copy the scene too, src/update-cpu.cpp:29-30). */
typedef struct rt_scene_desc {
    uint32_t width, height;        /* Scene::px_width / px_height */
    double vertical_fov;           /* Scene::vertical_fov, RADIANS */
    float bg_color[3];             /* Scene::bg_color */
    uint32_t max_reflections;      /* Scene::max_reflections */
    uint32_t n_objects;            /* Scene::objects.size() */
    uint32_t n_lights;             /* Scene::lights.size() */
    const double *coefs;           /* [n_objects][RT_NCOEF]   Object::surface */
    const float *reflection;       /* [n_objects]             Object::reflection_ratio */
    const float *albedo;           /* [n_objects][3]          Object::color */
    const uint8_t *light_is_spherical; /* [n_lights]          LightSource::is_spherical */
    const double *light_p;         /* [n_lights][3]           LightSource::p (direction already -normalised) */
    const float *light_color;      /* [n_lights][3]           LightSource::light_color (intensity folded in) */
} rt_scene_desc;

/* rt_config.flags */
#define RT_FLAG_STRICT 0u     /* default: FP64 geometry / FP32 colour with NO FMA contraction -- the arithmetic
                                 of src/update-cpu.cpp on x86-64; this is the parity mode */
#define RT_FLAG_FAST 1u       /* same algorithm with FMA contraction allowed (results differ by <=1 ulp per
                                 operation; pixels on solver discontinuities may flip) */
#define RT_FLAG_COUNT 2u      /* also count rays / intersection tests on the device (slower; for accounting) */
#define RT_FLAG_SIMPLE 4u     /* use the simple one-thread-per-pixel kernel (rt_kernels.hip) instead of the
                                 workgroup wavefront kernel (rt_wavefront.hip); same results, kept for A/B runs */
#define RT_FLAG_NOCULL 8u     /* wavefront kernel: test every object for every shadow ray (no bounding-volume
                                 culling); same results, kept for A/B runs and as a cross-check */
#define RT_FLAG_STATIC_ORDER 16u /* wavefront kernel: start the tiles in index order every frame instead of starting
                                 the tiles that had hits in the previous frame first; same results, for A/B runs */

#define RT_FLAG_NOSCAN 32u    /* wavefront kernel: no scan workgroups (every tile's own workgroup decides whether the tile is
                                 empty and paints it); same results, for A/B runs */

#define RT_FLAG_PLAIN_ORDER 64u /* wavefront kernel: the tiles that had hits start in plain descending-cost order instead of the order that
                                  balances the CUs' loads (rt_wavefront.hip, ord_rank_of_slot); same results, for A/B runs */
#define RT_FLAG_NOSPLIT 128u   /* wavefront kernel: every tile is rendered by one workgroup (default: the costliest tiles of the previous frame
                               * by two, one per half tile) -- for A/B */

#define RT_FLAG_NOLEAN 256u    /* wavefront kernel: scenes of unit spheres without mirrors are rendered by the general instantiation (hit queue per
                               * tile) instead of the wave-per-block one; same results, for A/B runs and as a cross-check */

/* rt_config.format -- framebuffer pixel format */
#define RT_FMT_RGBA32F 0u     /* 4 x float per pixel, alpha 1.0: the un-quantised colours the CPU back end
                                 produces (src/update-cpu.cpp:128-131) plus an alpha lane for 16-byte stores */
#define RT_FMT_RGBA8 1u       /* iround(c*255) RGBA8, alpha 255: the wire format of src/update-cuda.cu:149-156 */

typedef struct rt_config {
    int32_t device;      /* HIP device ordinal, -1 = the calling thread's current device */
    uint32_t rank;       /* this context renders the row bands b with b % world == rank */
    uint32_t world;      /* number of row-band owners (1 = whole frame) */
    uint32_t band_rows;  /* rows per band, 0 = default (8) */
    uint32_t flags;      /* RT_FLAG_* */
    uint32_t format;     /* RT_FMT_* */
} rt_config;

/* Device-side work counters of the last RT_FLAG_COUNT render (definitions: SURVEY.md 8(d)). */
typedef struct rt_counters {
    uint64_t primary_rays;  /* one per pixel */
    uint64_t shadow_rays;   /* shadow_ray calls: hits x lights (light_impl.h:17) */
    uint64_t reflect_rays;  /* reflect_ray calls (light_impl.h:46) */
    uint64_t tests;         /* ray-surface tests = intersect_ray calls in the reference (surface_impl.h:21) */
    uint64_t hits;          /* nearest-hit records shaded (normal_vector calls, surface_impl.h:157) */
    uint64_t solves;        /* root solves (sqrt + division, or the cubic solver) actually executed */
    uint64_t tests_executed; /* t2,t1,t0 evaluations actually executed (differs from `tests`: no early break,
                                minus culled objects) */
    uint64_t cull_evals;    /* bounding-sphere culling decisions evaluated (one lane each) */
} rt_counters;

/* The work the PRODUCT build executes for the frame of the last RT_FLAG_COUNT render, split the way the flop accounting needs
 * it (bench.py multiplies each entry with the cost of that unit, counted by tools/count_flops.cpp over the kernel's own math).
 * (A counting render itself traces more shadow rays than the product build -- it needs every first blocker's index for
 * rt_counters.tests -- but counts as executed only what the product build executes.) */
typedef struct rt_counters_detail {
    uint64_t tests_executed[4]; /* per surface class: unit sphere, other quadric, plane, cubic */
    uint64_t solves[3];         /* root solves per class: unit sphere, other quadric, plane (cubic: cubic_branch) */
    uint64_t cull_evals[5];     /* culling decisions: tile pyramid, primary cone, shadow phase directional / point light; records formed */
    uint64_t cubic_branch[4];   /* cubic tests by solver branch: Cardano, trigonometric, quadratic, linear / none (surface_impl.h:106-154) */
    uint64_t shadow_rays_traced; /* of rt_counters.shadow_rays: those not skipped because the hit faces away from the light */
    uint64_t hit_lights_shaded;  /* surface_color evaluations (light_impl.h:29) */
    uint64_t primary_rays_formed; /* pixels of the tiles that are actually traced (rt_counters.primary_rays counts every pixel) */
    uint64_t cubic_points;        /* (ABI 3) degree-3 surfaces: evaluations of F, grad F and the half Hessian at a ray origin (lanes); the data
                                   * of the frame's own origin comes from the host and is not counted */
    uint64_t cubic_refused;       /* (ABI 3) of tests_executed[3]: tests whose Taylor-form answer the guard refused (rt_math.hpp, cubic_guarded) and
                                   * that went through the reference's dense expansion and solver instead */
} rt_counters_detail;

typedef struct rt_ctx rt_ctx;
typedef struct rt_scene rt_scene;

int rt_abi_version(void);
const char *rt_last_error(void);
/* For layers built on this ABI (libmi355rt_multi.so): set the calling thread's error text. */
void rt_set_last_error(const char *message);

/* ---------------------------------------------------------------------------------------------------
 * Scene loading -- replaces Scene::load_from_file (include/scene.h:35, src/scene.cpp:154-203) and the
 * factories it calls (src/surface.cpp:4-60, src/light.cpp:4-26).  Same keys, defaults, validation and
 * error texts; own YAML-subset parser (yaml-cpp is not a dependency).
 * ------------------------------------------------------------------------------------------------- */
int rt_scene_load_file(const char *path, rt_scene **out);
/* Programmatic construction (Scene::Scene, src/scene.cpp:16-22; fov in DEGREES like the constructor). */
int rt_scene_new(uint32_t width, uint32_t height, double fov_deg, uint32_t max_reflections,
                 const float bg_color[3], rt_scene **out);
/* Object::Object (src/scene.cpp:9-14) with an explicit coefficient vector. */
int rt_scene_add_object(rt_scene *s, const double coefs[RT_NCOEF], float reflection_ratio, const float color[3]);
/* LightSource::directional / spherical (src/light.cpp:4-26); v = direction or position. */
int rt_scene_add_light(rt_scene *s, int is_spherical, float intensity, const double v[3], const float color[3]);
/* Surface factories, src/surface.cpp:4-60.  kind: 0 sphere(a=center,b[0]=radius) 1 plane(a=origin,b=normal)
 * 2 dingDong(a=origin) 3 clebsch 4 cayley. */
int rt_surface_make(int kind, const double a[3], const double b[3], double out_coefs[RT_NCOEF]);
/* Overrides of the public Scene fields (include/scene.h:19-22); bench configs use resolutions the YAML
 * files do not contain. */
int rt_scene_set_size(rt_scene *s, uint32_t width, uint32_t height);
int rt_scene_set_max_reflections(rt_scene *s, uint32_t max_reflections);
/* Borrowed view, valid until the scene is modified or freed. */
int rt_scene_get_desc(const rt_scene *s, rt_scene_desc *out);
void rt_scene_free(rt_scene *s);

/* The reference host's camera (src/ray-tracer.cpp:25-58): camera-to-world matrix
 * inverse(lookAt(pos, pos - direction(yaw, pitch), +y)), column-major, 16 doubles -- the argument update()
 * receives every frame.  Start-up pose (pos 0, yaw 90, pitch 0) is the identity to ~6e-17. */
int rt_camera_matrix(const double pos[3], double yaw_deg, double pitch_deg, double out_cam[16]);

/* ---------------------------------------------------------------------------------------------------
 * Rendering
 * ------------------------------------------------------------------------------------------------- */
/* Replaces init_update (include/update.h:6; src/update-cpu.cpp:22-43, src/update-cuda.cu:34-63):
 * copies the scene to the device, precomputes aspect and tan(fov/2), allocates the local framebuffer. */
int rt_create(rt_ctx **out, const rt_scene_desc *scene, const rt_config *cfg);

/* Replaces update (include/update.h:7; src/update-cpu.cpp:121-139, src/update-cuda.cu:160-190).
 *   cam     camera-to-world dmat4, column-major, 16 doubles (src/ray-tracer.cpp:54-58)
 *   dev_fb  device pointer receiving this rank's rows ([local_rows][width] pixels of cfg.format), or NULL for
 *           the context's own offscreen buffer
 *   stream  hipStream_t to launch on (NULL = default stream)
 *   ms      if non-NULL: the call synchronises and stores the device time of the render kernels in
 *           milliseconds (hipEvent pair, what the reference's update() returns); if NULL the call only
 *           enqueues work.
 * A context carries state from frame to frame on the device (which tiles had hits: the next frame starts those first; per-frame
 * tile words), so its frames run in the order they were issued: on one stream that is automatic, and when a call passes a
 * different stream than the previous one, that stream first waits for the previous frame (an event recorded behind every
 * render).  The state affects speed only, never the image.
 * Stream capture: with ms == NULL and the stream of the previous call, rt_render only enqueues (one kernel; the ordering event is
 * not recorded while the stream is capturing -- a context whose frames were captured must stay on that stream afterwards, a call on
 * another stream is refused with RT_ERR_INVALID), so a
 * sequence of K frames can be captured into one hipGraph and launched at once -- bench.py times its frames that way (the ~3 us the
 * command processor needs between two dependent launches disappear: 42 instead of 45 us per 1080p frame).  Every captured call
 * carries its own arguments (camera, launch-order generation, frame tag): launched once, in place of the K calls, the graph is
 * exactly those K frames; REPLAYING it renders correctly too but repeats frame tags and generations, i.e. without the benefit of
 * the ordering (use RT_FLAG_STATIC_ORDER for contexts whose graphs are replayed). */
int rt_render(rt_ctx *ctx, const double cam[16], void *dev_fb, void *stream, float *ms);

/* Row ownership: number of local rows, and for local row i its global y (row 0 = bottom of the image,
 * src/update-cpu.cpp:125-131).  rt_max_local_rows is the maximum over all ranks (gather stride). */
int rt_local_rows(const rt_ctx *ctx, uint32_t *n_rows);
int rt_max_local_rows(const rt_ctx *ctx, uint32_t *n_rows);
int rt_row_map(const rt_ctx *ctx, uint32_t *rows /* [local_rows] */);
size_t rt_pixel_bytes(const rt_ctx *ctx);

/* Offscreen buffer of the context (device pointer) and a blocking copy of it to host memory. */
void *rt_device_fb(rt_ctx *ctx);
int rt_download(rt_ctx *ctx, void *host_dst, size_t bytes);

/* Root-side reassembly after the gather (the only collective of the path): `gathered` holds
 * [world][max_local_rows][width] pixels in rank order, `full` receives [height][width] pixels in row order.
 * Both are device pointers; enqueued on `stream`. */
int rt_assemble(rt_ctx *ctx, const void *gathered, void *full, void *stream);

/* Sparse transport of an RGBA8 frame (what `rt_assemble` does, with fewer bytes over the links): most 16x16 tiles of a
 * typical frame are pure background, so a rank may send only the others.  rt_pack_sparse turns this rank's rows
 * (dev_fb, or the context's own buffer if NULL) into a fixed-size message of rt_sparse_bytes(capacity_tiles) bytes:
 *   uint32 { count, overflow, 0, 0 }, uint32 ids[capacity] (padded to 16 bytes), capacity x 256 RGBA8 pixels;
 * `overflow` != 0 means more than capacity_tiles tiles had content (send the dense frame instead).  The root gathers
 * the messages ([world][rt_sparse_bytes] in rank order) and rt_assemble_sparse rebuilds [height][width] pixels.
 * The reference has no counterpart (single GPU); the dense gather + rt_assemble stays the general path. */
size_t rt_sparse_bytes(uint32_t capacity_tiles);
/* rt_render that writes such a message directly (tiles in which a primary ray hit something; background tiles are not
 * stored anywhere): one kernel instead of render + pack, and no local framebuffer.  Arguments as rt_render. */
int rt_render_sparse(rt_ctx *ctx, const double cam[16], void *dev_msg, uint32_t capacity_tiles, void *stream, float *ms);
int rt_pack_sparse(rt_ctx *ctx, const void *dev_fb, void *dev_msg, uint32_t capacity_tiles, void *stream);
int rt_assemble_sparse(rt_ctx *ctx, const void *gathered_msgs, uint32_t capacity_tiles, void *full, void *stream);
/* The same without repainting the whole frame every time: `full` and `stamps` (rt_sparse_stamp_bytes(ctx) bytes of device
 * memory owned by the caller, one pair per output buffer) carry over from the previous call on that buffer; only tiles an
 * earlier frame delivered and this one did not are painted back to the background.  frame_tag: 0 on the first call for a
 * buffer (paints everything, clears the stamps), afterwards any value in [1, 0xFFFFFFFE] that differs from the previous
 * call's. */
size_t rt_sparse_stamp_bytes(rt_ctx *ctx);
int rt_assemble_sparse_incremental(rt_ctx *ctx, const void *gathered_msgs, uint32_t capacity_tiles, void *full, void *stamps, uint32_t frame_tag,
                                   void *stream);

/* Counters of the last render done with RT_FLAG_COUNT. */
int rt_get_counters(rt_ctx *ctx, rt_counters *out);
int rt_get_counters_detail(rt_ctx *ctx, rt_counters_detail *out); /* wavefront kernel only */

/* Diagnostics: the raw device counter block (32 words).  Words 0-7 are rt_counters; words 8+ are per-phase
 * wave-cycle totals that only a library built with `make STAMPS=1` fills in. */
int rt_debug_counters(rt_ctx *ctx, uint64_t out[32]);
/* Diagnostics (`make STAMPS=1` + MI355RT_DEBUG_COUNTERS=1 only): the last frame's per-wave rows of 16 words -- 0-11 cycles per phase,
 * 12 / 13 a 100 MHz clock at the wave's start / end; row = workgroup * 4 + wave.  out = NULL: only the row count. */
int rt_debug_stamp_rows(rt_ctx *ctx, uint64_t *out, size_t max_rows, size_t *n_rows);

/* Replaces cleanup_update (include/update.h:8). */
int rt_destroy(rt_ctx *ctx);

/* ---------------------------------------------------------------------------------------------------
 * Several GPUs of one node behind one call -- libmi355rt_multi.so (links librccl; libmi355rt.so itself does not).
 * The reference is single-GPU (src/update-cuda.cu:160-190); BASELINE.json's north star tiles the rows over the GPUs of a
 * node and gathers them on one.  One process: a context per (device, part) with the rows band-cyclic
 * over all n_devices * parts contexts, every device rendering part after part on its own stream, finished parts
 * travelling to devices[0] by ncclSend / ncclRecv over xGMI on a second stream while the next part renders, and
 * rt_assemble restoring row order there.  host/src/update-hip.cpp uses it when MI355RT_DEVICES names more than one
 * device, so update() (include/update.h:7) returns the whole frame whatever the number of GPUs.
 *   devices    HIP ordinals; devices[0] is the root that ends up with the frame.  All distinct: RCCL.  One ordinal repeated
 *              n times: the same choreography with device-to-device copies instead of RCCL (for a one-GPU box).
 *   parts      contexts per device (0 = 1): more parts = finer overlap of transfer and rendering
 *   flags      RT_FLAG_* of every context, plus RT_MULTI_SELF_EXCHANGE
 * rt_render_multi: root_full_fb = device pointer on devices[0] receiving [height][width] pixels, or NULL for the object's
 * own buffer (rt_multi_fb); enqueue-only unless ms is given (then: device time on the root from the start of its render
 * to the end of the reassembly, transfers included).  rt_multi_stream() is the root stream the frame is complete on.
 * ------------------------------------------------------------------------------------------------- */
#define RT_MULTI_SELF_EXCHANGE 0x10000u /* one device: send its rows to itself through RCCL instead of rendering in place
                                           (exercises the RCCL path on a one-GPU box) */
#define RT_MULTI_BANDWISE 0x20000u      /* rows travel band by band straight into their place in the full frame (one ncclSend / ncclRecv pair per band,
                                         * one strided copy per context on the root device): no rank-major receive slots, no rt_assemble pass */
typedef struct rt_multi rt_multi;
int rt_create_multi(rt_multi **out, const rt_scene_desc *scene, const int *devices, uint32_t n_devices, uint32_t band_rows, uint32_t parts,
                    uint32_t flags, uint32_t format);
int rt_render_multi(rt_multi *m, const double cam[16], void *root_full_fb, float *ms);
int rt_multi_wait(rt_multi *m);                       /* host waits until the last frame is complete on the root */
void *rt_multi_fb(rt_multi *m);                       /* the object's own full-frame buffer on devices[0] */
void *rt_multi_stream(rt_multi *m);                   /* hipStream_t on devices[0] */
int rt_multi_download(rt_multi *m, void *host_dst, size_t bytes);
int rt_multi_info(const rt_multi *m, uint32_t *n_contexts, uint32_t *transport /* 0 in place, 1 device copies, 2 RCCL */);
int rt_multi_destroy(rt_multi *m);

#ifdef __cplusplus
}
#endif
#endif /* MI355RT_H */
