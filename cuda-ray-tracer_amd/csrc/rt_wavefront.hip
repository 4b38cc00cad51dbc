// rt_wavefront.hip -- the production render kernel for gfx950 (MI355X): a workgroup-level wavefront
// pipeline whose working state lives entirely in LDS.  Compiled twice like rt_kernels.hip
// (RT_VARIANT = strict | fast).
//
// It computes, pixel for pixel, what the reference's render_pixel / get_color_and_object compute
// (src/update-cpu.cpp:45-119, through include/surface_impl.h and include/light_impl.h; the
// reference's GPU twin is update_kernel, src/update-cuda.cu:104-158) -- the arithmetic of every value
// that reaches a pixel is rt_math.hpp's, shared with the simple kernel.  What differs is the schedule:
//
//   one workgroup (4 waves) owns a 16x16 pixel tile and iterates ROUNDS (round 0 = primary rays, round k =
//   k-th mirror bounce) of phases separated by LDS-only workgroup barriers (lds_barrier):
//
//   A  nearest hit   one lane per live pixel.  Primary rays: unit spheres are first culled against the wave's
//                    8x8-pixel ray cone (primary_cone_mask).  Wave-uniform loops over the per-class tables only
//                    form t1, t0 and the sign of the discriminant; the sqrt + divisions of the root solve are
//                    DEFERRED to a short per-lane loop over the few objects that can hit (a 64-bit candidate mask
//                    per lane).  Hits are compacted into an LDS queue (ballot prefix + per-wave counts).
//                    A tile whose round finds no hit at all stops here: one barrier, store, done.
//   A' hits          per 64-hit chunk: bounding box of the chunk's hit points (six FP32 DPP reductions, rounded outwards), its
//                    ball, and a culling record per sphere (the light-independent half of the shadow-phase culling).
//   B  shadow rays   every wave visits every chunk and takes the lights == (wave - chunk) mod 4.  Per (chunk,
//                    light) the wave first CULLS: lane j decides whether sphere j can possibly touch any shadow ray
//                    of the chunk (distance of its centre to the chunk's swept bounding volume, with a generous
//                    margin -- purely conservative, see relevant_mask); the ballot of that is a wave-uniform object
//                    mask.  Then the same two-step test as in A over the surviving objects only -- except the hit's own
//                    sphere where the own-sphere rule applies.  One 64-bit mask per (chunk, light) in LDS.
//   C  shading       wave c shades chunk c, one lane per hit: lights in order, Lambert term for the unshadowed ones, FP32 accumulate,
//                    clamp (src/update-cpu.cpp:57-78).
//   D  blend/bounce  the pixel's owner lane blends the colour into its running result and, for mirrors, sets
//                    up the next round's ray (src/update-cpu.cpp:96-117).  One flag per wave + one barrier tell
//                    whether any pixel of the tile is still bouncing.
//
//   Every lane finally stores its pixel (16-byte RGBA32F or 4-byte RGBA8, rows of the tile contiguous).
//
//   That is the GENERAL schedule (template argument LEAN = false): scenes with planes, general quadrics, degree-3 surfaces or
//   mirrors, counting renders, sparse output, and all-sphere views in which the tiles with hits do not fill the GPU (there the
//   costliest tiles are split into half tiles and the lights of a chunk are spread over the workgroup's four waves).
//   The LEAN schedule (LEAN = true, lean_block below) is what an all-sphere scene without mirrors runs while its tiles with hits
//   fill the GPU: every wave keeps its own 8x8 block from the primary ray to the store -- hits stay in their pixel's lane (no
//   queue, no compaction), the lights are visited in order and shaded on the spot (no shadow masks, no phase C), and after the
//   staging barrier the workgroup never synchronises again.  rt_capi.cpp chooses between the two per frame from the previous
//   frame's census.
//
//   One launch renders the frame: classify workgroups decide a word per tile (empty / has hits / rendered by a list slot),
//   list slots render the tiles that had hits in the previous frame, costliest first (16 cost classes, dealt out in a snake over
//   the CUs; cost = what the tile's waves spent on their lights, two s_memtime reads), paint workgroups fill the empty tiles in
//   bulk, and index slots (one per tile) leave after one load unless their tile has hits that no list slot covers (temporal
//   coherence of an interactive camera; a stale list only costs time, never a pixel).
//
// Scene data (rt_scene_dev.h): the per-class tables are staged into LDS by every workgroup that traces (wave-uniform loops read
// them as LDS broadcasts, deferred solves / normals / albedo gather per lane); the lights are never staged -- the LightK table is
// read through the constant address space, so a light's constants are scalar loads into SGPRs.  The kernel is instantiated per
// scene feature (general quadrics, degree-3 surfaces, mirrors, lean): an instantiation contains no code and no registers for a
// feature the scene does not have.  The sphere instantiations fit 80 VGPRs (six workgroups per CU); no instantiation spills
// VGPRs or has a private segment (make refuses to link otherwise).  DESIGN.md section 5 has the measurements behind each choice.
#include <hip/hip_runtime.h>
#include <cstdlib>

#include "rt_math.hpp"
#include "rt_scene_dev.h"
#include "rt_wavefront_math.hpp"

#ifndef RT_VARIANT
#error "define RT_VARIANT=strict|fast"
#endif
#define RT_CAT2(a, b) a##_##b
#define RT_CAT(a, b) RT_CAT2(a, b)
#define RT_SYM(name) RT_CAT(name, RT_VARIANT)

namespace RT_SYM(rtw) {

using namespace rtm;

// One light out of the LightK table (rt_scene_dev.h) through the constant address space: member by member, every one a scalar
// load, issued together -- a directional light's constants then live in SGPRs.  `clight` must be in scope.
#define RT_LOAD_LIGHTK(lk, l) \
    LightK lk; \
    lk.p[0] = clight[l].p[0]; lk.p[1] = clight[l].p[1]; lk.p[2] = clight[l].p[2]; \
    lk.sdir[0] = clight[l].sdir[0]; lk.sdir[1] = clight[l].sdir[1]; lk.sdir[2] = clight[l].sdir[2]; \
    lk.u2 = clight[l].u2; lk.inv_uu = clight[l].inv_uu; lk.len_u = clight[l].len_u; lk.four_u2 = clight[l].four_u2; \
    lk.s_yz = clight[l].s_yz; lk.s_xz = clight[l].s_xz; lk.s_xy = clight[l].s_xy; \
    lk.color[0] = clight[l].color[0]; lk.color[1] = clight[l].color[1]; lk.color[2] = clight[l].color[2]; \
    lk.flags = clight[l].flags;
typedef const __attribute__((address_space(4))) LightK *ConstLights;

constexpr uint32_t WG = 256; // threads per workgroup = pixels per tile

// Pixel of thread `tid` inside its 16x16 tile: wave w owns the 8x8 quadrant (w & 1, w >> 1), lane l the pixel (l & 7, l >> 3)
// of it.  Square blocks keep a wave's primary rays in a narrow cone (tighter culling than 16x4 strips: the circumscribed
// circle has 11.3 instead of 16.5 pixels diameter) and make a 64-hit chunk of the queue spatially compact (smaller
// bounding balls in the shadow phase); a row of the block is still 128 contiguous bytes of RGBA32F framebuffer.
__device__ __forceinline__ uint32_t tile_px(uint32_t tid) { return ((tid >> 6) & 1u) * 8u + (tid & 7u); }
__device__ __forceinline__ uint32_t tile_py(uint32_t tid) { return (tid >> 7) * 8u + ((tid >> 3) & 7u); }

// Diagnostic build only (make STAMPS=1): per-phase wave-cycle totals into counters[8..], read with
// rt_debug_counters().  The product build contains no stamp.
#ifdef RT_WF_STAMPS
struct StampState {
    unsigned long long ph[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tl, w0, info = 0;
    __device__ __forceinline__ StampState() : tl(clock64()), w0(wall_clock64()) {}
};
#define RT_STAMP_DECL StampState st_;
#define RT_STAMP_ARG , StampState &st_
#define RT_STAMP_PASS , st_
#define RT_STAMP(i) do { unsigned long long t_ = clock64(); st_.ph[i] += t_ - st_.tl; st_.tl = t_; } while (0)
// one private 16-word row per wave (pointer in counters[31]) -- no atomics, which would distort the timings.  Words 0-11: cycles
// per phase; 12 / 13: constant-rate clock (100 MHz) at the wave's start / end, for tools/timeline.py
#define RT_STAMP_FLUSH(c, lane) do { if ((lane) == 0) { unsigned long long *row_ = reinterpret_cast<unsigned long long *>((c)[31]) + \
    ((size_t) (stamp_row_base_ + blockIdx.x) * 4 + (threadIdx.x >> 6)) * 16; for (int i_ = 0; i_ < 12; i_++) row_[i_] = st_.ph[i_]; row_[12] = st_.w0; row_[13] = wall_clock64(); row_[14] = st_.info; \
    row_[15] = ((unsigned long long) __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32) | (unsigned) __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)); } } while (0)
#define RT_STAMP_INFO(x) do { st_.info = (x); } while (0)
#else
#define RT_STAMP_DECL
#define RT_STAMP_ARG
#define RT_STAMP_PASS
#define RT_STAMP(i)
#define RT_STAMP_FLUSH(c, lane)
#define RT_STAMP_INFO(x)
#endif
// Occupancy target (waves per SIMD) of each instantiation = the highest one at which NOTHING spills to scratch.
// That is a correctness rule, not tuning: ROCm 7.2's compiler can place a VGPR spill at the top of a join block before
// the `s_or_b64 exec` that re-enables the lanes of a divergent region, so the lanes that sat the region out read back
// whatever the scratch slot held (tools/check_spills.py has the story and is run by `make`).  SGPR spills (to VGPR
// lanes) are harmless.  RT_WF_OCC_DELTA (make OCCD=-1 ...) shifts every target for experiments.
#ifndef RT_FAST
#define RT_FAST 0
#endif
#ifndef RT_WF_OCC_DELTA
#define RT_WF_OCC_DELTA 0
#endif
#ifndef RT_WF_LEAN_OCC
#define RT_WF_LEAN_OCC 6
#endif

template <bool COUNT, bool HAS_GQ, bool HAS_CUBIC, bool HAS_MIRROR, bool LEAN = false>
constexpr int wf_occupancy()
{
    if (LEAN) return COUNT ? 3 : (RT_FAST ? RT_WF_LEAN_OCC - 1 : RT_WF_LEAN_OCC); // (the FMA build's schedule needs two registers more) // the wave-per-block path (see "the lean path" in the kernel): no hit queue, few live values
    int occ = HAS_CUBIC ? ((!HAS_MIRROR && !HAS_GQ) ? 4 : 2) : (HAS_MIRROR ? (HAS_GQ ? 3 : 4) : (HAS_GQ ? 3 : 6)); // 80 / 128 / 168 / ~200 VGPRs (strict, no counters); the
                                                                                  // mirror-free general-quadric one needs 129-130 at 4.
    // Spheres and planes without mirrors run at SIX workgroups per CU (80 VGPRs, and an LDS carve-up that fits six times into 160 KB:
    // LdsLayout): same-box A/B against five -- 1080p 46.2 -> 44.3 us, 4K 127.6 -> 119.2, 8K 474 -> 428, orbit pose 19 59.2 -> 54.8.
    if (!HAS_CUBIC && RT_FAST && !COUNT && HAS_GQ) occ -= 1; // the FMA build's different schedule needs a few registers more with general quadrics
                                                             // (the sphere instantiations fit: 80 at 6, 128 at 4 with mirrors)
    if (!HAS_CUBIC && COUNT) occ -= (HAS_GQ || HAS_MIRROR) ? 2 : 3; // 8 + 18 counters in registers; counting renders are not timed
    if (HAS_CUBIC && COUNT) occ = 1;                // counting builds inline the cubic path (they report its solver branch)
    if (!HAS_CUBIC && COUNT && RT_FAST && HAS_MIRROR) occ -= 1;
    occ += RT_WF_OCC_DELTA;
    return occ < 1 ? 1 : occ;
}
#ifndef RT_WF_CAMTAB
#define RT_WF_CAMTAB 1
#endif


// counters[]: 0 primary 1 shadow 2 reflect 3 tests (reference-equivalent: what the reference would have run) 4 hits
//             5 solves executed 6 tests executed 7 cull evaluations
//             32.. the work the PRODUCT build executes for the same frame, split up for the flop accounting of bench.py
//             (rt_get_counters_detail).  A counting build runs every test the reference would run -- it needs the index of the
//             first blocker -- but counts as "executed" only what the product build executes (lanes that face the light):
//             32-35 tests executed per surface class (unit sphere, other quadric, plane, cubic)   36-38 root solves per class
//             39-43 culling: tile pyramid, primary cone, shadow phase directional / point light, records formed
//             44-47 cubic tests by solver branch   48 shadow rays traced (of counters[1] considered)   49 hits shaded per light
//             50 primary rays formed (pixels of the tiles that are traced; counters[0] counts every pixel, as the reference does)
//             51 Taylor data of a degree-3 surface formed at a ray origin (lanes)   52 degree-3 tests the guard handed to the dense path
enum { K_US = 0, K_GQ = 1, K_LIN = 2, K_CUB = 3 };
enum { C_TILE = 0, C_PRIMARY = 1, C_SHADOW_DIR = 2, C_SHADOW_SPH = 3, C_RECORDS = 4 };
constexpr int N_CNT_W = 21;
template <bool COUNT>
struct Cnt {
    __device__ __forceinline__ void add(int, unsigned long long = 1) {}
    __device__ __forceinline__ void exec(int, unsigned long long) {}
    __device__ __forceinline__ void solve(int) {}
    __device__ __forceinline__ void cull(int, unsigned long long) {}
    __device__ __forceinline__ void cubic(int, bool = true) {}
    __device__ __forceinline__ void traced() {}
    __device__ __forceinline__ void shaded() {}
    __device__ __forceinline__ void primary_traced() {}
    __device__ __forceinline__ void cubic_point() {}
    __device__ __forceinline__ void cubic_refused(bool) {}
    __device__ __forceinline__ void flush(unsigned long long *) {}
};
template <>
struct Cnt<true> {
    unsigned long long v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t w[N_CNT_W] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; // per thread and frame: 32 bits are plenty
    __device__ __forceinline__ void add(int i, unsigned long long n = 1) { v[i] += n; }
    __device__ __forceinline__ void exec(int cls, unsigned long long n) { v[6] += n; w[cls] += (uint32_t) n; }
    __device__ __forceinline__ void solve(int cls) { v[5] += 1; w[4 + cls] += 1; }
    __device__ __forceinline__ void cull(int kind, unsigned long long n) { if (kind != C_RECORDS) v[7] += n; w[7 + kind] += (uint32_t) n; }
    __device__ __forceinline__ void cubic(int branch, bool executed = true) // one test = expansion + solver
    {
        if (!executed) return;
        v[5] += 1; v[6] += 1; w[K_CUB] += 1;
        w[12] += branch == 0; w[13] += branch == 1; w[14] += branch == 2; w[15] += branch == 3; // (no indexed access: the counters stay in registers)
    }
    __device__ __forceinline__ void traced() { w[16] += 1; }
    __device__ __forceinline__ void shaded() { w[17] += 1; }
    __device__ __forceinline__ void primary_traced() { w[18] += 1; }
    __device__ __forceinline__ void cubic_point() { w[19] += 1; } // Taylor data of a degree-3 surface formed at a ray origin (rt_math.hpp, cubic_at)
    __device__ __forceinline__ void cubic_refused(bool r) { w[20] += r; } // degree-3 tests that cubic_guarded handed to the dense expansion + the reference's solver
    __device__ __forceinline__ void flush(unsigned long long *g)
    {
#pragma unroll
        for (int i = 0; i < 8; i++)
            if (v[i]) atomicAdd(&g[i], v[i]);
#pragma unroll
        for (int i = 0; i < N_CNT_W; i++) // (fully unrolled: an indexed access would move the counters to scratch memory)
            if (w[i]) atomicAdd(&g[32 + i], (unsigned long long) w[i]);
    }
};

// The scene as staged in LDS.
struct SceneLds {
    const UsEntry *us;
    const GqEntry *gq;
    const LinEntry *lin;
    const uint32_t *cub;
    const MatEntry *mat;
    const DevLight *light;
    uint32_t cubrec, cubtmp, cubprim; // degree-3 scenes: LDS byte addresses of the record regions (LdsLayout)
    double *cubtmp_p;                 // the working records again, as a pointer for the stores
};

// Workgroup barrier for LDS-only communication: release/acquire at workgroup scope on the LDS address space only,
// so it waits for lgkmcnt (LDS) but not for vmcnt -- outstanding framebuffer stores and the tile-counter atomic
// stay in flight across it.  (__syncthreads() would also drain vmcnt.)
__device__ __forceinline__ void lds_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// ---- degree-3 objects: one test, out of line ----
// The surface's Taylor data at the ray origin (rt_math.hpp: CubicAt) waits in LDS as a record of RT_CUB_REC doubles: rec + i * stride (bytes)
// is double i -- stride 8 for the packed records of the frame's origin, 8 * WG for the per-hit / per-lane ones; the object's CubicAbs (four
// doubles, from which the error bounds at this origin follow) likewise at abs + i * abs_stride.  Out of line (one copy per kernel, like the
// dense path it falls back to) and with few arguments: inlined at both call sites the guarded solver costs the instantiation a wave per SIMD.
typedef const __attribute__((address_space(3))) double *LdsD;
__device__ __forceinline__ uint32_t lds_addr(const void *p) { return (uint32_t) (uintptr_t) (const __attribute__((address_space(3))) void *) p; }
__device__ __forceinline__ void cubic_rec_store(double *rec, uint32_t stride_d, const CubicAt &a)
{
    rec[0] = a.f; rec[stride_d] = a.gx; rec[2 * stride_d] = a.gy; rec[3 * stride_d] = a.gz;
    rec[4 * stride_d] = a.hxx; rec[5 * stride_d] = a.hyy; rec[6 * stride_d] = a.hzz; rec[7 * stride_d] = a.hxy; rec[8 * stride_d] = a.hxz; rec[9 * stride_d] = a.hyz;
}
template <bool DENSE_INLINE>
__device__ __forceinline__ double cubic_test_body(const double *c, uint32_t rec_, uint32_t abs_, double ox, double oy, double oz, double dx, double dy, double dz, double max_t, bool decide,
                                                  bool &refused)
{
    // (bit 0 of either address: per-lane data, stride 8 * WG -- else packed, stride 8; LDS addresses of doubles are multiples of 8)
    const uint32_t rec = rec_ & ~1u, stride = (rec_ & 1u) ? WG * 8u : 8u, abs = abs_ & ~1u, abs_stride = (abs_ & 1u) ? WG * 8u : 8u;
#define RT_REC(i) (*(LdsD) (uintptr_t) (rec + (i) * stride))
#define RT_ABS(i) (*(LdsD) (uintptr_t) (abs + (i) * abs_stride))
    const CubicAt ca{RT_REC(0), RT_REC(1), RT_REC(2), RT_REC(3), RT_REC(4), RT_REC(5), RT_REC(6), RT_REC(7), RT_REC(8), RT_REC(9)};
    const CubicMag mo = cubic_mag_origin(CubicAbs{RT_ABS(0), RT_ABS(1), RT_ABS(2), RT_ABS(3)}, D3{ox, oy, oz});
#undef RT_REC
#undef RT_ABS
    return intersect_cubic_taylor<DENSE_INLINE>(c, ca, mo, D3{ox, oy, oz}, D3{dx, dy, dz}, max_t, decide, refused);
}
__device__ __noinline__ double cubic_test(const double *c, uint32_t rec, uint32_t abs, double ox, double oy, double oz, double dx, double dy, double dz, double max_t, bool decide)
{
    bool refused;
    return cubic_test_body<true>(c, rec, abs, ox, oy, oz, dx, dy, dz, max_t, decide, refused);
}

// Cross-lane helpers.  Reductions run on DPP row operations (VALU latency) instead of ds_bpermute round trips
// through the LDS crossbar; the result is taken from lane 63 with v_readlane and is wave-uniform.
__device__ __forceinline__ double readlane_d(double v, int l)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_d(double v)
{
    // lanes without a valid source (or masked rows) keep their own value: op(v, v) == v for min / max
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(v), __double2loint(v), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(v), __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
#define RT_WAVE_REDUCE(OP)                                                     \
    {                                                                          \
        double t;                                                              \
        t = dpp_d<0xB1, 0xf>(v); v = OP; /* quad_perm [1,0,3,2] */           \
        t = dpp_d<0x4E, 0xf>(v); v = OP; /* quad_perm [2,3,0,1] */           \
        t = dpp_d<0x124, 0xf>(v); v = OP; /* row_ror:4 */                    \
        t = dpp_d<0x128, 0xf>(v); v = OP; /* row_ror:8: every lane has its row of 16 */ \
        t = dpp_d<0x142, 0xa>(v); v = OP; /* row_bcast:15 into rows 1, 3 */  \
        t = dpp_d<0x143, 0xc>(v); v = OP; /* row_bcast:31 into rows 2, 3 */  \
        return readlane_d(v, 63);                                              \
    }
// 32-bit flavour: one DPP-modified v_min_f32 / v_max_f32 per step
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
#define RT_WAVE_REDUCE_F(OP)                                                     \
    {                                                                            \
        v = OP(v, dpp_f<0xB1, 0xf>(v));                                          \
        v = OP(v, dpp_f<0x4E, 0xf>(v));                                          \
        v = OP(v, dpp_f<0x124, 0xf>(v));                                         \
        v = OP(v, dpp_f<0x128, 0xf>(v));                                         \
        v = OP(v, dpp_f<0x142, 0xa>(v));                                         \
        v = OP(v, dpp_f<0x143, 0xc>(v));                                         \
        return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63)); \
    }
__device__ __forceinline__ float wave_min_f(float v) RT_WAVE_REDUCE_F(fminf)
__device__ __forceinline__ float wave_max_f(float v) RT_WAVE_REDUCE_F(fmaxf)
#undef RT_WAVE_REDUCE_F
__device__ __forceinline__ double wave_min(double v) RT_WAVE_REDUCE((t < v ? t : v))
__device__ __forceinline__ double wave_max(double v) RT_WAVE_REDUCE((t > v ? t : v))
#undef RT_WAVE_REDUCE

__device__ __forceinline__ unsigned long long primary_cone_mask(const UsEntry *us, uint32_t base, uint32_t end, const D3 &org,
                                                                const D3 &axis, double cos_t, uint32_t lane)
{
    bool rel = false;
    const uint32_t j = base + lane;
    if (j < end) {
        const UsEntry e = us[j];
        rel = sphere_in_cone(e.kx, e.ky, e.kz, e.r, e.inv_r, org, axis, cos_t);
    }
    return __ballot(rel);
}

// ---- launch order: which list entry does list slot q render? -------------------------------------------------------------------
// One generation of FrameArgs::order_state (uint32 words):
//   [0 .. 15]  number of tiles listed in cost class k (k = 0: the costliest)      [16] census      [17] largest tile cost seen
//   [RT_ORD_HDR + t]                        word of tile t: (position in its class list << 5) | (k + 1); 0 / stale = look again
//   [RT_ORD_HDR + (1 + k) * n_tiles + i]    i-th tile of class k
// The concatenation class 0, class 1, ... is the tiles in (roughly) descending cost; `rank` indexes it.
constexpr uint32_t ORD_CLASSES = 16;
struct OrdHeader {
    uint32_t cnt[ORD_CLASSES];
    uint32_t census, cost_max, n_listed;
    uint32_t n_candidates; // entries of the `split` costliest classes (what the host sizes the list slots for: the decision below may change from frame to frame)
    uint32_t n_heavy;      // those of them that ARE split this frame: each is rendered by TWO list slots, one per half tile (below)
};
// `split_arg` = (workgroup slots of the GPU for this kernel << 4) | cost classes that may be split (bits 4-7 and 20-31 of hot_flags)
template <typename Words>
__device__ __forceinline__ OrdHeader ord_header(Words rd, uint32_t split_arg = 0)
{
    const uint32_t split = split_arg & 15u, slots = split_arg >> 4;
    OrdHeader h;
    h.n_listed = 0;
#pragma unroll
    for (uint32_t k = 0; k < ORD_CLASSES; k++) {
        h.cnt[k] = rd[k];
        h.n_listed += h.cnt[k];
    }
    // Half tiles.  A tile full of hits keeps its four waves busy for 19 shadow items each while a tile with 64 hits is done after five,
    // and the frame ends when the costliest tile ends.  While the GPU has room (at most three quarters of its workgroup slots taken by
    // whole tiles -- beyond that the frame is bound by throughput and a second workgroup per tile only adds its fixed costs), an entry
    // of the `split` costliest classes gets TWO consecutive positions in the launch order: rows 0-7 of the tile and rows 8-15, each by
    // a workgroup of its own whose other two waves have no pixels but take their share of the shadow items.  Positions, not entries,
    // are what the list slots index:  position p < 2 * n_heavy -> entry p / 2, half 1 + (p & 1);  otherwise entry p - n_heavy, whole tile.
    // While the GPU holds two workgroups for EVERY listed tile, every tile is split (orbit poses 5 / 6: 712 - 746 tiles for 1536 slots); above
    // that only the costliest classes are: splitting most but not all tiles, or the cheap ones, costs more than it brings (pose 7, 839 tiles:
    // 36 us with the heavy ones split, 42 with everything the spare slots allow).
    const uint32_t spare = slots > h.n_listed ? slots - h.n_listed : 0u;
    h.n_heavy = 0;
#pragma unroll
    for (uint32_t k = 0; k < ORD_CLASSES; k++) h.n_heavy += k < split ? h.cnt[k] : 0u;
    if (split != 0u && h.n_listed * 2u <= slots) h.n_heavy = h.n_listed;
    if (h.n_listed * 4u > slots * 3u) h.n_heavy = 0;
    h.n_candidates = h.n_heavy; // (none while the GPU is full: list slots that only leave again are not free there)
    h.n_heavy = h.n_heavy < spare ? h.n_heavy : spare;
    h.census = rd[16];
    h.cost_max = rd[17];
    return h;
}
__device__ __forceinline__ uint32_t ord_positions(const OrdHeader &h) { return h.n_listed + h.n_heavy; }
// last position of entry `rank`: the entry is rendered by list slots iff that position is one of the n_eff in use
__device__ __forceinline__ uint32_t ord_last_position(const OrdHeader &h, uint32_t rank) { return rank < h.n_heavy ? 2u * rank + 1u : rank + h.n_heavy; }
// class and index of list entry `rank` (rank < h.n_listed)
__device__ __forceinline__ void ord_locate(const OrdHeader &h, uint32_t rank, uint32_t &k, uint32_t &idx)
{
    k = 0;
    idx = rank;
#pragma unroll
    for (uint32_t c = 0; c + 1 < ORD_CLASSES; c++) {
        const bool next = (k == c) && idx >= h.cnt[c];
        idx -= next ? h.cnt[c] : 0u;
        k += next ? 1u : 0u;
    }
}
// rank of the first entry of class k
__device__ __forceinline__ uint32_t ord_first(const OrdHeader &h, uint32_t k)
{
    uint32_t f = 0;
#pragma unroll
    for (uint32_t c = 0; c + 1 < ORD_CLASSES; c++) f += c < k ? h.cnt[c] : 0u;
    return f;
}
// Which entry list slot q gets.  At the start of a launch the dispatcher deals the workgroups out round robin -- workgroup b and
// b + 256 land on the same CU (measured: tools/timeline.py) -- and with one workgroup per slot nothing is re-balanced later, so a
// CU's load is the sum of the entries of its column.  Entries are in descending cost, so the rows alternate direction (a
// "snake"): the CU that got the costliest tile of one row gets the cheapest of the next.  `first_block` = blockIdx of slot 0.
__device__ __forceinline__ uint32_t ord_rank_of_slot(uint32_t q, uint32_t n_eff, uint32_t first_block)
{
    const uint32_t off = first_block & 255u, row = (q + off) >> 8;
    if (!(row & 1u)) return q;
    const uint32_t lo = row * 256u - off;                                   // (row >= 1)
    const uint32_t hi = (row + 1u) * 256u - off < n_eff ? (row + 1u) * 256u - off : n_eff;
    return lo + (hi - 1u - q);
}

// ---- tile words: who deals with an empty tile (all-sphere scenes) -------------------------------------------------------
// 83 % of config 2's tiles are empty, and a workgroup per empty tile that loads its arguments, reads the launch-order state,
// lets one wave test the tile, passes a barrier and paints 256 pixels holds a slot for ~2 us: 13 us of a 57 us frame at
// 1080p, 280 of 640 us at 8K (measured piece by piece: `make DEBUG_EXITS=1`, tools/empty_exits.py).  So the work is split:
//
//   classify   the first workgroups of the grid test 64 tiles each against the spheres (a wave takes sixteen consecutive
//              tiles, lane = (tile, sphere slot), four spheres per pass; the same five-plane pyramid test as the tile-level
//              early-out further down) and publish one word per tile:  (frame_tag << 3) | EMPTY / NONEMPTY / COVERED;
//   paint      behind the list slots, one workgroup per 16 tiles paints the background of the EMPTY ones (up to 1 KB
//              contiguous per store instruction), overlapping the tiles that trace;
//   index slot the workgroup that gets tile t in index order reads word t in its first instructions -- the pointer and the
//              tag are preloaded kernel arguments -- and leaves if it says EMPTY: one load, no argument fetch, no barrier.
//
// The four waves of an index-slot workgroup decide independently, so they must all see the SAME verdict: a word is decided
// exactly once per frame, by compare-and-swap from a value with an older tag, and every reader acts only on a value that
// carries this frame's tag.  A reader that does not find one polls (the classifying workgroups are the first of the grid, so
// in practice the word is there); after RT_TILE_MAX_POLLS polls it decides the word itself -- TIMEOUT: "this tile is its own
// workgroup's business", the pre-classification path with a barrier -- again by compare-and-swap, so if the classifier
// wins the race the reader follows the classifier, and vice versa.  Nothing ever waits without a bound, and whatever the
// dispatch order, every tile is either painted by a paint workgroup (EMPTY) or rendered / painted by exactly one tracing
// workgroup (NONEMPTY, TIMEOUT): the words change the time, never the image.  Words are written with agent-scope atomics
// and read with agent-scope loads (the workgroups involved may sit on different XCDs, whose L2s are not coherent for plain
// accesses).
constexpr uint32_t ST_EMPTY = 1u, ST_NONEMPTY = 2u, ST_TIMEOUT = 3u, ST_COVERED = 4u; // COVERED: not empty, and a list slot of this launch renders it
constexpr uint32_t ST_BITS = 3u, ST_MASK = 7u;
#ifndef RT_TILE_MAX_POLLS
#define RT_TILE_MAX_POLLS 48
#endif

__device__ __forceinline__ uint32_t tile_word_load(uint32_t *w) { return __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// value of *w once it carries `tag`, deciding it as TIMEOUT if it does not come; per-lane (callers mask the lanes)
__device__ __forceinline__ uint32_t tile_word_wait(uint32_t *w, uint32_t tag, uint32_t v)
{
    for (int it = 0; (v >> ST_BITS) != tag; ++it) {
        if (it >= RT_TILE_MAX_POLLS) {
            uint32_t expect = v;
            if (__hip_atomic_compare_exchange_strong(w, &expect, (tag << ST_BITS) | ST_TIMEOUT, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                v = (tag << ST_BITS) | ST_TIMEOUT;
            else
                v = expect; // somebody decided it meanwhile (only values with this frame's tag are written during this launch)
        } else {
            __builtin_amdgcn_s_sleep(16);
            v = tile_word_load(w);
        }
    }
    return v;
}

template <bool COUNT>
__device__ __forceinline__ void classify_tiles(const FrameArgs &fa, const UsEntry *us, uint32_t n_us, uint32_t *tile_state, uint32_t tag, uint32_t n_tiles,
                                               const uint32_t *ord_rd, uint32_t ord_cap, uint32_t ord_split, uint32_t wave, uint32_t lane, Cnt<COUNT> &cnt)
{
    const uint32_t i = lane & 15u, sl = lane >> 4;
    const uint32_t t = blockIdx.x * RT_SCAN_TILES + wave * 16u + i;
    const bool tv = t < n_tiles;
    uint32_t old = 0;
    if (sl == 0 && tv) old = tile_word_load(tile_state + t); // some older frame's word: the compare value of the publication below
    double kx = 0.0, ky = 0.0, kz = 0.0, r = 0.0, inv = 0.0;
    if (sl < n_us) { // this lane's first sphere: requested before the kernel arguments are even there
        const UsEntry *e = us + sl;
        kx = e->kx; ky = e->ky; kz = e->kz; r = e->r; inv = e->inv_r;
    }
    const uint32_t tc = tv ? t : n_tiles - 1u;
    const uint32_t tile_x = tc % fa.tiles_x, tile_y = tc / fa.tiles_x;
    const uint32_t x0 = tile_x * RT_TILE, y0l = tile_y * RT_TILE;
    const uint32_t x1 = x0 + RT_TILE - 1 < fa.width ? x0 + RT_TILE - 1 : fa.width - 1;
    const uint32_t y1l = y0l + RT_TILE - 1 < fa.local_rows ? y0l + RT_TILE - 1 : fa.local_rows - 1;
    const double gy0 = (double) global_row(fa, y0l), gy1 = (double) global_row(fa, y1l);
    const TilePlanes P = tile_planes(fa, fa.cx_a * ((double) x0 - 0.5) + fa.cx_b, fa.cx_a * ((double) x1 + 0.5) + fa.cx_b,
                                     fa.cy_a * (gy0 - 0.5) + fa.cy_b, fa.cy_a * (gy1 + 0.5) + fa.cy_b);
    const D3 org{fa.origin[0], fa.origin[1], fa.origin[2]};
    bool rel = sl < n_us && sphere_in_pyramid(kx, ky, kz, r, inv, org, P);
    for (uint32_t base = 4; base < n_us; base += 4) { // wave-uniform trip count
        const uint32_t j = base + sl;
        if (j < n_us) {
            const UsEntry *e = us + j;
            rel = rel || sphere_in_pyramid(e->kx, e->ky, e->kz, e->r, e->inv_r, org, P);
        }
    }
    if (!fa.tile_planes_ok) rel = true; // (the launcher starts no classifying workgroups then)
    unsigned long long m = __ballot(rel);
    m |= m >> 32;
    m |= m >> 16; // bit i: some sphere reaches into tile i of this wave
    const bool nonempty = (m >> i) & 1ull;
    if (sl == 0 && tv && (old >> ST_BITS) != tag) {
        uint32_t st = nonempty ? ST_NONEMPTY : ST_EMPTY;
        if (nonempty && ord_rd) { // launch-order lists in use: does a list slot of this launch render the tile?  (the test of the index slots, made here
                                  // once, so that the index slot of a covered tile leaves after one load as well)
            const OrdHeader oh = ord_header(ord_rd, ord_split); // (ord_split: classes | slots << 4)
            const uint32_t n_eff = ord_positions(oh) < ord_cap ? ord_positions(oh) : ord_cap;
            const uint32_t w = ord_rd[RT_ORD_HDR + t]; // (position in its class list << 5) | (class + 1): a hint where to look, possibly stale
            const uint32_t cls = w & 31u, pos = w >> 5;
            if (cls >= 1u && cls <= ORD_CLASSES && pos < n_tiles) {
                const uint32_t k = cls - 1u;
                uint32_t count = 0;
#pragma unroll
                for (uint32_t c = 0; c < ORD_CLASSES; c++) count = c == k ? oh.cnt[c] : count;
                if (pos < count && ord_last_position(oh, ord_first(oh, k) + pos) < n_eff && ord_rd[RT_ORD_HDR + (1u + k) * n_tiles + pos] == t) st = ST_COVERED;
            }
        }
        uint32_t expect = old;
        __hip_atomic_compare_exchange_strong(tile_state + t, &expect, (tag << ST_BITS) | st, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                             __HIP_MEMORY_SCOPE_AGENT); // fails only if a reader gave up on us (TIMEOUT): its decision stands
    }
    if (lane == 0) {
        const uint32_t first = blockIdx.x * RT_SCAN_TILES + wave * 16u;
        const uint32_t nt = first >= n_tiles ? 0u : (n_tiles - first < 16u ? n_tiles - first : 16u);
        cnt.cull(C_TILE, (unsigned long long) nt * n_us);
    }
}

// one workgroup per RT_PAINT_TILES tiles: background of the tiles classified EMPTY (src/update-cpu.cpp:93-95)
__device__ __forceinline__ void paint_tiles(const FrameArgs &fa, uint32_t *tile_state, uint32_t tag, uint32_t n_tiles, void *fb, const F3 &bg, uint32_t block,
                                            uint32_t wave, uint32_t lane)
{
    const uint32_t sub = lane >> 4, j0 = lane & 15u;
    const uint32_t t = block * RT_PAINT_TILES + wave * 4u + sub;
    const bool tv = t < n_tiles;
    uint32_t v = (tag << ST_BITS) | ST_NONEMPTY; // lanes without a word of their own
    if (j0 == 0 && tv) v = tile_word_wait(tile_state + t, tag, tile_word_load(tile_state + t));
    const unsigned long long em = __ballot(j0 == 0 && tv && (v & ST_MASK) == ST_EMPTY);
    if (!((em >> (16u * sub)) & 1ull)) return; // this lane's tile is traced (or handled) by a workgroup of its own
    const uint32_t tile_x = t % fa.tiles_x, tile_y = t / fa.tiles_x;
    const uint32_t x = tile_x * RT_TILE + j0, y0l = tile_y * RT_TILE;
    if (x >= fa.width) return;
    if (fa.rgba8) {
        uchar4 px;
        px.x = (unsigned char) (int) (bg.x * 255.0f + 0.5f);
        px.y = (unsigned char) (int) (bg.y * 255.0f + 0.5f);
        px.z = (unsigned char) (int) (bg.z * 255.0f + 0.5f);
        px.w = 255;
        uchar4 *dst = reinterpret_cast<uchar4 *>(fb) + (size_t) y0l * fa.width + x;
#pragma unroll
        for (uint32_t row = 0; row < RT_TILE; row++)
            if (y0l + row < fa.local_rows) dst[(size_t) row * fa.width] = px;
    } else {
        const float4 px = make_float4(bg.x, bg.y, bg.z, 1.0f);
        float4 *dst = reinterpret_cast<float4 *>(fb) + (size_t) y0l * fa.width + x;
#pragma unroll
        for (uint32_t row = 0; row < RT_TILE; row++)
            if (y0l + row < fa.local_rows) dst[(size_t) row * fa.width] = px;
    }
}

// Phase A.  PRIMARY: every lane's ray starts at the frame's origin, so the unit spheres are first culled against
// the wave's ray cone.  `S` may point at the LDS copy of the scene or (round 0, before anything is staged) at the
// blob in global memory: wave-uniform reads then become scalar loads, per-lane reads vector loads.
template <bool COUNT, bool HAS_GQ, bool HAS_CUBIC, bool PRIMARY>
__device__ __forceinline__ void nearest(const FrameArgs &fa, const SceneLds &S, const DevObject *__restrict__ gobj, const Mono &m,
                                        bool live, uint32_t lane, double &best_t, int &best, Cnt<COUNT> &cnt)
{
    best = -1;
    best_t = INFINITY;
    const bool quad = fabs(m.u2) > EPS; // unit spheres share t2 = u2: one degree decision per ray
    const double four_t2 = 4.0 * m.u2;
    D3 axis{0.0, 0.0, 1.0};
    double cos_t = 1.0;
    const bool cone = PRIMARY && fa.cull;
    if (cone) {
        // axis = direction of lane 36 (pixel (4, 4) of the wave's 8x8 block).  The angle to the axis is a
        // quasi-convex function on the image plane, so over the block it peaks at one of the four corner pixels
        // (lanes 0, 7, 56, 63): four v_readlane pairs instead of a 64-lane reduction.
        axis = D3{readlane_d(m.d.x, 36), readlane_d(m.d.y, 36), readlane_d(m.d.z, 36)};
        const double ca = dot3(axis, m.d);
        const double c0 = readlane_d(ca, 0), c1 = readlane_d(ca, 7), c2 = readlane_d(ca, 56), c3 = readlane_d(ca, 63);
        const double m01 = c0 < c1 ? c0 : c1, m23 = c2 < c3 ? c2 : c3;
        cos_t = m01 < m23 ? m01 : m23;
    }
    for (uint32_t base = 0; base < fa.n_us; base += 64) {
        const uint32_t end = (base + 64 < fa.n_us) ? base + 64 : fa.n_us;
        unsigned long long cand = 0;
        if (cone) {
            unsigned long long it = primary_cone_mask(S.us, base, end, m.o, axis, cos_t, lane);
            if (lane == 0) cnt.cull(C_PRIMARY, end - base);
            if (live) cnt.exec(K_US, (unsigned long long) __popcll(it));
            while (it) { // wave-uniform loop over the spheres that reach into this wave's cone
                const int b = __builtin_ctzll(it);
                it &= it - 1;
                const UsEntry e = S.us[base + b];
                const double t1 = us_t1(e, m);
                const double t0 = us_t0(e, m);
                const bool need = us_needs_solve(quad, four_t2, t1, t0);
                cand |= need ? (1ull << b) : 0ull;
            }
        } else {
            if (live) cnt.exec(K_US, end - base);
#pragma unroll 4
            for (uint32_t j = base; j < end; j++) { // wave-uniform: LDS broadcast reads, no branches
                const UsEntry e = S.us[j];
                const double t1 = us_t1(e, m);
                const double t0 = us_t0(e, m);
                const bool need = us_needs_solve(quad, four_t2, t1, t0);
                cand |= need ? (1ull << (j - base)) : 0ull;
            }
        }
        if (!live) cand = 0;
        while (cand) { // per lane: the few spheres whose root must actually be computed
            const int b = __builtin_ctzll(cand);
            cand &= cand - 1;
            const UsEntry e = S.us[base + b]; // LDS gather
            const double t1 = us_t1(e, m);
            const double t0 = us_t0(e, m);
            const double t = solve_quadlin(m.u2, t1, t0);
            cnt.solve(K_US);
            accept(t, (int) e.orig, best_t, best);
        }
    }
    for (uint32_t base = 0; HAS_GQ && base < fa.n_gq; base += 64) {
        const uint32_t end = (base + 64 < fa.n_gq) ? base + 64 : fa.n_gq;
        unsigned long long cand = 0;
        if (live) cnt.exec(K_GQ, end - base);
#pragma unroll 2
        for (uint32_t j = base; j < end; j++) {
            const GqEntry e = S.gq[j];
            const double t0 = gq_t0(e, m);
            cand |= needs_solve(gq_t2(e, m), gq_t1(e, m), t0) ? (1ull << (j - base)) : 0ull;
        }
        if (!live) cand = 0;
        while (cand) {
            const int b = __builtin_ctzll(cand);
            cand &= cand - 1;
            const GqEntry e = S.gq[base + b];
            const double t0 = gq_t0(e, m);
            const double t = solve_quadlin(gq_t2(e, m), gq_t1(e, m), t0);
            cnt.solve(K_GQ);
            accept(t, (int) e.orig, best_t, best);
        }
    }
    for (uint32_t j = 0; j < fa.n_lin; j++) { // planes: every lane needs the one division, nothing to defer
        const LinEntry e = S.lin[j];
        const double t1 = lin_t1(e, m);
        const double t0 = lin_t0(e, m);
        const double t = (fabs(t1) > EPS) ? -t0 / t1 : -1.0;
        if (live) {
            cnt.solve(K_LIN);
            cnt.exec(K_LIN, 1);
            accept(t, (int) e.orig, best_t, best);
        }
    }
    if (HAS_CUBIC) {
        for (uint32_t j = 0; j < fa.n_cub; j++) {
            const uint32_t k = __builtin_amdgcn_readfirstlane(S.cub[j]);
            if (live) {
                // F(o + t d) from the surface's Taylor data at the ray origin (rt_math.hpp: CubicAt, cubic_guarded), which waits in LDS: for
                // primary rays the origin is the frame's, and rt_render has evaluated the data of the first RT_CUB_AT_MAX degree-3 objects
                // there (S.cubprim, copied at staging); other rays form it in their lane's working record.  Where the guard refuses, the
                // reference's dense expansion and solver.
                uint32_t rec, abs = S.cubprim + RT_CUB_AT_MAX * (RT_CUB_REC * 8u) + j * 32u; // (bit 0 of an address: per-lane data, see cubic_test_body)
                if (PRIMARY && j < RT_CUB_AT_MAX) {
                    rec = S.cubprim + j * (RT_CUB_REC * 8u);
                } else {
                    rec = (S.cubtmp + (threadIdx.x << 3)) | 1u;
                    cubic_rec_store(S.cubtmp_p + threadIdx.x, WG, cubic_at(gobj[k].c, m.o));
                    if (j >= RT_CUB_AT_MAX) { // (beyond the objects whose CubicAbs the frame arguments carry: formed here, behind the lane's working record)
                        const CubicAbs ab = cubic_abs(gobj[k].c);
                        double *q = S.cubtmp_p + RT_CUB_REC * WG + threadIdx.x;
                        q[0] = ab.a3; q[WG] = ab.a2; q[2 * WG] = ab.a1; q[3 * WG] = ab.a0;
                        abs = (S.cubtmp + (threadIdx.x << 3) + RT_CUB_REC * WG * 8u) | 1u;
                    }
                    cnt.cubic_point();
                }
                double t;
                if (COUNT) { // counting builds also report which solver branch ran (flop accounting of bench.py): the guard answers only where the
                    int br;  // branch is beyond doubt, so the dense classification names it either way
                    bool refused;
                    t = cubic_test_body<false>(gobj[k].c, rec, abs, m.o.x, m.o.y, m.o.z, m.d.x, m.d.y, m.d.z, MAX_T, false, refused);
                    (void) intersect_cubic_branch(gobj[k].c, m.o.x, m.o.y, m.o.z, m.d.x, m.d.y, m.d.z, br);
                    cnt.cubic(br);
                    cnt.cubic_refused(refused);
                } else {
                    t = cubic_test(gobj[k].c, rec, abs, m.o.x, m.o.y, m.o.z, m.d.x, m.d.y, m.d.z, MAX_T, false);
                }
                accept(t, (int) k, best_t, best);
            }
        }
    }
}

template <bool SPHERICAL, typename Light>
__device__ __forceinline__ unsigned long long relevant_mask(const UsEntry *us, uint32_t base, uint32_t end, const Ball &ball,
                                                            const Light &lt, uint32_t lane)
{
    const uint32_t j = base + lane;
    return __ballot(j < end && sphere_relevant<SPHERICAL>(us[j], ball, lt));
}

// directional lights, first group of spheres: from the chunk's culling records; the light's direction is the ray direction the
// caller already holds (include/light_impl.h:23-25: the same for every hit)
__device__ __forceinline__ unsigned long long relevant_mask_directional(const CullRec *crec, uint32_t end, const D3 &sdir, double inv_uu, double len_u, uint32_t lane)
{
    return __ballot(lane < end && crec_relevant(crec[lane], sdir, inv_uu, len_u));
}

// Survivors of the ball test from which the box test (rt_wavefront_math.hpp) is run as well.  A chunk whose hits lie on a near and a
// far object has a long thin box and a fat ball: every sphere near the axis survives the ball (12.7 per item in the slowest tile of
// orbit pose 6, against 1.2 on average), each costing 27 instructions per wave to find that no lane needs it.  Same-box A/B
// (1080p, us): threshold 3: start 46.0 -> 46.9, orbit pose 6 67.2 -> 54.4, pose 19 58.4 -> 61.3;  6: 46.2, 54.5, 59.5;  9: no gain.
constexpr int BOX_STAGE_MIN = 6;
constexpr uint32_t BOX_EVAL_UNITS = 3; // a box-stage evaluation in units of a directional decision, for the work counters (tools/count_flops.cpp checks the ratio) // survivors of the ball test from which the box test is worth its ~30 instructions
constexpr uint32_t CREC_MAX = 64; // culling records cover the first group of 64 spheres; further groups take relevant_mask

// LDS carve-up (dynamic shared memory), shared by kernel and launcher.
struct LdsLayout {
    uint32_t scene, light, hp, hn, hdir, park, hidx, hpix, color, shadow, ball, box, crec, n_crec, misc, cubrec, cubtmp, cubprim, total, shadow_words;
    __host__ __device__ LdsLayout(uint32_t scene_bytes, uint32_t n_lights, bool has_mirror, uint32_t n_cull_spheres, bool lean = false, uint32_t n_cub = 0)
    {
        const uint32_t Q = lean ? 0u : WG; // the lean path keeps a wave's hits in registers: no queue, no shadow bits, no colour exchange
        n_crec = n_cull_spheres < CREC_MAX ? n_cull_spheres : CREC_MAX; // culling records per chunk (0: culling is off)
        shadow_words = (n_lights + 31) / 32;
        if (shadow_words == 0) shadow_words = 1;
        uint32_t off = 0;
        scene = off; off = align16(off + scene_bytes);
        light = off; // (no LDS copy of the lights any more: every path reads them through scalar loads, LightK)
        hp = off; off = align16(off + 3 * WG * 8); // (the lean path parks the hit point of each pixel here across the light loop: point lights only read it)
        hn = off; off = align16(off + 3 * Q * 8);
        hdir = off; off = align16(off + (has_mirror ? 3 * WG * 8 : 0)); // mirrors only: the pixel's incoming direction ...
        park = off; off = align16(off + (has_mirror ? 5 * WG * 4 : 0)); // ... and its running colour / ratio / depth, parked across B and C
        hidx = off; off = align16(off + Q * 4); // (object << 8) | owner lane
        hpix = hidx;
        color = off; off = align16(off + 3 * Q * 4);
        shadow = off; off = align16(off + (lean ? 0u : 4u * (n_lights ? n_lights : 1u) * 8u)); // one 64-bit mask per (chunk, light): hits that skip the light when shading
        ball = off; off = align16(off + 4 * (uint32_t) sizeof(Ball));
        box = off; off = align16(off + 4 * (uint32_t) sizeof(BoxH));
        crec = off; off = align16(off + 4 * n_crec * (uint32_t) sizeof(CullRec));
        misc = off; off = align16(off + 64);
        // degree-3 scenes: RT_CUB_REC doubles per record (rt_math.hpp: CubicAt) -- one per hit, at its shadow-ray origin (first degree-3 object;
        // SoA, stride WG), one per lane as working space (+ 4 for a CubicAbs: rays that do not start at the frame's origin, further degree-3
        // objects -- only scenes with mirrors or several such objects), and RT_CUB_AT_MAX packed ones at the frame's origin, followed by
        // the CubicAbs of those objects.  (The per-hit records are what decides how many workgroups a CU holds: 20 KB.)
        cubrec = off; off = align16(off + (n_cub ? RT_CUB_REC * WG * 8u : 0u));
        cubtmp = off; off = align16(off + ((n_cub > 1u || (n_cub && has_mirror)) ? (RT_CUB_REC + 4u) * WG * 8u : 0u));
        cubprim = off; off = align16(off + (n_cub ? (RT_CUB_REC + 4u) * RT_CUB_AT_MAX * 8u : 0u));
        total = off;
    }
};

constexpr int NO_BLOCKER = 0x7fffffff;

// Phase B for one (chunk, light) item: is the lane's shadow ray blocked, and (COUNT builds) by which lowest
// object index.  `blocker` keeps the lowest blocking index seen; without COUNT any blocker ends the search.
//
// OWN (the lean path, all-sphere scenes): `own` is the table index of the sphere the lane's hit lies on, and `own_skip` says that the
// reference's test of this shadow ray against that very sphere is known to find no blocker (own_sphere_skippable in the kernel has the
// argument), so the lane sits that sphere out; `own_excl` (wave-uniform) has the bit of a sphere that EVERY tested lane may skip.
template <bool COUNT, bool HAS_GQ, bool HAS_CUBIC, bool SPHERICAL, bool OWN = false, typename Light = DevLight> // SPHERICAL: the light's kind -- one copy of the loop per kind, each without the other's code
__device__ __forceinline__ int shadow_blocker(const FrameArgs &fa, const SceneLds &S, const DevObject *__restrict__ gobj,
                                              const Mono &sm, double max_t, bool valid, bool prod, const Ball *ballp, const BoxH *boxp, const CullRec *crec, const Light &lt,
                                              uint32_t lane, Cnt<COUNT> &cnt, uint32_t own = 0, bool own_skip = false, unsigned long long own_excl = 0ull, uint32_t cub_rec0 = 0u)
{
    // valid: lanes whose ray is tested.  prod: lanes the product build tests (== valid there); a counting build tests more lanes --
    // all that have a hit -- and counts executed work for the `prod` ones only (wave-level work: if any lane is one).
    const bool prod_any = COUNT ? (bool) __any(prod) : true;
    int blocker = NO_BLOCKER;
    // (a directional light's rays share their direction: t2 is the same in every lane, and said so the branch on it is a scalar one)
    const bool quad = SPHERICAL ? fabs(sm.u2) > EPS : __builtin_amdgcn_readfirstlane((int) (fabs(sm.u2) > EPS)) != 0;
    double four_t2 = 4.0 * sm.u2;
    if constexpr (OWN && !SPHERICAL) four_t2 = lt.four_u2; // (the light table has it: the same product, in SGPRs)
    for (uint32_t base = 0; base < fa.n_us; base += 64) {
        const uint32_t end = (base + 64 < fa.n_us) ? base + 64 : fa.n_us;
        unsigned long long cand = 0;
        if (fa.cull) {
            unsigned long long it;
            if (!SPHERICAL && base == 0) {
                it = relevant_mask_directional(crec, end, sm.d, lt.inv_uu, lt.len_u, lane); // sm.d is this light's FP32-rounded direction
                if (__popcll(it) >= BOX_STAGE_MIN) { // wave-uniform: many got through the ball -- look again with the box
                    if constexpr (OWN) it &= __ballot(lane < end && crec_in_box_shadow(crec[lane], *boxp, sm.d, lt.s_yz, lt.s_xz, lt.s_xy));
                    else it &= __ballot(lane < end && crec_in_box_shadow(crec[lane], *boxp, sm.d));
                    if (lane == 0 && prod_any) cnt.cull(C_SHADOW_DIR, BOX_EVAL_UNITS * (end - base)); // 30 counted operations: three directional decisions
                }
            }
            else it = relevant_mask<SPHERICAL>(S.us, base, end, *ballp, lt, lane); // (the ball is read from LDS here, not kept)
            if (lane == 0 && prod_any) cnt.cull(SPHERICAL ? C_SHADOW_SPH : C_SHADOW_DIR, end - base);
            if (OWN && base == 0) it &= ~own_excl;
            if (!OWN && prod) cnt.exec(K_US, (unsigned long long) __popcll(it));
            while (it) { // wave-uniform loop over the spheres that survived the culling
                const int b = __builtin_ctzll(it);
                it &= it - 1;
                bool act = valid;
                if (OWN) { // lanes whose own sphere this is sit it out; nobody left: next sphere
                    act = valid && !(own_skip && own == base + (uint32_t) b);
                    if (!__any(act)) continue;
                    if (prod && act) cnt.exec(K_US, 1);
                }
                const UsEntry e = S.us[base + b];
                const double t1 = us_t1(e, sm), t0 = us_t0(e, sm);
                const bool need = act && us_needs_solve(quad, four_t2, t1, t0);
                cand |= need ? (1ull << b) : 0ull;
            }
        } else {
            if (prod) cnt.exec(K_US, end - base);
#pragma unroll 4
            for (uint32_t j = base; j < end; j++) {
                const UsEntry e = S.us[j];
                const double t1 = us_t1(e, sm), t0 = us_t0(e, sm);
                const bool need = us_needs_solve(quad, four_t2, t1, t0);
                cand |= need ? (1ull << (j - base)) : 0ull;
            }
        }
        if (!valid) cand = 0;
        while (cand) {
            const int b = __builtin_ctzll(cand);
            cand &= cand - 1;
            const UsEntry e = S.us[base + b];
            if ((int) e.orig > blocker) continue;
            const double t = solve_quadlin(sm.u2, us_t1(e, sm), us_t0(e, sm));
            if (prod) cnt.solve(K_US);
            if (t > EPS && t < max_t) { // src/update-cpu.cpp:68
                blocker = (int) e.orig;
                if (!COUNT) break;
            }
        }
    }
    for (uint32_t base = 0; HAS_GQ && base < fa.n_gq; base += 64) {
        const uint32_t end = (base + 64 < fa.n_gq) ? base + 64 : fa.n_gq;
        unsigned long long cand = 0;
        if (prod) cnt.exec(K_GQ, end - base);
#pragma unroll 2
        for (uint32_t j = base; j < end; j++) {
            const GqEntry e = S.gq[j];
            cand |= needs_solve(gq_t2(e, sm), gq_t1(e, sm), gq_t0(e, sm)) ? (1ull << (j - base)) : 0ull;
        }
        if (!valid || (!COUNT && blocker != NO_BLOCKER)) cand = 0;
        while (cand) {
            const int b = __builtin_ctzll(cand);
            cand &= cand - 1;
            const GqEntry e = S.gq[base + b];
            if ((int) e.orig > blocker) continue;
            const double t = solve_quadlin(gq_t2(e, sm), gq_t1(e, sm), gq_t0(e, sm));
            if (prod) cnt.solve(K_GQ);
            if (t > EPS && t < max_t) {
                blocker = (int) e.orig;
                if (!COUNT) break;
            }
        }
    }
    for (uint32_t j = 0; j < fa.n_lin; j++) {
        const LinEntry e = S.lin[j];
        const double t1 = lin_t1(e, sm), t0 = lin_t0(e, sm);
        const double t = (fabs(t1) > EPS) ? -t0 / t1 : -1.0;
        if (valid) {
            if (prod) cnt.solve(K_LIN);
            if (prod) cnt.exec(K_LIN, 1);
            if (t > EPS && t < max_t && (int) e.orig < blocker) blocker = (int) e.orig;
        }
    }
    if (HAS_CUBIC) {
        for (uint32_t j = 0; j < fa.n_cub; j++) {
            const uint32_t k = __builtin_amdgcn_readfirstlane(S.cub[j]);
            if (valid && (int) k < blocker && (COUNT || blocker == NO_BLOCKER)) {
                // the surface's Taylor data at the shadow-ray origin is the same for every light: phase A' has put the first degree-3 object's
                // into the hit's record (cub_rec0: its LDS address); further objects form theirs in the lane's working record
                uint32_t rec = cub_rec0 | 1u, abs = S.cubprim + RT_CUB_AT_MAX * (RT_CUB_REC * 8u) + j * 32u;
                if (j != 0) {
                    rec = (S.cubtmp + (threadIdx.x << 3)) | 1u;
                    cubic_rec_store(S.cubtmp_p + threadIdx.x, WG, cubic_at(gobj[k].c, sm.o));
                    if (j >= RT_CUB_AT_MAX) {
                        const CubicAbs ab = cubic_abs(gobj[k].c);
                        double *q = S.cubtmp_p + RT_CUB_REC * WG + threadIdx.x;
                        q[0] = ab.a3; q[WG] = ab.a2; q[2 * WG] = ab.a1; q[3 * WG] = ab.a0;
                        abs = (S.cubtmp + (threadIdx.x << 3) + RT_CUB_REC * WG * 8u) | 1u;
                    }
                    if (prod) cnt.cubic_point();
                }
                double t;
                if (COUNT) {
                    int br;
                    bool refused;
                    t = cubic_test_body<false>(gobj[k].c, rec, abs, sm.o.x, sm.o.y, sm.o.z, sm.d.x, sm.d.y, sm.d.z, max_t, true, refused);
                    (void) intersect_cubic_branch(gobj[k].c, sm.o.x, sm.o.y, sm.o.z, sm.d.x, sm.d.y, sm.d.z, br);
                    cnt.cubic(br, prod);
                    if (prod) cnt.cubic_refused(refused);
                } else {
                    t = cubic_test(gobj[k].c, rec, abs, sm.o.x, sm.o.y, sm.o.z, sm.d.x, sm.d.y, sm.d.z, max_t, true);
                }
                if (t > EPS && t < max_t) blocker = (int) k;
            }
        }
    }
    return blocker;
}

struct HotArgs { // the preloaded prefix of the kernel arguments, for the offset of what follows
    const unsigned char *hot_us;
    const uint32_t *hot_ord_rd;
    uint32_t hot_n_us, hot_ord_cap, hot_n_tiles, hot_flags;
    uint32_t *hot_tile_state;
    uint32_t hot_frame_tag, hot_n_scan;
};
struct ColdArgs {
    FrameArgs fa;
    const unsigned char *gscene;
    const DevLight *glight;
    void *fb;
    unsigned long long *counters;
    const double *camx, *camy;
};
static_assert(sizeof(HotArgs) == 48 && alignof(ColdArgs) == 8, "kernel argument layout");
// the cold arguments where they sit in the kernarg segment (constant address space: uniform addresses become scalar loads); the
// pointer goes through an empty asm so that no load from it can be moved above this point
__device__ __forceinline__ const ColdArgs &cold_args()
{
    typedef const __attribute__((address_space(4))) unsigned char *KernargBytes;
    KernargBytes p = (KernargBytes) __builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return *(const ColdArgs *) (p + sizeof(HotArgs));
}

// The lean path's shadow test for ONE directional light with |d|^2 > EPS (product builds, scenes of at most 64 unit spheres, all of them
// cullable): is the lane's shadow ray blocked?  The same decisions as shadow_blocker<.., SPHERICAL = false, OWN = true> -- same culling
// records, same box stage, same own-sphere rule, same reference arithmetic per surviving sphere -- laid out for the common case, in
// which no sphere survives the culling: nothing that only the survivors need (the ray's t1 part, the discriminant's factor) is formed
// before it is known that there are any, the light's constants are scalar operands (LightK), and there is no loop over sphere groups.
__device__ __forceinline__ bool lean_dir_blocked(const UsEntry *us, const CullRec *crec, const BoxH *boxp, const D3 &o, double u0, const LightK &lk, bool wanted,
                                                 uint32_t own, bool own_skip, unsigned long long own_excl, uint32_t n_us, uint32_t lane)
{
    const D3 d{lk.sdir[0], lk.sdir[1], lk.sdir[2]};
    unsigned long long it = __ballot(lane < n_us && crec_relevant(crec[lane], d, lk.inv_uu, lk.len_u));
    if (__popcll(it) >= BOX_STAGE_MIN) it &= __ballot(lane < n_us && crec_in_box_shadow(crec[lane], *boxp, d, lk.s_yz, lk.s_xz, lk.s_xy));
    it &= ~own_excl;
    if (it == 0ull) return false; // wave-uniform: nothing can block any ray of this block
    // the ray's part of t1 (mono_set_od: 2 o.d summed as the reference sums it)
    const double u1 = (2.0 * o.x * d.x + 2.0 * o.y * d.y) + 2.0 * o.z * d.z;
    unsigned long long cand = 0ull;
    do { // wave-uniform loop over the spheres that survived the culling
        const uint32_t b = (uint32_t) __builtin_ctzll(it);
        it &= it - 1ull;
        const bool act = wanted && !(own_skip && own == b); // lanes whose own sphere this is sit it out (own_sphere_skippable)
        if (!__any(act)) continue;
        const UsEntry e = us[b];
        const double t1 = ((u1 + e.kx * d.x) + e.ky * d.y) + e.kz * d.z;           // us_t1
        const double t0 = (((u0 + e.kx * o.x) + e.ky * o.y) + e.kz * o.z) + e.c;   // us_t0
        const bool need = act && us_needs_solve(true, lk.four_u2, t1, t0);
        cand |= need ? (1ull << b) : 0ull;
    } while (it != 0ull);
    bool blocked = false;
    while (cand != 0ull) { // per lane: the few spheres whose root must actually be computed
        const uint32_t b = (uint32_t) __builtin_ctzll(cand);
        cand &= cand - 1ull;
        const UsEntry e = us[b]; // LDS gather
        const double t1 = ((u1 + e.kx * d.x) + e.ky * d.y) + e.kz * d.z;
        const double t0 = (((u0 + e.kx * o.x) + e.ky * o.y) + e.kz * o.z) + e.c;
        const double t = solve_quadlin(lk.u2, t1, t0);
        if (t > EPS && t < 1e6) { // src/update-cpu.cpp:68, max_t of a directional light (include/light_impl.h:24)
            blocked = true;
            break;
        }
    }
    return blocked;
}

// ---- the lean path: one wave renders one 8 x 8 block, start to finish (LEAN instantiations only) ----
// Scenes of unit spheres without mirrors (BASELINE configs 2 and 5).  A wave keeps the hits of its own 64 pixels IN REGISTERS -- no
// compaction into a queue of the tile, hence no barrier after the staging one, no shadow bits, no colour exchange -- and walks the
// lights IN ORDER, adding each unshadowed light's term to the pixel's colour as soon as its shadow test is done
// (src/update-cpu.cpp:62-77 is that very loop).  What the queue bought (full lanes in the shadow phase of tiles on a silhouette) is
// small where spheres cover many blocks: 1080p has 3.5 % more (block, light) items this way.  What it cost: five barriers per tile,
// 12 KB of LDS, the one-lane-per-hit shading phase and its second pass over the normals and lights, the bits in between.
// In: the lane's primary direction and whether its pixel is inside the image; `tile_` only labels the stamps.  Out: the pixel's colour,
// the block's hit mask, the shader-clock ticks spent on its lights (`listing_`: the launch order wants them).
struct LeanLds { // the LDS regions of the lean path: `hp` one slot per thread ([3][hp_stride]), the others one per wave
    double *hp;
    uint32_t hp_stride;
    Ball *sball;
    BoxH *sbox;
    CullRec *screc;
    const unsigned char *smem;
    uint32_t crec_off, n_crec;
};
template <bool COUNT>
__device__ __forceinline__ void lean_block(const FrameArgs &fa, const SceneLds &S, const DevObject *__restrict__ gobj, const LeanLds &M, const DevLight *glight, Cnt<COUNT> &cnt,
                                           uint32_t tile_, const D3 &o, const D3 &dir, bool live, bool listing_, const uint32_t lane, const uint32_t tid, const uint32_t wave,
                                           F3 &res, unsigned long long &hm, unsigned long long &b_ticks RT_STAMP_ARG)
{
    (void) tile_;
    double *hp = M.hp;
    Ball *sball = M.sball;
    BoxH *sbox = M.sbox;
    CullRec *screc = M.screc;
    const unsigned char *smem = M.smem;
    struct { uint32_t crec, n_crec; } L{M.crec_off, M.n_crec};
    ConstLights clight = (ConstLights) (glight + fa.n_lights);
            if (live) cnt.add(0);
            cnt.primary_traced();
            RT_STAMP(1);
            double best_t;
            int best;
            {
                Mono m;
                mono_set_o<false>(m, o);
                mono_set_d<false>(m, dir);
                mono_set_od<false>(m);
                nearest<COUNT, false, false, true>(fa, S, gobj, m, live, lane, best_t, best, cnt);
            }
            RT_STAMP(2);
            if (live) cnt.add(3, fa.n_obj);
            const bool hit = live && best >= 0;
            hm = __ballot(hit);
            RT_STAMP_INFO(((unsigned long long) tile_ << 32) | (unsigned) __popcll(hm));
            b_ticks = 0ull;
            if (hm != 0ull) { // wave-uniform: this block has hits
                const uint32_t bi = hit ? (uint32_t) best : 0u; // (every object is a unit sphere: table index == object index)
                const UsEntry eo = S.us[bi];
                const D3 sp{o.x + best_t * dir.x, o.y + best_t * dir.y, o.z + best_t * dir.z};
                const D3 sn = sphere_normal(eo, sp);
                if (hit) cnt.add(4);
                hp[tid] = sp.x; hp[M.hp_stride + tid] = sp.y; hp[2 * M.hp_stride + tid] = sp.z; // this lane's own slot: only point lights need the hit point again
                Mono sm;
                mono_set_o<false>(sm, D3{sp.x + SHADOW_BIAS * sn.x, sp.y + SHADOW_BIAS * sn.y, sp.z + SHADOW_BIAS * sn.z});
                // own_sphere_skippable.  The shadow ray of a hit on sphere s starts at o = p + 1e-2 n, outside s, and the reference tests it
                // against s like against any other object (src/update-cpu.cpp:66-71).  Its t0 = F_s(o) does not depend on the light, so it
                // is formed here once, with the reference's operations (us_t0).  For a directional light in front of the surface
                // ((float) dot(n, l) > 0: the lanes the product traces at all) with |d|^2 > EPS that test cannot find a root > EPS while
                // own_lo < t0 < own_hi:
                //   * its computed t1 is 2 rho (n.d) up to rounding, rho = |o - centre|, and n.d >= -1.1e-7 |d| (dot(n, l) > 0 up to 3 ulp;
                //     d is l rounded to FP32), so either t1 > 0 -- then, t0 being > 0, both roots are <= 0 (us_needs_solve) --
                //   * or t1^2 <= 2 (2.2e-7 rho |d|)^2 + 2 (2.7e-15 S |d|)^2  with rho^2 = t0 + r^2 < 3 (r^2 + 1)  (t0 < own_hi = (r + 1)^2)  and
                //     S = |o|_1 + |centre|_1 <= 2 |centre|_1 + 3 r + 3, which is below 4 |d|^2 t0 as soon as t0 > 1e-13 (r^2 + 1) + 1e-29 S^2:
                //     the discriminant is negative and the solver returns -1 (include/surface_impl.h:141-144).
                // own_lo = 1e-10 (r^2 + 1) + 1e-20 S^2 keeps three orders of magnitude between the two (rt_create); a true t0 is
                // 0.02 r + 1e-4.  Lanes outside the window simply take the test.
                const double t0_own = us_t0(eo, sm);
                const bool skip_geom = hit && t0_own > (double) eo.own_lo && t0_own < (double) eo.own_hi;
                const uint32_t b0 = (uint32_t) __builtin_amdgcn_readlane((int) bi, __builtin_ctzll(hm));
                // every hit of the block on one sphere, all inside the window: the sphere's bit, cleared from the culling verdict at once
                const unsigned long long own_all = (__ballot(hit && (bi != b0 || !skip_geom)) == 0ull && b0 < 64u) ? (1ull << b0) : 0ull;

                // ---- the block's bounding box / ball and its culling records (phase A' of the other path), private to this wave ----
                {
                    const float inf = __builtin_inff();
                    const float lox = wave_min_f(hit ? __double2float_rd(sp.x) : inf), hix = wave_max_f(hit ? __double2float_ru(sp.x) : -inf);
                    const float loy = wave_min_f(hit ? __double2float_rd(sp.y) : inf), hiy = wave_max_f(hit ? __double2float_ru(sp.y) : -inf);
                    const float loz = wave_min_f(hit ? __double2float_rd(sp.z) : inf), hiz = wave_max_f(hit ? __double2float_ru(sp.z) : -inf);
                    const double dx = (double) hix - (double) lox, dy = (double) hiy - (double) loy, dz = (double) hiz - (double) loz;
                    Ball b;
                    b.cx = 0.5 * ((double) lox + (double) hix);
                    b.cy = 0.5 * ((double) loy + (double) hiy);
                    b.cz = 0.5 * ((double) loz + (double) hiz);
                    b.R = 0.5 * sqrt(dx * dx + dy * dy + dz * dz) * (1.0 + 1e-9) + 1.01e-2; // half diagonal (rounded up) + the shadow bias of the ray origins
                    if (lane == 0) {
                        sball[wave] = b;
                        sbox[wave] = BoxH{0.5 * dx * (1.0 + 1e-9) + 1.01e-2, 0.5 * dy * (1.0 + 1e-9) + 1.01e-2, 0.5 * dz * (1.0 + 1e-9) + 1.01e-2, 0.0};
                    }
                    if (lane < L.n_crec) screc[wave * L.n_crec + lane] = cull_record(S.us[lane], b);
                    if (lane == 0) cnt.cull(C_RECORDS, L.n_crec);
                    // (written and read by this wave only; LDS executes a wave's accesses in order)
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
                }
                RT_STAMP(5);

                // ---- lights in order: shadow test, then the light's term (src/update-cpu.cpp:62-77) ----
                const unsigned long long b_t0 = listing_ ? __builtin_amdgcn_s_memtime() : 0ull;
                const CullRec *my_crec = reinterpret_cast<const CullRec *>(smem + (L.crec + wave * L.n_crec * (uint32_t) sizeof(CullRec)));
                // The lights through the constant address space (LightK, rt_scene_dev.h): one batch of scalar loads per light, its fields in
                // SGPRs -- no vector register holds a per-light constant; the LDS copy of the lights is not used by this path.  One body
                // per light kind, with nothing merged between them (a value that is uniform in one and per lane in the other would become a
                // vector register in both).
                const Ball *my_ball = sball + wave;
                const BoxH *my_box = sbox + wave;
                // Point lights first, shadow tests only: one bit per (lane, point light) in a register.  Their culling test is the widest
                // stretch of this kernel in registers; in a pass of its own it does not meet the colour accumulators and constants of
                // the ordered pass below, which only reads the bit when the light's turn comes (include/light_impl.h:19-21: (light -
                // point) through FP32, max_t = 1).
                const unsigned long long pt_mask = ((unsigned long long) fa.pt_mask[1] << 32) | fa.pt_mask[0];
                unsigned long long pt_blocked = 0ull; // bit l: point light l is blocked for this lane's hit
                for (unsigned long long pm = pt_mask; pm != 0ull; pm &= pm - 1ull) { // wave-uniform
                    const uint32_t l = (uint32_t) __builtin_ctzll(pm);
                    RT_LOAD_LIGHTK(lk, l)
                    const D3 spl{hp[tid], hp[M.hp_stride + tid], hp[2 * M.hp_stride + tid]};
                    const double dx = lk.p[0] - spl.x, dy = lk.p[1] - spl.y, dz = lk.p[2] - spl.z;
                    const double q = dot3(sn, D3{dx, dy, dz});
                    const double mag = fabs(sn.x * dx) + fabs(sn.y * dy) + fabs(sn.z * dz);
                    const bool wanted = hit && !(q < -1e-9 * mag); // behind the surface by a margin 10^7 times the rounding: the term is +0 (as in the other path)
                    if (COUNT || __any(wanted)) {
                        Mono sd;
                        sd.o = sm.o; sd.u0 = sm.u0;
                        mono_set_d<false>(sd, D3{(double) (float) dx, (double) (float) dy, (double) (float) dz});
                        mono_set_od<false>(sd);
                        if (wanted) cnt.traced();
                        const int blocker = shadow_blocker<COUNT, false, false, true, true>(fa, S, gobj, sd, 1.0, COUNT ? hit : wanted, wanted, my_ball, my_box, my_crec, lk, lane, cnt);
                        if (hit) cnt.add(3, blocker != NO_BLOCKER ? (unsigned long long) blocker + 1ull : (unsigned long long) fa.n_obj);
                        if (hit && blocker != NO_BLOCKER) pt_blocked |= 1ull << l;
                    } else if (COUNT && hit) {
                        cnt.add(3, fa.n_obj);
                    }
                }
                // ---- lights in order (src/update-cpu.cpp:62-77) ----
                const MatEntry mt = S.mat[bi];
                const F3 aop{mt.albedo[0] / PI_F, mt.albedo[1] / PI_F, mt.albedo[2] / PI_F}; // object_color / pi, once per hit
                F3 acc{0.0f, 0.0f, 0.0f};
                if (!COUNT && fa.lights_plain && fa.n_us <= 64u) { // launch-uniform: product build, every directional light has finite colours and |d|^2 > EPS
                    // The loop below (which handles every combination) specialised: no per-light flags to look at, the light-behind-the-surface
                    // test straight out of one compare (lanes without a hit carry a zero normal: (float) dot = 0, not in front), the shadow test in
                    // lean_dir_blocked.  Same values, same order.
                    const D3 snz{hit ? sn.x : 0.0, hit ? sn.y : 0.0, hit ? sn.z : 0.0};
                    for (uint32_t l = 0; l < fa.n_lights; l++) {
                        RT_LOAD_LIGHTK(lk, l)
                        if (!((pt_mask >> l) & 1ull)) { // directional: include/light_impl.h:23-25
                            const float lam = (float) dot3(snz, D3{lk.p[0], lk.p[1], lk.p[2]}); // surface_color's (float) dot(n, dir), include/light_impl.h:42
                            const bool wanted = 0.0f < lam; // a light behind the surface adds exactly +0, shadowed or not: such lanes sit the light out
                            if (__any(wanted)) {
                                const bool blocked = lean_dir_blocked(S.us, my_crec, my_box, sm.o, sm.u0, lk, wanted, bi, skip_geom, own_all, fa.n_us, lane);
                                if (wanted && !blocked) { // ((albedo / pi) * colour) * max(0, n.l), left to right (include/light_impl.h:43); max(0, lam) = lam here
                                    acc.x += aop.x * lk.color[0] * lam;
                                    acc.y += aop.y * lk.color[1] * lam;
                                    acc.z += aop.z * lk.color[2] * lam;
                                }
                            }
                        } else if (hit && !((pt_blocked >> l) & 1ull)) { // point light, not blocked
                            const D3 spl{hp[tid], hp[M.hp_stride + tid], hp[2 * M.hp_stride + tid]};
                            const double lp[3] = {lk.p[0], lk.p[1], lk.p[2]};
                            const float lc[3] = {lk.color[0], lk.color[1], lk.color[2]};
                            const F3 col = surface_color_pre(lp, lc, true, spl, sn, aop);
                            acc.x += col.x;
                            acc.y += col.y;
                            acc.z += col.z;
                        }
                    }
                } else
                for (uint32_t l = 0; l < fa.n_lights; l++) {
                    RT_LOAD_LIGHTK(lk, l)
                    if (hit) cnt.add(1);
                    if (!((pt_mask >> l) & 1ull)) { // ---- directional: include/light_impl.h:23-25, one direction for every ray ----
                        const bool bfe = (lk.flags & 2u) != 0u;
                        const float lam = (float) dot3(sn, D3{lk.p[0], lk.p[1], lk.p[2]}); // surface_color's (float) dot(n, dir), include/light_impl.h:42
                        // a light behind the surface adds exactly +0 (all colours finite: bfe), shadowed or not: such lanes sit the light out
                        const bool wanted = hit && (!bfe || 0.0f < lam);
                        if (COUNT || __any(wanted)) {
                            const bool quad_l = (lk.flags & 4u) != 0u; // (|d|^2 <= EPS: the reference takes its linear branch, to which the own-sphere argument does not apply)
                            if (wanted) cnt.traced();
                            int blocker;
                            if (!COUNT && bfe && quad_l && fa.n_us <= 64u) { // wave-uniform: the common case, in its own lay-out (same decisions)
                                blocker = lean_dir_blocked(S.us, my_crec, my_box, sm.o, sm.u0, lk, wanted, bi, skip_geom, own_all, fa.n_us, lane) ? 0 : NO_BLOCKER;
                            } else {
                                Mono sd;
                                sd.o = sm.o; sd.u0 = sm.u0;
                                sd.d = D3{lk.sdir[0], lk.sdir[1], lk.sdir[2]};
                                sd.u2 = lk.u2;
                                mono_set_od<false>(sd);
                                blocker = shadow_blocker<COUNT, false, false, false, true>(fa, S, gobj, sd, 1e6, COUNT ? hit : wanted, wanted, my_ball, my_box, my_crec, lk, lane, cnt, bi,
                                                                                           bfe && quad_l && wanted && skip_geom, (!COUNT && bfe && quad_l) ? own_all : 0ull);
                            }
                            if (hit) cnt.add(3, blocker != NO_BLOCKER ? (unsigned long long) blocker + 1ull : (unsigned long long) fa.n_obj);
                            if (wanted && blocker == NO_BLOCKER) { // ((albedo / pi) * colour) * max(0, n.l), left to right (include/light_impl.h:43)
                                const float mx = (0.0f < lam) ? lam : 0.0f;
                                cnt.shaded();
                                acc.x += aop.x * lk.color[0] * mx;
                                acc.y += aop.y * lk.color[1] * mx;
                                acc.z += aop.z * lk.color[2] * mx;
                            }
                        }
                    } else if (hit && !((pt_blocked >> l) & 1ull)) { // ---- point light, not blocked (lanes that sat its test out add their +0 like the other path does) ----
                        const D3 spl{hp[tid], hp[M.hp_stride + tid], hp[2 * M.hp_stride + tid]};
                        const double lp[3] = {lk.p[0], lk.p[1], lk.p[2]};
                        const float lc[3] = {lk.color[0], lk.color[1], lk.color[2]};
                        const F3 col = surface_color_pre(lp, lc, true, spl, sn, aop);
                        cnt.shaded();
                        acc.x += col.x;
                        acc.y += col.y;
                        acc.z += col.z;
                    }
                }
                if (hit) { // glm::min(vec3(1.0f), acc), src/update-cpu.cpp:77
                    res.x = (acc.x < 1.0f) ? acc.x : 1.0f;
                    res.y = (acc.y < 1.0f) ? acc.y : 1.0f;
                    res.z = (acc.z < 1.0f) ? acc.z : 1.0f;
                }
                if (listing_) b_ticks = __builtin_amdgcn_s_memtime() - b_t0;
                RT_STAMP(6);
            }
    }

// HAS_MIRROR = some object has reflection_ratio > EPS.  Without mirrors every pixel is finished after round 0, the
// round loop is known to run once and the bounce state (ray direction, blend ratio, depth) is dead during the shadow
// phase -- which is what lets the mirror-free instantiation fit 128 VGPRs (4 waves per SIMD).
//
// LEAN = the wave-per-block path for scenes of unit spheres only (no mirrors, dense output): see "the lean path" below.
template <bool COUNT, bool HAS_GQ, bool HAS_CUBIC, bool HAS_MIRROR, bool LEAN = false>
__global__ __launch_bounds__(256, (wf_occupancy<COUNT, HAS_GQ, HAS_CUBIC, HAS_MIRROR, LEAN>())) void wavefront_tile_kernel(
    // The first twelve dwords of the kernel arguments are PRELOADED into SGPRs at wave launch (kernarg preload, gfx940+; the
    // object is built with -amdgpu-kernarg-preload-count=12): with them a workgroup decides what it is in its very first
    // instructions -- an index slot reads its tile's word and, nine times out of ten, leaves without ever fetching the rest.
    // They repeat values of `fa`; rt_launch_wavefront fills both.
    const unsigned char *hot_us,      // gscene + fa.off_us: the unit-sphere table
    const uint32_t *hot_ord_rd,       // the launch-order generation this frame reads (fa.order_state + ord_read * ord_stride)
    uint32_t hot_n_us, uint32_t hot_ord_cap, uint32_t hot_n_tiles,
    uint32_t hot_flags,               // 1: fa.all_cullable   2: launch-order lists in use (fa.order_state && fa.ord_on)   4: paint workgroups   8: list slots take the entries in list order (A/B)
    uint32_t *hot_tile_state,         // fa.tile_state (NULL: no tile words in this launch: every tile's own workgroup decides)
    uint32_t hot_frame_tag, uint32_t hot_n_scan, // fa.frame_tag, fa.n_scan (classifying workgroups)
    // Everything else.  NEVER referenced by name: the compiler loads every kernel argument it sees used in the entry block, and with
    // ~150 argument dwords it then spills them to VGPR lanes right there -- five serialised fetch / wait / spill rounds before the
    // first branch, paid by every one of the 8 160 (1080p) .. 129 600 (8K) index slots.  The body reads them through the kernarg
    // segment pointer instead (cold_args() below), after the index slots have left.
    const ColdArgs cold_never_named)
{
    // `wave` through readfirstlane: the compiler cannot see that threadIdx.x >> 6 is the same in every lane of a wave, and would otherwise
    // treat every loop and branch that depends on it (the light loop of phase B, "wave < n_chunks", ...) as divergent
    const uint32_t tid = threadIdx.x, wave = (uint32_t) __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6)), lane = tid & 63;
    // ---- roles: [classify][list slots][paint + index slots]; `slot` counts the tracing workgroups (list slots, then index order).
    // Decided from the preloaded arguments alone.
    constexpr bool ALL_SPHERES_POSSIBLE = !HAS_GQ && !HAS_CUBIC; // fa.all_cullable needs a scene of spheres only
    const uint32_t ord_slots = (hot_flags & 2u) ? hot_ord_cap : 0u;
    uint32_t slot = blockIdx.x;
    uint32_t tstate = ST_TIMEOUT; // verdict on this workgroup's tile; TIMEOUT = decide here (no tile words, list slots, gave up polling)
    uint32_t role = 0, paint_block = 0; // 0 trace a tile, 1 classify, 2 paint
    if (ALL_SPHERES_POSSIBLE && hot_tile_state) { // launch-uniform
        if (blockIdx.x < hot_n_scan) { // workgroup-uniform
            role = 1;
        } else {
            slot -= hot_n_scan;
            if (slot >= ord_slots) {
                // behind the list slots: one paint workgroup in front of every RT_PAINT_TILES index slots (its own sixteen tiles' slots),
                // so that the painting -- HBM-bound -- runs beside the index slots, which are bound by the dispatch rate
                const uint32_t q = slot - ord_slots;
                if (hot_flags & 4u) {
                    const uint32_t grp = q / (RT_PAINT_TILES + 1u), pos = q - grp * (RT_PAINT_TILES + 1u);
                    if (pos == 0u) { // workgroup-uniform
                        role = 2;
                        paint_block = grp;
                    }
                    slot = ord_slots + grp * RT_PAINT_TILES + (pos - 1u);
                    if (role == 0 && slot - ord_slots >= hot_n_tiles) return; // the last group may be partial
                }
                if (role == 0) {
                    // index slot of tile slot - ord_slots: its word decides, and every wave of the workgroup reads the same decided value
                    uint32_t *w = hot_tile_state + (slot - ord_slots);
                    uint32_t v = 0;
                    if (lane == 0) v = tile_word_wait(w, hot_frame_tag, tile_word_load(w));
                    tstate = (uint32_t) __builtin_amdgcn_readfirstlane((int) v) & ST_MASK;
                    // EMPTY: a paint workgroup paints it and nothing else is to do.  (Counting builds count its rays further down;
                    // slot 0 may have frame duties when the launch-order lists are off.)
                    if (!COUNT && tstate == ST_EMPTY && slot != 0u) return;
                    if (tstate == ST_COVERED) return; // a list slot renders (and counts) it; slot 0 is a list slot whenever the lists are in use
                }
            }
        }
    }
    // ---- everything below needs the rest of the arguments ----
    const ColdArgs &cold = cold_args();
    const FrameArgs &fa = cold.fa;
    const unsigned char *__restrict__ gscene = cold.gscene;
    const DevLight *__restrict__ glight = cold.glight;
    void *__restrict__ fb = cold.fb;
    unsigned long long *__restrict__ counters = cold.counters;
    const double *__restrict__ camx = cold.camx, *__restrict__ camy = cold.camy;
    ConstLights clight = (ConstLights) (glight + fa.n_lights); // the LightK table behind the DevLight array (rt_create)
    constexpr bool OWNG = !HAS_GQ && !HAS_CUBIC; // instantiations whose scenes may consist of unit spheres only: the own-sphere rule applies there
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr bool NEED_CROSS = HAS_GQ || HAS_CUBIC;
    static_assert(!LEAN || (!HAS_GQ && !HAS_CUBIC && !HAS_MIRROR), "the lean path renders unit spheres only");
    const LdsLayout L(fa.stage_bytes, fa.n_lights, HAS_MIRROR, fa.cull ? fa.n_us : 0u, LEAN, HAS_CUBIC ? fa.n_cub : 0u);
    const DevObject *gobj = reinterpret_cast<const DevObject *>(gscene); // full object records: global memory only
    SceneLds S; // class tables + materials staged in LDS (only tiles with hits ever stage them); LDS offset = blob offset - off_us
    S.us = reinterpret_cast<const UsEntry *>(smem + L.scene);
    S.gq = reinterpret_cast<const GqEntry *>(smem + L.scene + (fa.off_gq - fa.off_us));
    S.lin = reinterpret_cast<const LinEntry *>(smem + L.scene + (fa.off_lin - fa.off_us));
    S.cub = reinterpret_cast<const uint32_t *>(smem + L.scene + (fa.off_cub - fa.off_us));
    S.mat = reinterpret_cast<const MatEntry *>(smem + L.scene + (fa.off_mat - fa.off_us));
    S.light = reinterpret_cast<const DevLight *>(smem + L.light);
    S.cubrec = lds_addr(smem + L.cubrec); S.cubtmp = lds_addr(smem + L.cubtmp); S.cubprim = lds_addr(smem + L.cubprim);
    S.cubtmp_p = reinterpret_cast<double *>(smem + L.cubtmp);
    SceneLds G; // the same tables where they live in global memory (round 0 runs straight from there)
    G.mat = reinterpret_cast<const MatEntry *>(gscene + fa.off_mat);
    G.us = reinterpret_cast<const UsEntry *>(gscene + fa.off_us);
    G.gq = reinterpret_cast<const GqEntry *>(gscene + fa.off_gq);
    G.lin = reinterpret_cast<const LinEntry *>(gscene + fa.off_lin);
    G.cub = reinterpret_cast<const uint32_t *>(gscene + fa.off_cub);
    G.light = glight;
    G.cubrec = S.cubrec; G.cubtmp = S.cubtmp; G.cubprim = S.cubprim; G.cubtmp_p = S.cubtmp_p; // (LDS either way)
    double *hp = reinterpret_cast<double *>(smem + L.hp);   // [3][WG] hit points (SoA: lane-consecutive, conflict-free)
    double *hn = reinterpret_cast<double *>(smem + L.hn);   // [3][WG] hit normals
    double *hdir = reinterpret_cast<double *>(smem + L.hdir); // [3][WG] ray direction of the pixel, parked across phases B / C (mirrors only)
    uint32_t *hidx = reinterpret_cast<uint32_t *>(smem + L.hidx); // (object of hit h << 8) | its owner lane (pixel)
    float *scolor = reinterpret_cast<float *>(smem + L.color);    // [3][WG] direct lighting of the pixel's hit this round
    float *spark = reinterpret_cast<float *>(smem + L.park);      // [5][WG] mirrors: running colour, blend ratio, depth of the pixel
    unsigned long long *sblk = reinterpret_cast<unsigned long long *>(smem + L.shadow); // [4 chunks][n_lights] bit i: hit i of the chunk skips the light when shading (blocked, or the light is behind it)
    Ball *sball = reinterpret_cast<Ball *>(smem + L.ball);
    BoxH *sbox = reinterpret_cast<BoxH *>(smem + L.box);
    CullRec *screc = reinterpret_cast<CullRec *>(smem + L.crec); // [4 chunks][L.n_crec]
    uint32_t *s_wcount = reinterpret_cast<uint32_t *>(smem + L.misc); // [4] hits found by each wave this round
    uint32_t *s_live = s_wcount + 4;                                  // [4] per-wave "still bouncing" flags
    uint32_t *s_sparse = s_wcount + 8;                                // [1] sparse output: this tile's slot in the message (or none)
    uint32_t *s_zero = s_wcount + 10;                                 // [1] always 0 (see the final store)
    uint32_t *s_decode = s_wcount + 12;                               // [3] what wave 0 decoded from the launch order: tile, flags, cost scale
    uint32_t *s_half = s_wcount + 11;                                 // [1] which half of the tile this workgroup renders (0: all of it)
    uint32_t *s_cost = s_wcount + 9;                                  // [1] round 0: work of the shadow phase, for the next frame's launch order
    uint32_t *s_done = s_wcount + 15;                                 // [1] lean path: waves of this workgroup that have finished their block

    RT_STAMP_DECL
    const uint32_t stamp_row_base_ = 0u;
    (void) stamp_row_base_;
    // No prologue: 83 % of the tiles of a typical frame contain no hit at all, and for those the whole job is
    // "primary rays miss, store the background".  Round 0 therefore reads the (culled, tiny) part of the tables it
    // needs straight from global memory -- the blob is a few KB and lives in L2 / the scalar cache -- and the scene is
    // only staged into LDS once the tile is known to have hits (`staged`).
    bool staged = false;
    RT_STAMP(0);

    Cnt<COUNT> cnt;
    const F3 bg{fa.bg[0], fa.bg[1], fa.bg[2]};

    if (ALL_SPHERES_POSSIBLE && role == 1) { // workgroup-uniform
        classify_tiles<COUNT>(fa, reinterpret_cast<const UsEntry *>(hot_us), hot_n_us, hot_tile_state, hot_frame_tag, hot_n_tiles,
                              (hot_flags & 2u) ? hot_ord_rd : nullptr, hot_ord_cap, ((hot_flags >> 4) & 15u) | ((hot_flags >> 20) << 4), wave, lane, cnt);
        RT_STAMP(11);
        RT_STAMP_FLUSH(counters, lane);
        cnt.flush(counters);
        return;
    }
    if (ALL_SPHERES_POSSIBLE && role == 2) { // workgroup-uniform
        paint_tiles(fa, hot_tile_state, hot_frame_tag, hot_n_tiles, fb, bg, paint_block, wave, lane);
        RT_STAMP(11);
        RT_STAMP_FLUSH(counters, lane);
        return;
    }
#ifdef RT_WF_DEBUG_EXITS
    if ((hot_flags & 0x100u) && slot >= ord_slots) return; // timing experiment: index slots leave at once
#endif

    // ---- launch order from the previous frame ----
    // The frame time is set by when the LAST expensive tile starts (a tile full of hits costs ~20x an empty one and
    // they cluster), so the tiles that had hits in the previous frame are started first, the ones with the most hits
    // before the others.  The grid has ord_cap + n_tiles slots: slot b < n_eff renders entry b of the previous frame's
    // lists (four cost classes, heaviest first; n_eff = the listed tiles that fit into ord_cap slots), the last
    // n_tiles slots are the tiles in index order, and the ones a list slot covers exit at once.  Every tile is
    // rendered exactly once whatever the lists and ord_cap say, so they change the time, never the image.
    //
    // This pays most when few tiles have hits and the launch is only a few workgroups per slot deep (config 2: 15 % of 8160
    // tiles, -7 %); otherwise it costs a dependent load at the start of every workgroup and one same-address device
    // atomic per tile with hits (those sustain ~90 per microsecond): +5 % on a frame where every tile has hits.  So
    // one tile in 16 takes part in a census, the host sees it through a mapped word and switches the ordering off
    // (ord_on = 0: index order, nothing read or appended, census only) while >= 25 % of the tiles have hits.
    // tile-level early-out (below): lane j of wave 0 tests sphere j, whatever the tile -- so the entry is requested
    // before the tile is even known and its latency overlaps with the order-state reads
    double pkx = 0.0, pky = 0.0, pkz = 0.0, pr = 0.0, pinv = 0.0;
    if (ALL_SPHERES_POSSIBLE && (hot_flags & 1u) && wave == 0 && lane < hot_n_us) {
        const UsEntry *pe = reinterpret_cast<const UsEntry *>(hot_us) + lane;
        pkx = pe->kx; pky = pe->ky; pkz = pe->kz; pr = pe->r; pinv = pe->inv_r;
    }
    uint32_t *ord_wr = nullptr;
    uint32_t tile = slot;
    bool listed = false;  // this tile had hits in the previous frame
    uint32_t half = 0;    // list slots of the costliest entries: 1 = rows 0-7 of the tile only, 2 = rows 8-15 only (0: the whole tile)
    bool listing = false; // this frame appends to the lists
    uint32_t cost_scale = 0; // largest tile cost of the previous frame: the scale of the cost classes
    bool covered = false; // index-order slot whose tile a list slot renders: leaves before its first side effect
    if (fa.order_state && !fa.ord_on) { // launch-uniform: census only
        ord_wr = fa.order_state + (size_t) fa.ord_write * fa.ord_stride;
        if (slot == 0 && tid == 0) {
            const uint32_t *ord_rd = fa.order_state + (size_t) fa.ord_read * fa.ord_stride;
            uint32_t *z = fa.order_state + (size_t) fa.ord_zero * fa.ord_stride;
            for (uint32_t k = 0; k < RT_ORD_HDR; k += 4) *reinterpret_cast<uint4 *>(z + k) = make_uint4(0, 0, 0, 0);
            if (fa.ord_host) { fa.ord_host[0] = 0; fa.ord_host[1] = ord_rd[16]; fa.ord_host[2] = ord_rd[16] * 16u; } // [2]: tiles with hits (here: the census' estimate)
        }
    }
    if (hot_flags & 2u) { // launch-uniform
        // ONE wave decodes the launch order for the workgroup and hands the result to the other three through LDS.  The decode is a
        // few hundred scalar instructions (sixteen class counts: sums, a search, a rank), a CU has ONE scalar unit, and at the start of a
        // frame all 24 waves of a CU are here at once: with every wave decoding for itself the set-up phase was bound by that unit
        // (~13 000 scalar issue cycles per CU, the 5 us every tracing workgroup spent before its first ray; measured: 70 more scalar
        // instructions per wave cost 1 700 cycles per wave).
        ord_wr = fa.order_state + (size_t) fa.ord_write * fa.ord_stride;
        // ask for the camera / frame part of the kernel arguments here (all waves need them), before anybody waits for anything
        asm volatile("" ::"s"(fa.cam[0]), "s"(fa.cam[2]), "s"(fa.cam[5]), "s"(fa.cam[6]), "s"(fa.cam[8]), "s"(fa.cam[10]), "s"(fa.origin[0]),
                     "s"(fa.origin[2]), "s"(fa.aspect), "s"(fa.tan_half_fov), "s"(fa.width), "s"(fa.local_rows), "s"(fa.tiles_x), "s"(fa.n_us),
                     "s"(fa.off_us), "s"(fa.band_rows), "s"(fa.all_cullable), "s"(gscene));
        if (wave == 0) { // wave-uniform
            // the generation being read was written by the previous launch and is not touched by this one: constant address
            // space, so that the reads become scalar loads (through the generic pointer they are vector loads + readfirstlane)
            typedef const __attribute__((address_space(4))) uint32_t *ConstWords;
            ConstWords ord_rd = (ConstWords) hot_ord_rd;
            const OrdHeader oh = ord_header(ord_rd, ((hot_flags >> 4) & 15u) | ((hot_flags >> 20) << 4)); // listed tiles per cost class, census, largest cost, half tiles
            // the per-tile word of an index-order slot is requested with them (for a list slot: tile 0's, unused)
            const uint32_t idx_tile = slot >= ord_slots ? slot - ord_slots : 0u;
            const uint32_t w = ord_rd[RT_ORD_HDR + idx_tile]; // (position in its class list << 5) | (class + 1), 0 = had no hits
            const uint32_t n_eff = ord_positions(oh) < hot_ord_cap ? ord_positions(oh) : hot_ord_cap; // positions in use (half tiles count twice)
            if (slot == 0 && lane == 0) {
                uint32_t *z = fa.order_state + (size_t) fa.ord_zero * fa.ord_stride; // the generation the NEXT frame appends to
                for (uint32_t k = 0; k < RT_ORD_HDR; k += 4) *reinterpret_cast<uint4 *>(z + k) = make_uint4(0, 0, 0, 0);
                if (fa.ord_host) { fa.ord_host[0] = oh.n_listed + oh.n_candidates; fa.ord_host[1] = oh.census; fa.ord_host[2] = oh.n_listed; } // host-mapped: sizes / switches later launches
            }
            uint32_t d_half = 0;
            uint32_t d_tile = slot, d_flags = oh.census * 64u < hot_n_tiles ? 4u : 0u; // 1 listed, 2 covered, 4 listing (the host's switch lags a few frames: same rule here), 8 leave
            if (slot < ord_slots) {
                if (slot >= n_eff) {
                    d_flags |= 8u;
                } else {
                    const uint32_t p = (hot_flags & 8u) ? slot : ord_rank_of_slot(slot, n_eff, blockIdx.x - slot); // 8: A/B switch, positions in list order
                    const uint32_t rank = p < 2u * oh.n_heavy ? p >> 1 : p - oh.n_heavy;
                    d_half = p < 2u * oh.n_heavy ? 1u + (p & 1u) : 0u;
                    uint32_t k, idx;
                    ord_locate(oh, rank, k, idx);
                    d_tile = ord_rd[RT_ORD_HDR + (1u + k) * hot_n_tiles + idx];
                    d_flags |= (d_tile >= hot_n_tiles || ord_last_position(oh, rank) >= n_eff) ? 8u : 1u; // (>= n_tiles: never true for lists this kernel wrote;
                                                             // a pair cut in two by the end of the slots: the tile's index slot renders all of it)
                }
            } else {
                d_tile = idx_tile;
                // A tile without hits writes nothing, so its word may be left over from an older frame of this generation:
                // the word is only a hint where to look, and the tile counts as covered iff that list entry really names it.
                const uint32_t cls = w & 31u, pos = w >> 5;
                // (NONEMPTY: its classifier already found it uncovered.  EMPTY: the classifier does not look -- a list slot may still trace a tile
                // that had hits in the previous frame; it stores the same background a paint workgroup does, but a counting build must
                // not book the tile's rays twice, so there the index slot looks for itself.)
                if ((tstate == ST_TIMEOUT || (COUNT && tstate == ST_EMPTY)) && cls >= 1u && cls <= ORD_CLASSES && pos < hot_n_tiles) {
                    const uint32_t k = cls - 1u;
                    uint32_t count = 0;
#pragma unroll
                    for (uint32_t c = 0; c < ORD_CLASSES; c++) count = c == k ? oh.cnt[c] : count;
                    if (pos < count && ord_last_position(oh, ord_first(oh, k) + pos) < n_eff && ord_rd[RT_ORD_HDR + (1u + k) * hot_n_tiles + pos] == d_tile) d_flags |= 2u;
                }
            }
            if (lane == 0) { s_decode[0] = d_tile; s_decode[1] = d_flags; s_decode[2] = oh.cost_max; s_half[0] = d_half; }
        }
        if (LEAN) RT_STAMP(3); // (lean path, stamped builds: set-up split into "until the decode barrier", ...
        lds_barrier();
        if (LEAN) RT_STAMP(4); // ... "waiting at it", ...
        tile = (uint32_t) __builtin_amdgcn_readfirstlane((int) s_decode[0]);
        const uint32_t d_flags = (uint32_t) __builtin_amdgcn_readfirstlane((int) s_decode[1]);
        cost_scale = (uint32_t) __builtin_amdgcn_readfirstlane((int) s_decode[2]);
        if (d_flags & 8u) return; // workgroup-uniform: a list slot beyond the listed tiles
        listed = (d_flags & 1u) != 0u;
        covered = (d_flags & 2u) != 0u;
        listing = (d_flags & 4u) != 0u;
        half = (uint32_t) __builtin_amdgcn_readfirstlane((int) s_half[0]);
    }
#ifdef RT_WF_DEBUG_EXITS
    if ((hot_flags & 0x200u) && !listed) { // timing experiment: index slots leave after the order-state / tile-state loads have come back
        asm volatile("" ::"s"(tile), "s"((uint32_t) covered), "s"(tstate));
        return;
    }
#endif
    {
        // (the frame's sizes, asked for together: the compiler fetches kernel arguments again rather than keep them in SGPRs, and left to
        // itself it does so one branch at a time -- each a scalar round trip on every tracing workgroup's way to its first ray)
        asm volatile("" ::"s"(fa.width), "s"(fa.local_rows), "s"(fa.tiles_x), "s"(fa.band_rows), "s"(fa.rank), "s"(fa.world), "s"(fa.all_cullable),
                     "s"(fa.tile_planes_ok));
        // pixel of this lane: each wave covers an 8 x 8 quadrant of the tile (tile_px / tile_py)
        const uint32_t tile_x = tile % fa.tiles_x, tile_y = tile / fa.tiles_x;
        const uint32_t x = tile_x * RT_TILE + tile_px(tid), lr = tile_y * RT_TILE + tile_py(tid);
        const bool inside = x < fa.width && lr < fa.local_rows && (half == 0u || (tid >> 7) + 1u == half); // (tile_py: threads 0-127 are rows 0-7)
        // lanes outside the image trace a clamped pixel (keeps the wave's primary cone tight); their result is dropped
        const uint32_t xc = x < fa.width ? x : fa.width - 1, lrc = lr < fa.local_rows ? lr : fa.local_rows - 1;
        const uint32_t y = global_row(fa, lrc);

        // ---- tile-level early-out ----
        // When every object is a cullable sphere, ONE wave first tests the pyramid of the whole 16x16 tile (five planes,
        // sphere_in_pyramid above: no division or square root on this path, which 83 % of config 2's workgroups take and
        // nothing else) against all spheres.  A tile that cannot be hit ends here: the other three waves never form a
        // ray, and the tile is just the background colour.
        if (ALL_SPHERES_POSSIBLE && tstate == ST_EMPTY) { // workgroup-uniform.  Only counting builds and slot 0 get here (see the role decode)
            if (COUNT && !covered && inside) {
                cnt.add(0);
                cnt.add(3, fa.n_obj);
            }
            RT_STAMP_FLUSH(counters, lane);
            cnt.flush(counters);
            return;
        }
        // A tile whose word says NONEMPTY goes straight to phase A; TIMEOUT (no tile words in this launch, or nobody classified
        // the tile in time) decides here.
        if (ALL_SPHERES_POSSIBLE && fa.all_cullable && fa.tile_planes_ok && !listed && tstate == ST_TIMEOUT) { // launch-uniform x workgroup-uniform; a listed tile almost surely has hits again
            if (wave == 0) {
                // this lane's sphere (first group of 64) was requested at the top of the kernel
                const uint32_t x0 = tile_x * RT_TILE, y0l = tile_y * RT_TILE;
                const uint32_t x1 = x0 + RT_TILE - 1 < fa.width ? x0 + RT_TILE - 1 : fa.width - 1;
                const uint32_t y1l = y0l + RT_TILE - 1 < fa.local_rows ? y0l + RT_TILE - 1 : fa.local_rows - 1;
                const double gy0 = (double) global_row(fa, y0l), gy1 = (double) global_row(fa, y1l); // increasing in the local row
                const TilePlanes P = tile_planes(fa, fa.cx_a * ((double) x0 - 0.5) + fa.cx_b, fa.cx_a * ((double) x1 + 0.5) + fa.cx_b,
                                                 fa.cy_a * (gy0 - 0.5) + fa.cy_b, fa.cy_a * (gy1 + 0.5) + fa.cy_b);
                const D3 org{fa.origin[0], fa.origin[1], fa.origin[2]};
                unsigned long long any = __ballot(lane < fa.n_us && sphere_in_pyramid(pkx, pky, pkz, pr, pinv, org, P));
                if (lane == 0) cnt.cull(C_TILE, fa.n_us < 64 ? fa.n_us : 64);
                for (uint32_t base = 64; base < fa.n_us; base += 64) {
                    const uint32_t end = (base + 64 < fa.n_us) ? base + 64 : fa.n_us;
                    bool rel = false;
                    if (base + lane < end) {
                        const UsEntry e = G.us[base + lane];
                        rel = sphere_in_pyramid(e.kx, e.ky, e.kz, e.r, e.inv_r, org, P);
                    }
                    any |= __ballot(rel);
                    if (lane == 0) cnt.cull(C_TILE, end - base);
                }
                if (lane == 0) s_live[0] = any != 0ull ? 1u : 0u;
            }
            if (covered) return; // workgroup-uniform
            lds_barrier();
#ifdef RT_WF_DEBUG_EXITS
            if ((hot_flags & 0x400u)) return; // timing experiment: leave after the barrier, no paint
#endif
            if (s_live[0] == 0) { // workgroup-uniform: the tile is pure background (src/update-cpu.cpp:93-95)
                if (inside && fa.sparse) { // sparse output: background tiles are not stored at all
                    cnt.add(0);
                    cnt.add(3, fa.n_obj);
                } else if (inside) {
                    cnt.add(0);
                    cnt.add(3, fa.n_obj);
                    const size_t pix = (size_t) lr * fa.width + x;
                    if (fa.rgba8) {
                        uchar4 px;
                        px.x = (unsigned char) (int) (bg.x * 255.0f + 0.5f);
                        px.y = (unsigned char) (int) (bg.y * 255.0f + 0.5f);
                        px.z = (unsigned char) (int) (bg.z * 255.0f + 0.5f);
                        px.w = 255;
                        reinterpret_cast<uchar4 *>(fb)[pix] = px;
                    } else {
                        reinterpret_cast<float4 *>(fb)[pix] = make_float4(bg.x, bg.y, bg.z, 1.0f);
                    }
                }
                RT_STAMP(1);
                RT_STAMP_FLUSH(counters, lane);
                cnt.flush(counters);
                return;
            }
            lds_barrier(); // s_live[0] is reused by the round loop
        }

        if (covered) return; // workgroup-uniform
        // A tile that gets here (its word says NONEMPTY, or it passed the tile-level test, or was listed, or the scene has no such
        // test) will almost surely find hits and stage the scene: request this thread's share of the copy now, in front of the
        // camera-table reads.  Two 16-byte pieces per thread cover 8 KB of tables + lights; larger scenes copy the rest the ordinary way.
        const uint4 *stage_scene = reinterpret_cast<const uint4 *>(gscene + fa.off_us);
        const uint4 *stage_light = reinterpret_cast<const uint4 *>(glight);
        const uint32_t n16 = fa.stage_bytes / 16, tot16 = n16; // (the class tables and materials; the lights are read through scalar loads)
        uint4 pre0 = make_uint4(0, 0, 0, 0), pre1 = make_uint4(0, 0, 0, 0);
        if (tid < tot16) pre0 = tid < n16 ? stage_scene[tid] : stage_light[tid - n16];
        if (tid + WG < tot16) pre1 = tid + WG < n16 ? stage_scene[tid + WG] : stage_light[tid + WG - n16];
        F3 res = bg;
        D3 o{fa.origin[0], fa.origin[1], fa.origin[2]};
#if RT_WF_CAMTAB
        D3 dir = primary_dir_tab(fa, camx[xc], camy[y]);
#else
        D3 dir = primary_dir(fa, (int) xc, (int) y);
#endif
        // All-sphere scenes: only tiles that some sphere reaches into get here (tile words / tile-level test / lists), so the
        // tables go to LDS BEFORE phase A and round 0 reads them there.  (Reading them from global memory in round 0 -- scalar
        // loads for the wave-uniform entries -- was the cheap way while most workgroups that got here found their tile empty;
        // with ~4800 waves starting phase A in the same microsecond it serialised them on the scalar cache: phase A took 5-10 us.)
        if (ALL_SPHERES_POSSIBLE && fa.all_cullable) { // launch-uniform
            uint4 *dst = reinterpret_cast<uint4 *>(smem + L.scene);
            if (tid < tot16) dst[tid] = pre0;
            if (tid + WG < tot16) dst[tid + WG] = pre1;
            for (uint32_t i = tid + 2 * WG; i < tot16; i += WG) dst[i] = i < n16 ? stage_scene[i] : stage_light[i - n16];
            staged = true;
            if (LEAN && tid == 0) { s_done[0] = 0u; s_cost[0] = 0u; s_zero[0] = 0u; if (!(hot_flags & 2u)) s_half[0] = 0u; }
            if (LEAN) RT_STAMP(7); // ... "tile known -> staging loads back and written", ...
            lds_barrier();
            if (LEAN) RT_STAMP(8); // ... "staging barrier"; phase 1 is then the camera-table reads and the primary direction)
        }
        bool live = inside; // this pixel still has a ray to trace
        uint32_t ord_cls = 0, ord_pos = 0; // one thread per tile with hits: the tile's entry in the next frame's launch order
        if constexpr (LEAN) {
            // (lean_block, defined at the top of the kernel: this wave's own 8 x 8 quadrant of the tile)
            unsigned long long hm = 0ull, b_ticks = 0ull;
            LeanLds M;
            M.hp = hp; M.hp_stride = WG; M.sball = sball; M.sbox = sbox; M.screc = screc; M.smem = smem; M.crec_off = L.crec; M.n_crec = L.n_crec;
            lean_block<COUNT>(fa, S, gobj, M, glight, cnt, tile, o, dir, live, listing, lane, tid, wave, res, hm, b_ticks RT_STAMP_PASS);
            // ---- the next frame's launch order: the last wave of the workgroup to get here speaks for the tile ----
            // cost = shader-clock ticks / 64 its waves spent on their lights, summed (what the tile takes of its CU)
            if (ord_wr && lane == 0) { // launch-uniform x one lane per wave
                if (hm != 0ull) __hip_atomic_fetch_add(&s_cost[0], (uint32_t) (b_ticks >> 6) + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const uint32_t before = __hip_atomic_fetch_add(&s_done[0], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (before == 3u) {
                    const uint32_t cost4 = __hip_atomic_load(&s_cost[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    const uint32_t cost = (cost4 + 3u) >> 2; // per wave: the scale the other schedule reports on (its slowest wave's time), so that a switch does not leave the scale a factor of four off
                    if (cost != 0u) { // the tile has hits
                        if (((tile * 0x9E3779B1u) >> 28) == 0u) atomicAdd(&ord_wr[16], 1u); // census, also while the ordering is off
                        if (listing) {
                            const uint32_t q = cost_scale ? (uint32_t) (((unsigned long long) cost * ORD_CLASSES) / ((unsigned long long) cost_scale + 1ull)) : 0u; // (no scale yet: the cheapest class -- nothing is split on a guess)
                            ord_cls = ORD_CLASSES - (q < ORD_CLASSES - 1u ? q : ORD_CLASSES - 1u); // class + 1
                            ord_pos = atomicAdd(&ord_wr[ord_cls - 1u], 1u);
                            if (cost > cost_scale - cost_scale / 4u || ((tile * 0x9E3779B1u) >> 28) == 0u) atomicMax(&ord_wr[17], cost); // (see the other schedule's report)
                        }
                    }
                }
            }
        } else {
        bool first = true;
        if (HAS_CUBIC && fa.n_cub != 0u) {
            // degree-3 objects: the records of the frame's ray origin (FrameArgs::cub_rec) into LDS, where cubic_test reads them.  Every wave
            // writes all of them -- the same values to the same words -- and reads them back only after its own stores (LDS operations of a
            // wave complete in order): no workgroup barrier in front of round 0.
            double *prim = reinterpret_cast<double *>(smem + L.cubprim);
            if (lane < (RT_CUB_REC + 4) * RT_CUB_AT_MAX) prim[lane] = (&fa.cub_rec[0][0])[lane]; // (cub_rec[4][10], then cub_abs[4][4]: contiguous in FrameArgs)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
        }
        if (fa.sparse && tid == 0) s_sparse[0] = 0xFFFFFFFFu; // visible after the first barrier of round 0
        if (!(hot_flags & 2u) && tid == 0) s_half[0] = 0u;    // (with the lists on, wave 0 wrote it with the decode)
        if (tid == 0) s_zero[0] = 0u;
        float cur_ratio = 1.0f;
        uint32_t n_refl = 0;
        if (inside) cnt.add(0);
        cnt.primary_traced(); // every lane of a tracing workgroup forms a primary ray (lanes outside the image a clamped one)

        RT_STAMP(1);
        for (;;) { // ----------------------- rounds -----------------------
            // ---------------- phase A: nearest hit ----------------
            double best_t;
            int best;
            {
                Mono m;
                mono_set_o<NEED_CROSS>(m, o);
                mono_set_d<NEED_CROSS>(m, dir);
                mono_set_od<NEED_CROSS>(m);
                if (first && staged) nearest<COUNT, HAS_GQ, HAS_CUBIC, true>(fa, S, gobj, m, live, lane, best_t, best, cnt);
                else if (first) nearest<COUNT, HAS_GQ, HAS_CUBIC, true>(fa, G, gobj, m, live, lane, best_t, best, cnt);
                else nearest<COUNT, HAS_GQ, HAS_CUBIC, false>(fa, S, gobj, m, live, lane, best_t, best, cnt);
            }
            RT_STAMP(2);
            if (live) cnt.add(3, fa.n_obj);
            const bool hit = live && best >= 0;
            if (live && !hit) { // the ray leaves the scene (src/update-cpu.cpp:93-95,112-115)
                if (!first) blend(res, cur_ratio, bg);
                live = false;
            }
            D3 sp{0.0, 0.0, 0.0}, sn{0.0, 0.0, 1.0};
            uint32_t own_ok = 0u; // bit 31 of the hit's queue word (below)
            if (hit) {
                sp = D3{o.x + best_t * dir.x, o.y + best_t * dir.y, o.z + best_t * dir.z};
                if (ALL_SPHERES_POSSIBLE && fa.n_us == fa.n_obj) { // launch-uniform: every object is a unit sphere -- table index == object index
                    const UsEntry eo = staged ? S.us[best] : G.us[best]; // three coefficients from the table instead of twenty from the object record
                    sn = sphere_normal(eo, sp);
                    Mono mo;
                    mono_set_o<false>(mo, D3{sp.x + SHADOW_BIAS * sn.x, sp.y + SHADOW_BIAS * sn.y, sp.z + SHADOW_BIAS * sn.z});
                    const double t0_own = us_t0(eo, mo); // the reference's own t0 of this hit's shadow rays against the sphere it lies on
                    own_ok = (t0_own > (double) eo.own_lo && t0_own < (double) eo.own_hi) ? 0x80000000u : 0u;
                } else {
                    sn = normal_vector(gobj[best].c, sp); // 20 coefficients of the hit object, gathered from global (L2) per hit
                }
                cnt.add(4);
            }
            // ---------------- compaction of the tile's hits into the LDS queue ----------------
            const unsigned long long hm = __ballot(hit);
            if (lane == 0) s_wcount[wave] = (uint32_t) __popcll(hm);
            RT_STAMP(3);
            lds_barrier();
            RT_STAMP(4);
            // (FMA build: through readfirstlane -- values read from LDS are not known to be uniform and sit in VGPRs through all phases; as
            // scalars they make room for six workgroups per CU there: 1080p 46.4 -> 43.6 us, 8K 462 -> 413.  The strict build fits
            // without, and is 1 % faster without.)
#if RT_FAST
#define RT_WAVE_COUNT(i) ((uint32_t) __builtin_amdgcn_readfirstlane((int) s_wcount[i]))
#else
#define RT_WAVE_COUNT(i) (s_wcount[i])
#endif
            const uint32_t c0 = RT_WAVE_COUNT(0), c1 = RT_WAVE_COUNT(1), c2 = RT_WAVE_COUNT(2), c3 = RT_WAVE_COUNT(3);
#undef RT_WAVE_COUNT
            const uint32_t n_hits = c0 + c1 + c2 + c3;
            if (first && fa.sparse && tid == 0 && n_hits) // sparse output: a tile with hits takes a slot of the message; the slot
                s_sparse[0] = atomicAdd(reinterpret_cast<uint32_t *>(fb), 1u); // parked in LDS at once (not in a register across the shadow phase); the barriers of this round publish it before the final store
            if (first && ord_wr && tid == 0 && n_hits && s_half[0] != 2u) { // the next frame's launch order (census: a split tile counts through its upper half)
                if (((tile * 0x9E3779B1u) >> 28) == 0u) atomicAdd(&ord_wr[16], 1u); // census, also while the ordering is off
            }
            if (first) RT_STAMP_INFO(((unsigned long long) tile << 32) | n_hits);
            if (n_hits == 0) break; // workgroup-uniform: nothing left to shade or bounce (all lanes are dead by now)
            const uint32_t n_chunks = (n_hits + 63) >> 6;
            if (!staged) { // first round with hits: bring the scene and the lights into LDS (one contiguous image), 16 B per lane per step
                uint4 *dst = reinterpret_cast<uint4 *>(smem + L.scene);
                if (tid < tot16) dst[tid] = pre0; // the first two pieces are already in registers
                if (tid + WG < tot16) dst[tid + WG] = pre1;
                for (uint32_t i = tid + 2 * WG; i < tot16; i += WG) dst[i] = i < n16 ? stage_scene[i] : stage_light[i - n16];
                staged = true;
            }
            uint32_t my_slot = 0;
            if (hit) {
                my_slot = (wave > 0 ? c0 : 0u) + (wave > 1 ? c1 : 0u) + (wave > 2 ? c2 : 0u) + (uint32_t) __popcll(hm & ((1ull << lane) - 1ull));
                hp[my_slot] = sp.x; hp[WG + my_slot] = sp.y; hp[2 * WG + my_slot] = sp.z;
                hn[my_slot] = sn.x; hn[WG + my_slot] = sn.y; hn[2 * WG + my_slot] = sn.z;
                // object of hit h, its owner lane (pixel) in the low byte; bit 31: scenes of unit spheres only -- the reference's own t0 of this
                // hit's shadow rays against the sphere it lies on is inside that sphere's window (own_sphere_skippable in the lean path has
                // the argument: such a ray towards a directional light in front of the surface is not tested against that sphere)
                hidx[my_slot] = own_ok | ((uint32_t) best << 8) | tid;
                if (HAS_MIRROR) { // the bounce needs the incoming direction again in phase D; keep it out of registers meanwhile
                    hdir[tid] = dir.x; hdir[WG + tid] = dir.y; hdir[2 * WG + tid] = dir.z;
                }
            }
            if (HAS_MIRROR) { // so does the blend: running colour, ratio and depth wait in LDS while phases B and C run
                spark[tid] = res.x; spark[WG + tid] = res.y; spark[2 * WG + tid] = res.z;
                spark[3 * WG + tid] = cur_ratio;
                reinterpret_cast<uint32_t *>(spark)[4 * WG + tid] = n_refl;
            }
            if (first && tid == 0) s_cost[0] = 0;
            lds_barrier();

            // ---------------- phase A': chunk bounding balls ----------------
            if (wave < n_chunks) { // wave-uniform
                const bool v = tid < n_hits;
                const double px = hp[v ? tid : wave * 64u], py = hp[WG + (v ? tid : wave * 64u)], pz = hp[2 * WG + (v ? tid : wave * 64u)];
                if (fa.n_us != 0u) { // (launch-uniform: box, ball and records only serve the culling of spheres)
                // Bounding box of the chunk's hit points, reduced in FP32 with outward rounding (a box that is one float ulp larger
                // costs nothing -- it only feeds the conservative culling -- and a 32-bit DPP min / max step is one instruction
                // where the FP64 one is five); centre and radius of its ball in FP64.
                const float lox = wave_min_f(__double2float_rd(px)), hix = wave_max_f(__double2float_ru(px));
                const float loy = wave_min_f(__double2float_rd(py)), hiy = wave_max_f(__double2float_ru(py));
                const float loz = wave_min_f(__double2float_rd(pz)), hiz = wave_max_f(__double2float_ru(pz));
                const double dx = (double) hix - (double) lox, dy = (double) hiy - (double) loy, dz = (double) hiz - (double) loz;
                Ball b;
                b.cx = 0.5 * ((double) lox + (double) hix);
                b.cy = 0.5 * ((double) loy + (double) hiy);
                b.cz = 0.5 * ((double) loz + (double) hiz);
                // half diagonal of the box (rounded up) + the 1e-2 shadow bias of the ray origins
                b.R = 0.5 * sqrt(dx * dx + dy * dy + dz * dz) * (1.0 + 1e-9) + 1.01e-2;
                if (lane == 0) {
                    sball[wave] = b;
                    sbox[wave] = BoxH{0.5 * dx * (1.0 + 1e-9) + 1.01e-2, 0.5 * dy * (1.0 + 1e-9) + 1.01e-2, 0.5 * dz * (1.0 + 1e-9) + 1.01e-2, 0.0};
                }
                // the light-independent half of the shadow-phase culling, lane = sphere (first group of 64)
                if (lane < L.n_crec) screc[wave * L.n_crec + lane] = cull_record(S.us[lane], b);
                if (lane == 0) cnt.cull(C_RECORDS, L.n_crec);
                }
                if (HAS_CUBIC && fa.n_cub != 0u) {
                    // degree-3 scenes: the first such object's Taylor data (and error bounds) at this hit's shadow-ray origin -- the same for every
                    // light -- into the hit's record; the origin is formed exactly as phase B forms it
                    const uint32_t hs = v ? tid : wave * 64u;
                    const D3 so{px + SHADOW_BIAS * hn[hs], py + SHADOW_BIAS * hn[WG + hs], pz + SHADOW_BIAS * hn[2 * WG + hs]};
                    const double *c0 = gobj[__builtin_amdgcn_readfirstlane(S.cub[0])].c;
                    if (v) {
                        cubic_rec_store(reinterpret_cast<double *>(smem + L.cubrec) + tid, WG, cubic_at(c0, so));
                        cnt.cubic_point();
                    }
                }
            }
            lds_barrier();
            RT_STAMP(5);

            // ---------------- phase B: shadow rays ----------------
            // Every wave visits every chunk and takes the lights l == (wave - chunk) mod 4 of it, so the per-chunk
            // part of the ray (origin, its monomials) is formed once per wave and the lights of a tile with few
            // hits are still spread over all four waves.
            // The tile's cost for the next frame's launch order: how long its slowest wave spends in the shadow phase (shader clock / 64).
            // Crude -- it depends a little on what else ran on the CU -- but free: two s_memtime per wave instead of book-keeping per item.
            const unsigned long long b_t0 = (first && listing) ? __builtin_amdgcn_s_memtime() : 0ull;
            for (uint32_t c = 0; c < n_chunks; c++) { // wave-uniform
                const uint32_t h = c * 64 + lane;
                const bool valid = h < n_hits;
                const uint32_t hs = valid ? h : c * 64;
                Mono sm;
                const D3 nrm{hn[hs], hn[WG + hs], hn[2 * WG + hs]};
                {
                    const D3 p{hp[hs], hp[WG + hs], hp[2 * WG + hs]};
                    mono_set_o<NEED_CROSS>(sm, D3{p.x + SHADOW_BIAS * nrm.x, p.y + SHADOW_BIAS * nrm.y, p.z + SHADOW_BIAS * nrm.z});
                }
                // this hit's own sphere and its window bit (scenes of unit spheres only; see the queue)
                const uint32_t hinfo_b = hidx[hs];
                const uint32_t own = (hinfo_b >> 8) & 0x7FFFFFu;
                const bool own_ok = OWNG && valid && (hinfo_b >> 31) != 0u && fa.n_us == fa.n_obj;
                const CullRec *my_crec = reinterpret_cast<const CullRec *>(smem + (L.crec + c * L.n_crec * (uint32_t) sizeof(CullRec)));
                const uint32_t cub_rec0 = HAS_CUBIC ? S.cubrec + (hs << 3) : 0u; // degree-3 scenes: this hit's record (phase A')
                for (uint32_t l = (wave + 4u - (c & 3u)) & 3u; l < fa.n_lights; l += 4) {
                    // the light through the constant address space (LightK, rt_scene_dev.h): scalar loads, its fields in SGPRs; one body per
                    // light kind, nothing merged between them
                    RT_LOAD_LIGHTK(lk, l)
                    if (valid) cnt.add(1);
                    unsigned long long skip; // hits of the chunk that skip this light when shading
                    if (!(lk.flags & 1u)) { // ---- directional: include/light_impl.h:23-25 ----
                        // A directional light behind the surface contributes exactly +0 whether or not it is shadowed:
                        // surface_color multiplies by max(0.0f, (float) dot(n, light.p)) (include/light_impl.h:43), and the
                        // colour it scales is finite (checked at rt_create: flag 2).  Such lanes sit the shadow test out AND the shading of this
                        // light.  (COUNT builds test them anyway: the reference-equivalent test count needs the index of the first blocker.)
                        const bool bfe = (lk.flags & 2u) != 0u;
                        const float lam = (float) dot3(nrm, D3{lk.p[0], lk.p[1], lk.p[2]});
                        const bool wanted = valid && (!bfe || 0.0f < lam); // the lanes the product build traces
                        skip = COUNT ? 0ull : __ballot(valid && !wanted);
                        if (COUNT || __any(wanted)) {
                            const bool quad_l = (lk.flags & 4u) != 0u;
                            Mono sd = sm;
                            if (NEED_CROSS) {
                                mono_set_d<true>(sd, D3{lk.sdir[0], lk.sdir[1], lk.sdir[2]});
                            } else {
                                sd.d = D3{lk.sdir[0], lk.sdir[1], lk.sdir[2]};
                                sd.u2 = lk.u2;
                            }
                            mono_set_od<NEED_CROSS>(sd);
                            if (wanted) cnt.traced();
                            const int blocker = shadow_blocker<COUNT, HAS_GQ, HAS_CUBIC, false, OWNG>(fa, S, gobj, sd, 1e6, COUNT ? valid : wanted, wanted, sball + c, sbox + c, my_crec, lk, lane, cnt,
                                                                                                      own, own_ok && bfe && quad_l && wanted, 0ull, cub_rec0);
                            // the reference stops at the first blocker in index order (src/update-cpu.cpp:66-71)
                            if (valid) cnt.add(3, blocker != NO_BLOCKER ? (unsigned long long) blocker + 1ull : (unsigned long long) fa.n_obj);
                            skip |= __ballot(valid && blocker != NO_BLOCKER);
                        } else if (COUNT && valid) {
                            cnt.add(3, fa.n_obj);
                        }
                    } else { // ---- point light: include/light_impl.h:19-21, (light - point) through FP32 ----
                        const D3 p{hp[hs], hp[WG + hs], hp[2 * WG + hs]}; // re-read: not kept in registers across lights
                        // Same argument for a point light behind the surface: the shading term is max(0, (float) dot(n, normalize(l - p)))
                        // (include/light_impl.h:38-43), normalisation scales by a positive factor, so the sign is that of q = dot(n, l - p)
                        // unless q is lost in rounding -- lanes sit the TEST out only when q < 0 by a margin 10^7 times the rounding error of
                        // either form (their term is then shaded as +0).
                        const double dx = lk.p[0] - p.x, dy = lk.p[1] - p.y, dz = lk.p[2] - p.z;
                        const double q = dot3(nrm, D3{dx, dy, dz});
                        const double mag = fabs(nrm.x * dx) + fabs(nrm.y * dy) + fabs(nrm.z * dz);
                        const bool wanted = valid && !(q < -1e-9 * mag);
                        skip = 0ull;
                        if (COUNT || __any(wanted)) {
                            Mono sd = sm;
                            mono_set_d<NEED_CROSS>(sd, D3{(double) (float) dx, (double) (float) dy, (double) (float) dz});
                            mono_set_od<NEED_CROSS>(sd);
                            if (wanted) cnt.traced();
                            const int blocker = shadow_blocker<COUNT, HAS_GQ, HAS_CUBIC, true, OWNG>(fa, S, gobj, sd, 1.0, COUNT ? valid : wanted, wanted, sball + c, sbox + c, my_crec, lk, lane, cnt,
                                                                                                     0u, false, 0ull, cub_rec0);
                            if (valid) cnt.add(3, blocker != NO_BLOCKER ? (unsigned long long) blocker + 1ull : (unsigned long long) fa.n_obj);
                            skip = __ballot(valid && blocker != NO_BLOCKER);
                        } else if (COUNT && valid) {
                            cnt.add(3, fa.n_obj);
                        }
                    }
                    if (lane == 0) sblk[c * fa.n_lights + l] = skip;
                }
            }
            if (first && listing && lane == 0) atomicMax(&s_cost[0], (uint32_t) ((__builtin_amdgcn_s_memtime() - b_t0) >> 6) + 1u);
            RT_STAMP(6);
            lds_barrier();
            RT_STAMP(7);
            // (of a split tile's two workgroups the first to get here speaks for the tile: whichever half has hits at all keeps it listed)
            if (first && listing && tid == 0 &&
                (s_half[0] == 0u || atomicMax(&fa.order_state[3u * (size_t) fa.ord_stride + tile], fa.ord_frame) != fa.ord_frame)) {
                // cost class 0 (the costliest) .. 15, on the scale of the previous frame's largest cost; one device atomic per tile with hits
                // (plus one for the scale), their results only needed at the very end of the workgroup.  A half tile reports twice its own
                // cost -- what the whole tile would have cost, or a little more: once split, a tile stays split.
                const uint32_t cost = s_half[0] ? 2u * s_cost[0] : s_cost[0];
                const uint32_t q = cost_scale ? (uint32_t) (((unsigned long long) cost * ORD_CLASSES) / ((unsigned long long) cost_scale + 1ull)) : 0u; // (no scale yet: the cheapest class -- nothing is split on a guess)
                ord_cls = ORD_CLASSES - (q < ORD_CLASSES - 1u ? q : ORD_CLASSES - 1u); // class + 1
                ord_pos = atomicAdd(&ord_wr[ord_cls - 1u], 1u);
                // (only candidates for the maximum bother the counter -- and one tile in sixteen whatever its cost, the census' sample: when every
                // cost has dropped below the old scale, e.g. after a frame whose tiles were all split and reported twice a half's cost, the next
                // scale is the sample's maximum instead of nothing.  A scale that restarted from 0 put every tile into the costliest class: all
                // split, in arrival order, inflated costs, no candidate again -- a three-frame cycle of 43 / 50 / 72 us at orbit pose 6.)
                if (cost > cost_scale - cost_scale / 4u || ((tile * 0x9E3779B1u) >> 28) == 0u) atomicMax(&ord_wr[17], cost);
            }

            // ---------------- phase C: shade each hit, lights in order ----------------
            // (hit h is lane h & 63 of wave h >> 6: a wave shades the chunk with its own number, and a light's mask is one broadcast read)
            if (wave < n_chunks) { // wave-uniform
                const uint32_t h = tid;
                const bool v = h < n_hits;
                const uint32_t hs = v ? h : wave * 64u;
                const D3 n{hn[hs], hn[WG + hs], hn[2 * WG + hs]};
                const uint32_t hinfo = hidx[hs];
                const MatEntry mt = S.mat[(hinfo >> 8) & 0x7FFFFFu];
                const F3 aop{mt.albedo[0] / PI_F, mt.albedo[1] / PI_F, mt.albedo[2] / PI_F}; // object_color / pi, once per hit
                F3 acc{0.0f, 0.0f, 0.0f};
                for (uint32_t l = 0; l < fa.n_lights; l++) {
                    RT_LOAD_LIGHTK(lk, l)
                    const bool lit = v && !((sblk[wave * fa.n_lights + l] >> lane) & 1ull);
                    if (!(lk.flags & 1u)) { // directional: ((albedo / pi) * colour) * max(0, n.l), left to right (include/light_impl.h:43)
                        if (lit) {
                            const float lam = (float) dot3(n, D3{lk.p[0], lk.p[1], lk.p[2]});
                            const float mx = (0.0f < lam) ? lam : 0.0f;
                            cnt.shaded();
                            acc.x += aop.x * lk.color[0] * mx;
                            acc.y += aop.y * lk.color[1] * mx;
                            acc.z += aop.z * lk.color[2] * mx;
                        }
                    } else if (lit) {
                        const D3 p{hp[hs], hp[WG + hs], hp[2 * WG + hs]};
                        const double lp[3] = {lk.p[0], lk.p[1], lk.p[2]};
                        const float lc[3] = {lk.color[0], lk.color[1], lk.color[2]};
                        const F3 col = surface_color_pre(lp, lc, true, p, n, aop);
                        cnt.shaded();
                        acc.x += col.x;
                        acc.y += col.y;
                        acc.z += col.z;
                    }
                }
                if (v) {
                    const uint32_t px = hinfo & 255u;
                    scolor[px] = (acc.x < 1.0f) ? acc.x : 1.0f; // glm::min(vec3(1.0f), acc)
                    scolor[WG + px] = (acc.y < 1.0f) ? acc.y : 1.0f;
                    scolor[2 * WG + px] = (acc.z < 1.0f) ? acc.z : 1.0f;
                }
            }
            RT_STAMP(8);
            lds_barrier();

            // ---------------- phase D: blend, set up the bounce ----------------
            if (HAS_MIRROR) {
                res = F3{spark[tid], spark[WG + tid], spark[2 * WG + tid]};
                cur_ratio = spark[3 * WG + tid];
                n_refl = reinterpret_cast<const uint32_t *>(spark)[4 * WG + tid];
                // ray direction and origin are re-defined here for EVERY lane (only bouncing lanes use them again), so that
                // neither is live across phases B and C: 12 VGPRs the register allocator otherwise has to carry
                dir = D3{hdir[tid], hdir[WG + tid], hdir[2 * WG + tid]};
                o = D3{0.0, 0.0, 0.0};
            }
            if (hit) {
                const F3 oc{scolor[tid], scolor[WG + tid], scolor[2 * WG + tid]};
                if (first) res = oc;
                else blend(res, cur_ratio, oc);
                const float refl = HAS_MIRROR ? S.mat[(hidx[my_slot] >> 8) & 0x7FFFFFu].refl : 0.0f; // `best` is not kept across B / C
                if (!HAS_MIRROR || !((double) refl > EPS)) {
                    live = false;
                } else {
                    cur_ratio *= refl;
                    if (n_refl == fa.max_refl) {
                        blend(res, cur_ratio, bg);
                        live = false;
                    } else {
                        n_refl++;
                        // hit point and normal come back from this lane's queue slot (kept out of registers across B / C)
                        const D3 sp{hp[my_slot], hp[WG + my_slot], hp[2 * WG + my_slot]};
                        const D3 sn{hn[my_slot], hn[WG + my_slot], hn[2 * WG + my_slot]};
                        dir = reflect_ray(dir, sn);
                        cnt.add(2);
                        o = D3{sp.x + SHADOW_BIAS * sn.x, sp.y + SHADOW_BIAS * sn.y, sp.z + SHADOW_BIAS * sn.z};
                    }
                }
            }
            if (!HAS_MIRROR) break; // no mirrors: nothing can still be bouncing
            first = false; // wave-uniform: round 0 is over for everybody
            // any pixel of the tile still bouncing?  one flag per wave, one barrier (which also orders the queue
            // reads of this round before the next round's writes)
            const unsigned long long live_mask = __ballot(live); // all lanes vote, then lane 0 publishes
            if (lane == 0) s_live[wave] = live_mask != 0ull ? 1u : 0u;
            lds_barrier();
            const uint32_t more = s_live[0] | s_live[1] | s_live[2] | s_live[3];
            RT_STAMP(9);
            if (!more) break;
        }
        } // !LEAN

        // the pixel's coordinates are formed again here (from an opaque copy of tid) instead of being kept in registers
        // through all the rounds; one more lever that keeps the mirror instantiations free of scratch spills
        // ... and so are the frame's sizes: read through a fresh pointer to the arguments, they are short-lived scalar loads here instead of
        // SGPRs (or, as it happened, a spill slot that was rematerialised away but still cost the kernel a private segment) across all phases
        // (Only where registers are the limit -- spheres and planes without mirrors, six workgroups per CU; the other instantiations lose a per cent with it.)
        constexpr bool LEAN_REGS = !HAS_GQ && !HAS_MIRROR; // (and the degree-3 one without general quadrics: 128 VGPRs, four workgroups per CU)
        const FrameArgs &fe = LEAN_REGS ? cold_args().fa : fa;
        const uint32_t tid_ = LEAN_REGS ? tid ^ s_zero[0] : tid; // (an LDS word that is always 0: the compiler cannot know, so it cannot keep x / y of round 0 alive instead)
        const uint32_t sx_ = (tile % fe.tiles_x) * RT_TILE + tile_px(tid_), sy_ = (tile / fe.tiles_x) * RT_TILE + tile_py(tid_);
        if (fe.sparse) { // launch-uniform: fb is a message (rt_pack_sparse's layout); only tiles with round-0 hits are in it
            const uint32_t mslot = s_sparse[0]; // workgroup-uniform
            uint32_t *msg = reinterpret_cast<uint32_t *>(fb);
            if (mslot < fe.sparse_cap) {
                uchar4 px;
                px.x = (unsigned char) (int) (res.x * 255.0f + 0.5f);
                px.y = (unsigned char) (int) (res.y * 255.0f + 0.5f);
                px.z = (unsigned char) (int) (res.z * 255.0f + 0.5f);
                px.w = 255;
                const uint32_t off_tiles = (4u + fe.sparse_cap + 3u) & ~3u;
                reinterpret_cast<uchar4 *>(msg + off_tiles)[(size_t) mslot * 256u + tile_py(tid_) * 16u + tile_px(tid_)] = px;
                if (tid == 0) msg[4u + mslot] = tile;
            } else if (mslot != 0xFFFFFFFFu && tid == 0) {
                msg[1] = 1u; // more tiles with hits than the message holds
            }
        } else if (sx_ < fe.width && sy_ < fe.local_rows && (s_half[0] == 0u || (tid_ >> 7) + 1u == s_half[0])) { // (a half tile: the other rows are another workgroup's)
            const size_t pix = (size_t) sy_ * fe.width + sx_;
            if (fe.rgba8) {
                uchar4 px;
                px.x = (unsigned char) (int) (res.x * 255.0f + 0.5f);
                px.y = (unsigned char) (int) (res.y * 255.0f + 0.5f);
                px.z = (unsigned char) (int) (res.z * 255.0f + 0.5f);
                px.w = 255;
                reinterpret_cast<uchar4 *>(fb)[pix] = px;
            } else {
                reinterpret_cast<float4 *>(fb)[pix] = make_float4(res.x, res.y, res.z, 1.0f);
            }
        }
        if (ord_cls && ord_pos < fe.n_tiles) { // thread 0 of a tile with hits, ordering on (the bound can only fail if frames
                                                // were replayed with stale arguments, e.g. from a captured graph: stay in bounds)
            ord_wr[RT_ORD_HDR + ord_cls * fe.n_tiles + ord_pos] = tile; // ord_cls = class + 1
            ord_wr[RT_ORD_HDR + tile] = (ord_pos << 5) | ord_cls;
        }
        RT_STAMP(10);
    }

    RT_STAMP_FLUSH(counters, lane);
    cnt.flush(counters);
}


} // namespace RT_SYM(rtw)

extern "C" size_t RT_SYM(rt_wavefront_lds_bytes)(uint32_t stage_bytes, uint32_t n_lights, int has_mirror, uint32_t n_cull_spheres, int lean, uint32_t n_cub)
{
    return RT_SYM(rtw)::LdsLayout(stage_bytes, n_lights, has_mirror != 0, n_cull_spheres, lean != 0, n_cub).total;
}

// One workgroup per 16x16 tile; the dispatcher hands tiles to CUs as they free up, which is the dynamic load
// balancing this workload needs (a tile full of hits costs ~20x an empty one).  Two persistent-workgroup variants
// were measured and dropped: a global tile counter serialises at ~90 same-address atomics/us (8160 tiles -> 93 us),
// static striding does not balance ("Experiments that did not pay").  The grid carries ord_cap extra slots in
// front for the tiles that had hits in the previous frame (see "launch order from the previous frame" in the kernel).
extern "C" hipError_t RT_SYM(rt_launch_wavefront)(const FrameArgs *fa, const DevObject *gobj, const DevLight *glight,
                                                   void *fb, unsigned long long *counters, int count,
                                                   const double *camx, const double *camy, hipStream_t stream)
{
    using namespace RT_SYM(rtw);
    if (fa->n_tiles == 0) return hipSuccess;
    const uint32_t n_scan = fa->tile_state ? fa->n_scan : 0u; // classifying workgroups; with them: paint workgroups, unless the output is sparse
    const uint32_t n_paint = (n_scan && !fa->sparse) ? (fa->n_tiles + RT_PAINT_TILES - 1u) / RT_PAINT_TILES : 0u;
    // index region: groups of one paint workgroup + RT_PAINT_TILES index slots (the last group may be partial: surplus slots leave)
    const uint32_t n_index = n_paint ? n_paint * (RT_PAINT_TILES + 1u) : fa->n_tiles;
    const dim3 grid(n_scan + ((fa->order_state && fa->ord_on) ? fa->ord_cap : 0u) + n_index), block(WG);
    const unsigned char *gs = reinterpret_cast<const unsigned char *>(gobj);
    int sel = (count ? 8 : 0) | (fa->has_mirror ? 4 : 0) | (fa->n_gq ? 2 : 0) | (fa->n_cub ? 1 : 0);
    // the lean path (rt_render decides: unit spheres only, all cullable, no mirrors, dense output): instantiations 16 / 17
    const bool lean = fa->lean && (sel & 7) == 0 && fa->all_cullable && fa->n_us == fa->n_obj && !fa->sparse;
    if (lean) sel = count ? 17 : 16;
    const size_t lds = LdsLayout(fa->stage_bytes, fa->n_lights, fa->has_mirror != 0, fa->cull ? fa->n_us : 0u, lean, fa->n_cub).total;
    const bool ordering = fa->order_state && fa->ord_on;
    const unsigned char *hot_us = gs + fa->off_us;
    const uint32_t *hot_ord_rd = ordering ? fa->order_state + (size_t) fa->ord_read * fa->ord_stride : nullptr;
    uint32_t hot_flags = (fa->all_cullable ? 1u : 0u) | (ordering ? 2u : 0u) | (n_paint ? 4u : 0u) | (fa->ord_plain ? 8u : 0u) |
                         ((fa->ord_split & 15u) << 4); // bits 4-7: cost classes whose tiles may be rendered as two half tiles
#ifdef RT_WF_DEBUG_EXITS
    if (const char *dbg = getenv("MI355RT_DEBUG_EXIT")) hot_flags |= (uint32_t) atoi(dbg) << 8;
#endif
    if (fa->ord_split && ordering) {
        // workgroup slots of the whole GPU for this instantiation and LDS size (bits 20-31): tiles are only split into half tiles while
        // there are slots to spare.  Asked once per (device, instantiation, LDS size) and thread.
        struct SlotsKey { int dev, sel; size_t lds; uint32_t slots; };
        static thread_local SlotsKey memo[4] = {{-1, 0, 0, 0}, {-1, 0, 0, 0}, {-1, 0, 0, 0}, {-1, 0, 0, 0}};
        static thread_local unsigned memo_next = 0;
        int dev = 0;
        (void) hipGetDevice(&dev);
        uint32_t slots = 0;
        bool found = false;
        for (const SlotsKey &k : memo)
            if (k.dev == dev && k.sel == sel && k.lds == lds) { slots = k.slots; found = true; }
        if (!found) {
            int per_cu = 0, cus = 0;
#define RT_OCC(C, M, G, Q) (void) hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, wavefront_tile_kernel<C, G, Q, M>, WG, lds)
            switch (sel) {
            case 0: RT_OCC(false, false, false, false); break;
            case 1: RT_OCC(false, false, false, true); break;
            case 2: RT_OCC(false, false, true, false); break;
            case 3: RT_OCC(false, false, true, true); break;
            case 4: RT_OCC(false, true, false, false); break;
            case 5: RT_OCC(false, true, false, true); break;
            case 6: RT_OCC(false, true, true, false); break;
            case 7: RT_OCC(false, true, true, true); break;
            default: break; // counting builds: no split (they are not timed)
            }
#undef RT_OCC
            (void) hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
            const long long total = (long long) (per_cu > 0 ? per_cu : 0) * (cus > 0 ? cus : 0);
            slots = (uint32_t) (total > 4095 ? 4095 : total);
            memo[memo_next++ & 3u] = SlotsKey{dev, sel, lds, slots};
        }
        hot_flags |= slots << 20;
    }
    const ColdArgs cold{*fa, gs, glight, fb, counters, camx, camy};
#define RT_LAUNCH(C, M, G, Q) hipLaunchKernelGGL((wavefront_tile_kernel<C, G, Q, M>), grid, block, lds, stream, hot_us, hot_ord_rd, fa->n_us, fa->ord_cap, fa->n_tiles, hot_flags, n_scan ? fa->tile_state : nullptr, fa->frame_tag, n_scan, cold)
    switch (sel) {
    case 0: RT_LAUNCH(false, false, false, false); break;
    case 1: RT_LAUNCH(false, false, false, true); break;
    case 2: RT_LAUNCH(false, false, true, false); break;
    case 3: RT_LAUNCH(false, false, true, true); break;
    case 4: RT_LAUNCH(false, true, false, false); break;
    case 5: RT_LAUNCH(false, true, false, true); break;
    case 6: RT_LAUNCH(false, true, true, false); break;
    case 7: RT_LAUNCH(false, true, true, true); break;
    case 8: RT_LAUNCH(true, false, false, false); break;
    case 9: RT_LAUNCH(true, false, false, true); break;
    case 10: RT_LAUNCH(true, false, true, false); break;
    case 11: RT_LAUNCH(true, false, true, true); break;
    case 12: RT_LAUNCH(true, true, false, false); break;
    case 13: RT_LAUNCH(true, true, false, true); break;
    case 14: RT_LAUNCH(true, true, true, false); break;
    case 15: RT_LAUNCH(true, true, true, true); break;
    case 16: hipLaunchKernelGGL((wavefront_tile_kernel<false, false, false, false, true>), grid, block, lds, stream, hot_us, hot_ord_rd, fa->n_us, fa->ord_cap, fa->n_tiles, hot_flags, n_scan ? fa->tile_state : nullptr, fa->frame_tag, n_scan, cold); break;
    default: hipLaunchKernelGGL((wavefront_tile_kernel<true, false, false, false, true>), grid, block, lds, stream, hot_us, hot_ord_rd, fa->n_us, fa->ord_cap, fa->n_tiles, hot_flags, n_scan ? fa->tile_state : nullptr, fa->frame_tag, n_scan, cold); break;
    }
#undef RT_LAUNCH
    return hipGetLastError();
}
