// rt_kernels.hip -- HIP kernels of the per-pixel ray-tracing path for gfx950 (MI355X, wave64).
//
// Compiled twice by the Makefile:
//   -DRT_VARIANT=strict -ffp-contract=off    parity mode (separately rounded IEEE operations)
//   -DRT_VARIANT=fast   -ffp-contract=fast   FMA contraction allowed
// Each build exports rt_launch_trace_<variant>() / rt_launch_assemble_<variant>() to rt_capi.cpp.
//
// Replaces the reference's only kernel, update_kernel (src/update-cuda.cu:104-158), and its device
// helper get_color_and_object (:65-102).  Not a translation of it: see the notes in rt_math.hpp and
// DESIGN.md ("Kernels").
#include <hip/hip_runtime.h>

#include "rt_math.hpp"
#include "rt_scene_dev.h"

#ifndef RT_VARIANT
#error "define RT_VARIANT=strict|fast"
#endif
#define RT_CAT2(a, b) a##_##b
#define RT_CAT(a, b) RT_CAT2(a, b)
#define RT_SYM(name) RT_CAT(name, RT_VARIANT)

namespace RT_SYM(rtk) {

using namespace rtm;

// Per-lane work counters (RT_FLAG_COUNT builds only).
template <bool COUNT>
struct Cnt {
    __device__ __forceinline__ void primary() {}
    __device__ __forceinline__ void shadow() {}
    __device__ __forceinline__ void reflect() {}
    __device__ __forceinline__ void test() {}
    __device__ __forceinline__ void hit() {}
    __device__ __forceinline__ void solve() {}
    __device__ __forceinline__ void flush(unsigned long long *) {}
};
template <>
struct Cnt<true> {
    unsigned long long v[6] = {0, 0, 0, 0, 0, 0};
    __device__ __forceinline__ void primary() { v[0]++; }
    __device__ __forceinline__ void shadow() { v[1]++; }
    __device__ __forceinline__ void reflect() { v[2]++; }
    __device__ __forceinline__ void test() { v[3]++; }
    __device__ __forceinline__ void hit() { v[4]++; }
    __device__ __forceinline__ void solve() { v[5]++; }
    __device__ __forceinline__ void flush(unsigned long long *g)
    {
        for (int i = 0; i < 6; i++)
            if (v[i]) atomicAdd(&g[i], v[i]);
    }
};

// Nearest hit + direct lighting for one ray: get_color_and_object, src/update-cpu.cpp:45-80
// (SURVEY.md Q7, Q12).  gobj/glight are the scene in global memory, indexed wave-uniformly (the compiler
// turns those reads into scalar loads: operands arrive in SGPRs, no VGPR or LDS bandwidth spent on
// them); sobj is the same scene staged in LDS for the reads whose index differs per lane.
template <bool COUNT>
__device__ __forceinline__ int trace(const FrameArgs &fa, const DevObject *__restrict__ gobj,
                                     const DevLight *__restrict__ glight, const DevObject *sobj, const D3 &o,
                                     const D3 &d, F3 &color, D3 &sp, D3 &sn, Cnt<COUNT> &cnt)
{
    Mono m;
    make_mono(m, o, d);
    int best = -1;
    double best_t = INFINITY;
    for (uint32_t k = 0; k < fa.n_obj; k++) {
        double t = intersect(gobj[k].c, gobj[k].cls, m, MAX_T, false);
        cnt.test();
        if (t >= EPS && t < MAX_T && t < best_t) {
            best_t = t;
            best = (int) k;
        }
    }
    if (best < 0) return -1;

    cnt.hit();
    sp = D3{o.x + best_t * d.x, o.y + best_t * d.y, o.z + best_t * d.z};
    const DevObject *bo = &sobj[best]; // per-lane index: LDS gather
    sn = normal_vector(bo->c, sp);
    const F3 albedo{bo->albedo[0], bo->albedo[1], bo->albedo[2]};
    const D3 so{sp.x + SHADOW_BIAS * sn.x, sp.y + SHADOW_BIAS * sn.y, sp.z + SHADOW_BIAS * sn.z};
    F3 acc{0.0f, 0.0f, 0.0f};
    for (uint32_t l = 0; l < fa.n_lights; l++) {
        const DevLight *lt = &glight[l];
        const bool spherical = lt->spherical != 0;
        double max_t;
        D3 sd = shadow_dir(lt->p, spherical, sp, max_t);
        cnt.shadow();
        Mono sm;
        make_mono(sm, so, sd);
        bool in_shadow = false;
        for (uint32_t k = 0; k < fa.n_obj; k++) {
            double t = intersect(gobj[k].c, gobj[k].cls, sm, max_t, true);
            cnt.test();
            if (t > EPS && t < max_t) {
                in_shadow = true;
                break;
            }
        }
        if (!in_shadow) {
            F3 c = surface_color(lt->p, lt->color, spherical, sp, sn, albedo);
            acc.x += c.x;
            acc.y += c.y;
            acc.z += c.z;
        }
    }
    // glm::min(vec3(1.0f), acc)
    color.x = (acc.x < 1.0f) ? acc.x : 1.0f;
    color.y = (acc.y < 1.0f) ? acc.y : 1.0f;
    color.z = (acc.z < 1.0f) ? acc.z : 1.0f;
    return best;
}

__device__ __forceinline__ void blend(F3 &res, float ratio, const F3 &c)
{
    // UPDATE_COLOR, src/update-cpu.cpp:100
    res.x = (1.0f - ratio) * res.x + ratio * c.x;
    res.y = (1.0f - ratio) * res.y + ratio * c.y;
    res.z = (1.0f - ratio) * res.z + ratio * c.z;
}

// One workgroup = one 16x16 pixel tile = 4 waves of 8x8 pixels (square tiles keep the lanes of a wave
// on the same objects; a row of 8 RGBA32F pixels is one full 128-byte line).
// blockIdx.x enumerates tiles row-major over this rank's local rows; consecutive workgroups go to
// consecutive XCDs, which spreads hit-heavy neighbourhoods over all 8 XCDs.
template <bool COUNT, bool RGBA8>
__global__ __launch_bounds__(256) void trace_tile_kernel(const FrameArgs fa, const DevObject *__restrict__ gobj,
                                                          const DevLight *__restrict__ glight,
                                                          void *__restrict__ fb,
                                                          unsigned long long *__restrict__ counters)
{
    extern __shared__ __align__(16) unsigned char smem[];
    DevObject *sobj = reinterpret_cast<DevObject *>(smem);

    // stage the object records into LDS once per workgroup (16-byte copies)
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(gobj);
        uint4 *dst = reinterpret_cast<uint4 *>(smem);
        const uint32_t n16 = fa.n_obj * (uint32_t) (sizeof(DevObject) / 16);
        for (uint32_t i = threadIdx.x; i < n16; i += blockDim.x) dst[i] = src[i];
    }
    __syncthreads();

    const uint32_t tile_x = blockIdx.x % fa.tiles_x;
    const uint32_t tile_y = blockIdx.x / fa.tiles_x;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t x = tile_x * RT_TILE + (wave & 1) * 8 + (lane & 7);
    const uint32_t lr = tile_y * RT_TILE + (wave >> 1) * 8 + (lane >> 3);
    if (x >= fa.width || lr >= fa.local_rows) return;
    const uint32_t y = global_row(fa, lr);

    Cnt<COUNT> cnt;
    const D3 origin{fa.origin[0], fa.origin[1], fa.origin[2]};
    D3 dir = primary_dir(fa, (int) x, (int) y);
    cnt.primary();

    // render_pixel, src/update-cpu.cpp:82-119, written as ONE bounce loop so that the
    // trace code exists once: iteration 0 is the primary ray, iteration k the k-th mirror bounce.  The loop
    // ends for the wave when no lane is still bouncing (exec mask empty).
    const F3 bg{fa.bg[0], fa.bg[1], fa.bg[2]};
    F3 res = bg;
    D3 o = origin;
    float cur_ratio = 1.0f;
    uint32_t n_refl = 0;
    bool first = true;
    for (;;) {
        F3 oc;
        D3 sp, sn;
        const int idx = trace<COUNT>(fa, gobj, glight, sobj, o, dir, oc, sp, sn, cnt);
        if (idx < 0) {
            if (!first) blend(res, cur_ratio, bg); // a bounce that leaves the scene picks up the background
            break;
        }
        if (first) res = oc;
        else blend(res, cur_ratio, oc);
        first = false;
        const float refl = sobj[idx].refl;
        if (!((double) refl > EPS)) break;
        cur_ratio *= refl;
        if (n_refl == fa.max_refl) {
            blend(res, cur_ratio, bg);
            break;
        }
        n_refl++;
        dir = reflect_ray(dir, sn);
        cnt.reflect();
        o = D3{sp.x + SHADOW_BIAS * sn.x, sp.y + SHADOW_BIAS * sn.y, sp.z + SHADOW_BIAS * sn.z};
    }

    const size_t pix = (size_t) lr * fa.width + x;
    if (RGBA8) {
        // wire format of src/update-cuda.cu:149-156: iround(c * 255), alpha 255
        uchar4 px;
        px.x = (unsigned char) (int) (res.x * 255.0f + 0.5f);
        px.y = (unsigned char) (int) (res.y * 255.0f + 0.5f);
        px.z = (unsigned char) (int) (res.z * 255.0f + 0.5f);
        px.w = 255;
        reinterpret_cast<uchar4 *>(fb)[pix] = px;
    } else {
        reinterpret_cast<float4 *>(fb)[pix] = make_float4(res.x, res.y, res.z, 1.0f);
    }
    cnt.flush(counters);
}

// Root-side reassembly after the gather: gathered[rank][local_row][x] -> full[y][x] (16-byte or
// 4-byte pixels).  Pure HBM streaming: one read + one write of the frame.
template <typename PIX>
__global__ __launch_bounds__(256) void assemble_kernel(const PIX *__restrict__ gathered, PIX *__restrict__ full,
                                                        uint32_t width, uint32_t height, uint32_t world,
                                                        uint32_t band_rows, uint32_t max_local_rows)
{
    const size_t n = (size_t) width * height;
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) {
        const uint32_t y = (uint32_t) (i / width), x = (uint32_t) (i - (size_t) y * width);
        const uint32_t band = y / band_rows, r = band % world;
        const uint32_t lr = (band / world) * band_rows + (y - band * band_rows);
        full[i] = gathered[((size_t) r * max_local_rows + lr) * width + x];
    }
}

// ---- sparse frame transport (RGBA8) ------------------------------------------------------------------------------------
// Most tiles of a typical frame are pure background, and the gather to the root (the only collective of the path) is
// bound by the xGMI links, so a rank can send only its 16x16 tiles that contain something else:
//   message = { count, overflow, capacity, 0 } + ids[capacity] (padded to 16 B) + tiles[capacity][256] pixels,
// a fixed size per rank (collectives want equal sizes); `overflow` is set when more than `capacity` tiles qualify.
// pack_sparse: one wave per local tile, one 16-byte piece (4 pixels of a tile row) per lane.
__global__ __launch_bounds__(64) void pack_sparse_kernel(const uint32_t *__restrict__ fb, uint32_t width, uint32_t local_rows, uint32_t tiles_x,
                                                         uint32_t bg, uint32_t cap, uint32_t *__restrict__ msg)
{
    const uint32_t tile = blockIdx.x, lane = threadIdx.x;
    const uint32_t x = (tile % tiles_x) * 16u + (lane & 3u) * 4u, lr = (tile / tiles_x) * 16u + (lane >> 2);
    uint4 v = make_uint4(bg, bg, bg, bg);
    if (lr < local_rows) {
        const uint32_t *row = fb + (size_t) lr * width;
        if ((width & 3u) == 0u && x + 3u < width) {
            v = *reinterpret_cast<const uint4 *>(row + x);
        } else {
            if (x < width) v.x = row[x];
            if (x + 1u < width) v.y = row[x + 1u];
            if (x + 2u < width) v.z = row[x + 2u];
            if (x + 3u < width) v.w = row[x + 3u];
        }
    }
    const unsigned long long any = __ballot(v.x != bg || v.y != bg || v.z != bg || v.w != bg);
    if (any == 0ull) return; // wave-uniform
    uint32_t pos = 0;
    if (lane == 0) pos = atomicAdd(&msg[0], 1u);
    pos = __builtin_amdgcn_readfirstlane(pos);
    if (pos >= cap) {
        if (lane == 0) msg[1] = 1u;
        return;
    }
    const uint32_t off_tiles = (4u + cap + 3u) & ~3u;
    if (lane == 0) msg[4u + pos] = tile;
    reinterpret_cast<uint4 *>(msg + off_tiles)[(size_t) pos * 64u + lane] = v;
}

__global__ __launch_bounds__(256) void fill_kernel(uint32_t *__restrict__ full, size_t n, uint32_t bg)
{
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) full[i] = bg;
}

// Incremental reassembly: the frame buffer keeps what earlier frames scattered into it, so only tiles that an EARLIER
// frame delivered and this one did not have to go back to the background.  stamps[r][t] = tag of the last frame that
// delivered tile t of rank r (0 = the tile is background in the buffer).  One wave per 64 (rank, tile) entries looks;
// a stale entry is repainted by its whole wave, one 16-byte piece per lane (rare: the camera moved away from it).
__global__ __launch_bounds__(64) void clear_stale_kernel(uint32_t *__restrict__ full, uint32_t *__restrict__ stamps, uint32_t width, uint32_t height,
                                                         uint32_t world, uint32_t band_rows, uint32_t tiles_x, uint32_t max_tiles, uint32_t tag, uint32_t bg)
{
    const uint32_t lane = threadIdx.x;
    const size_t e = (size_t) blockIdx.x * 64u + lane, n = (size_t) world * max_tiles;
    uint32_t st = 0;
    if (e < n) st = stamps[e];
    unsigned long long stale = __ballot(st != 0u && st != tag);
    if (st != 0u && st != tag) stamps[e] = 0u;
    while (stale) { // wave-uniform
        const uint32_t l = (uint32_t) __builtin_ctzll(stale);
        stale &= stale - 1ull;
        const size_t es = (size_t) blockIdx.x * 64u + l;
        const uint32_t r = (uint32_t) (es / max_tiles), tile = (uint32_t) (es - (size_t) r * max_tiles);
        const uint32_t x = (tile % tiles_x) * 16u + (lane & 3u) * 4u, lr = (tile / tiles_x) * 16u + (lane >> 2);
        const uint32_t b = lr / band_rows;
        const uint64_t y = ((uint64_t) b * world + r) * band_rows + (lr - b * band_rows);
        if (y >= height) continue;
        uint32_t *row = full + (size_t) y * width;
        if (x < width) row[x] = bg;
        if (x + 1u < width) row[x + 1u] = bg;
        if (x + 2u < width) row[x + 2u] = bg;
        if (x + 3u < width) row[x + 3u] = bg;
    }
}

// root: tile j of rank r's message goes back to its pixels of the full frame (band-cyclic rows, like assemble_kernel)
__global__ __launch_bounds__(64) void scatter_sparse_kernel(const uint32_t *__restrict__ gathered, uint32_t *__restrict__ full, uint32_t width,
                                                            uint32_t height, uint32_t world, uint32_t band_rows, uint32_t tiles_x, uint32_t cap,
                                                            uint32_t *__restrict__ stamps, uint32_t max_tiles, uint32_t tag)
{
    const uint32_t r = blockIdx.x / cap, j = blockIdx.x - r * cap, lane = threadIdx.x;
    const uint32_t off_tiles = (4u + cap + 3u) & ~3u;
    const size_t msg_words = (size_t) off_tiles + (size_t) cap * 256u;
    const uint32_t *msg = gathered + (size_t) r * msg_words;
    const uint32_t count = msg[0] < cap ? msg[0] : cap;
    if (j >= count) return;
    const uint32_t tile = msg[4u + j];
    if (stamps && lane == 0 && tile < max_tiles) stamps[(size_t) r * max_tiles + tile] = tag; // this frame delivered the tile
    const uint4 v = reinterpret_cast<const uint4 *>(msg + off_tiles)[(size_t) j * 64u + lane];
    const uint32_t x = (tile % tiles_x) * 16u + (lane & 3u) * 4u, lr = (tile / tiles_x) * 16u + (lane >> 2);
    const uint32_t b = lr / band_rows;
    const uint64_t y = ((uint64_t) b * world + r) * band_rows + (lr - b * band_rows);
    if (y >= height) return;
    uint32_t *row = full + (size_t) y * width;
    if (x < width) row[x] = v.x;
    if (x + 1u < width) row[x + 1u] = v.y;
    if (x + 2u < width) row[x + 2u] = v.z;
    if (x + 3u < width) row[x + 3u] = v.w;
}

} // namespace RT_SYM(rtk)

extern "C" hipError_t RT_SYM(rt_launch_trace)(const FrameArgs *fa, const DevObject *gobj, const DevLight *glight,
                                               void *fb, unsigned long long *counters, int rgba8, int count,
                                               hipStream_t stream)
{
    using namespace RT_SYM(rtk);
    const uint32_t tiles_y = (fa->local_rows + RT_TILE - 1) / RT_TILE;
    const dim3 grid(fa->tiles_x * tiles_y), block(256);
    if (grid.x == 0) return hipSuccess;
    const size_t lds = (size_t) fa->n_obj * sizeof(DevObject);
    if (count) {
        if (rgba8)
            hipLaunchKernelGGL((trace_tile_kernel<true, true>), grid, block, lds, stream, *fa, gobj, glight, fb, counters);
        else
            hipLaunchKernelGGL((trace_tile_kernel<true, false>), grid, block, lds, stream, *fa, gobj, glight, fb, counters);
    } else {
        if (rgba8)
            hipLaunchKernelGGL((trace_tile_kernel<false, true>), grid, block, lds, stream, *fa, gobj, glight, fb, counters);
        else
            hipLaunchKernelGGL((trace_tile_kernel<false, false>), grid, block, lds, stream, *fa, gobj, glight, fb, counters);
    }
    return hipGetLastError();
}

extern "C" hipError_t RT_SYM(rt_launch_assemble)(const void *gathered, void *full, uint32_t width, uint32_t height,
                                                  uint32_t world, uint32_t band_rows, uint32_t max_local_rows,
                                                  int rgba8, hipStream_t stream)
{
    using namespace RT_SYM(rtk);
    const size_t n = (size_t) width * height;
    if (n == 0) return hipSuccess;
    const uint32_t blocks = (uint32_t) ((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    if (rgba8)
        hipLaunchKernelGGL((assemble_kernel<uchar4>), dim3(blocks), dim3(256), 0, stream,
                           (const uchar4 *) gathered, (uchar4 *) full, width, height, world, band_rows, max_local_rows);
    else
        hipLaunchKernelGGL((assemble_kernel<float4>), dim3(blocks), dim3(256), 0, stream,
                           (const float4 *) gathered, (float4 *) full, width, height, world, band_rows, max_local_rows);
    return hipGetLastError();
}

extern "C" hipError_t RT_SYM(rt_launch_pack_sparse)(const void *fb, void *msg, uint32_t width, uint32_t local_rows, uint32_t bg, uint32_t cap,
                                                     hipStream_t stream)
{
    using namespace RT_SYM(rtk);
    hipError_t e = hipMemsetAsync(msg, 0, 16, stream); // count, overflow
    if (e != hipSuccess) return e;
    const uint32_t tiles_x = (width + 15u) / 16u, tiles_y = (local_rows + 15u) / 16u;
    if (tiles_x * tiles_y == 0u || cap == 0u) return hipSuccess;
    hipLaunchKernelGGL(pack_sparse_kernel, dim3(tiles_x * tiles_y), dim3(64), 0, stream, (const uint32_t *) fb, width, local_rows, tiles_x, bg, cap,
                       (uint32_t *) msg);
    return hipGetLastError();
}

// stamps == NULL: stateless (fill everything, scatter).  stamps != NULL: incremental -- `full` and `stamps` carry over from
// the previous call on this buffer (tag 0 = first call: fill everything and clear the stamps), max_tiles entries per rank.
extern "C" hipError_t RT_SYM(rt_launch_assemble_sparse)(const void *gathered, void *full, uint32_t width, uint32_t height, uint32_t world,
                                                         uint32_t band_rows, uint32_t bg, uint32_t cap, void *stamps, uint32_t max_tiles, uint32_t tag,
                                                         hipStream_t stream)
{
    using namespace RT_SYM(rtk);
    const size_t n = (size_t) width * height;
    if (n == 0) return hipSuccess;
    const uint32_t tiles_x = (width + 15u) / 16u;
    if (!stamps || tag == 0u) {
        const uint32_t blocks = (uint32_t) ((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
        hipLaunchKernelGGL(fill_kernel, dim3(blocks), dim3(256), 0, stream, (uint32_t *) full, n, bg);
        if (stamps) {
            hipError_t e = hipMemsetAsync(stamps, 0, sizeof(uint32_t) * (size_t) world * max_tiles, stream);
            if (e != hipSuccess) return e;
        }
    }
    const uint32_t use_tag = tag == 0u ? 0xFFFFFFFFu : tag; // the first call stamps with a value no later call may use
    if (cap != 0u && world != 0u)
        hipLaunchKernelGGL(scatter_sparse_kernel, dim3(world * cap), dim3(64), 0, stream, (const uint32_t *) gathered, (uint32_t *) full, width, height,
                           world, band_rows, tiles_x, cap, (uint32_t *) stamps, max_tiles, use_tag);
    if (stamps && tag != 0u && world != 0u && max_tiles != 0u) {
        const size_t entries = (size_t) world * max_tiles;
        hipLaunchKernelGGL(clear_stale_kernel, dim3((uint32_t) ((entries + 63) / 64)), dim3(64), 0, stream, (uint32_t *) full, (uint32_t *) stamps, width,
                           height, world, band_rows, tiles_x, max_tiles, use_tag, bg);
    }
    return hipGetLastError();
}
