// launch_overheads.hip -- the fixed costs the frame schedule is built around, measured on the box:
//   1. stream-ordered kernel-to-kernel gap (dependent launches of a trivial kernel)
//   2. cost of workgroups that only read one word and leave (256 threads, 27 KB LDS, ~96 VGPRs: the footprint of the
//      render kernel's workgroups), for the tile counts of 1080p / 4K / 8K
//   3. filling a float4 frame with one colour: one workgroup per 16x16 tile vs. row-contiguous streaming
// Build: hipcc -O3 --offload-arch=gfx950 tools/launch_overheads.hip -o tools/bin/launch_overheads
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void tiny(unsigned *p) { if (threadIdx.x == 1000) p[0] = 1; }

__global__ __launch_bounds__(256, 5) void exit_only(const unsigned *count, float4 *out)
{
    extern __shared__ unsigned char smem[];
    asm volatile("" ::: "v95"); // allocate VGPRs like the render kernel's mirror-free instantiation
    const unsigned n = __builtin_amdgcn_readfirstlane(count[0]);
    if (blockIdx.x >= n) return;
    smem[threadIdx.x] = 1;
    __syncthreads();
    out[(size_t) blockIdx.x * 256 + threadIdx.x] = make_float4(smem[255 - threadIdx.x], 0, 0, 1);
}

// one workgroup per 16x16 tile, lane layout of the render kernel (8x8 block per wave)
__global__ __launch_bounds__(256) void fill_tiles(float4 *fb, unsigned width, unsigned tiles_x, float4 c)
{
    const unsigned tid = threadIdx.x, tile = blockIdx.x;
    const unsigned px = ((tid >> 6) & 1u) * 8u + (tid & 7u), py = (tid >> 7) * 8u + ((tid >> 3) & 7u);
    const unsigned x = (tile % tiles_x) * 16 + px, y = (tile / tiles_x) * 16 + py;
    fb[(size_t) y * width + x] = c;
}

// row-contiguous: each thread writes `per` float4s, consecutive threads consecutive addresses
__global__ __launch_bounds__(256) void fill_rows(float4 *fb, size_t n, float4 c)
{
    const size_t stride = (size_t) gridDim.x * 256;
    for (size_t i = (size_t) blockIdx.x * 256 + threadIdx.x; i < n; i += stride) fb[i] = c;
}

template <typename F>
static double time_us(F f, int reps, hipStream_t s)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; i++) f();
    CK(hipStreamSynchronize(s));
    std::vector<double> t;
    for (int r = 0; r < 5; r++) {
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < reps; i++) f();
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        t.push_back(ms * 1e3 / reps);
    }
    std::sort(t.begin(), t.end());
    return t[2];
}

int main()
{
    hipStream_t s;
    CK(hipStreamCreate(&s));
    unsigned *d_cnt;
    CK(hipMalloc(&d_cnt, 64));
    CK(hipMemset(d_cnt, 0, 64));
    float4 *fb;
    const size_t n8k = (size_t) 7680 * 4320;
    CK(hipMalloc(&fb, n8k * sizeof(float4)));
    printf("dependent launch of a trivial kernel: %.2f us per launch\n", time_us([&] { hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s, d_cnt); }, 200, s));
    const unsigned grids[3] = {8160, 32400, 129600};
    for (unsigned g : grids) {
        const double all_exit = time_us([&] { hipLaunchKernelGGL(exit_only, dim3(g), dim3(256), 27 * 1024, s, d_cnt, fb); }, 50, s);
        printf("%6u workgroups that read one word and leave (27 KB LDS, 96 VGPRs): %.2f us per launch\n", g, all_exit);
    }
    for (unsigned g : grids) {
        const double all_exit = time_us([&] { hipLaunchKernelGGL(exit_only, dim3(g), dim3(256), 0, s, d_cnt, fb); }, 50, s);
        printf("%6u workgroups that read one word and leave (no LDS): %.2f us per launch\n", g, all_exit);
    }
    const unsigned dims[3][2] = {{1920, 1080}, {3840, 2160}, {7680, 4320}};
    for (auto &d : dims) {
        const unsigned w = d[0], h = d[1], tx = w / 16, ty = (h + 15) / 16;
        const size_t n = (size_t) w * (ty * 16 <= 4320 ? h : h);
        const float4 c = make_float4(0.f, 0.1f, 0.2f, 1.f);
        const double a = time_us([&] { hipLaunchKernelGGL(fill_tiles, dim3(tx * (h / 16)), dim3(256), 0, s, fb, w, tx, c); }, 20, s);
        double best = 1e30;
        unsigned best_g = 0;
        for (unsigned g : {256u, 512u, 1024u, 2048u, 4096u, 8192u}) {
            const double b = time_us([&] { hipLaunchKernelGGL(fill_rows, dim3(g), dim3(256), 0, s, fb, n, c); }, 20, s);
            if (b < best) { best = b; best_g = g; }
        }
        printf("fill %ux%u float4 (%.1f MB): one workgroup per tile %.2f us (%.0f GB/s); row streaming %.2f us (%.0f GB/s, grid %u)\n", w, h, n * 16 / 1e6, a,
               n * 16 / a / 1e3, best, n * 16 / best / 1e3, best_g);
    }
    return 0;
}
