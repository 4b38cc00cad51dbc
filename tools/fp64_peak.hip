// fp64_peak.hip -- measures what the FP64 vector pipe of the GPU actually sustains (v_fma_f64 / v_mul_f64 /
// v_add_f64), to pin the "peak" that bench.py's roofline divides by.  Build: hipcc -O3 --offload-arch=gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(256) void burn(double *out, int iters, double a, double b)
{
    double x[16];
    for (int i = 0; i < 16; i++) x[i] = threadIdx.x * 1e-3 + i;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if (MODE == 0) x[i] = __builtin_fma(x[i], a, b);       // 2 flop / instr
            else if (MODE == 1) x[i] = x[i] * a;                    // 1 flop / instr
            else x[i] = x[i] + b;                                   // 1 flop / instr
        }
    }
    double s = 0;
    for (int i = 0; i < 16; i++) s += x[i];
    if (s == 12345.678) out[0] = s;
}

template <int MODE>
double run(const char *name, int flop_per_instr)
{
    double *d;
    hipMalloc(&d, 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int blocks = 256 * 8, iters = 20000;
    burn<MODE><<<blocks, 256>>>(d, 100, 1.0000001, 1e-9);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    burn<MODE><<<blocks, 256>>>(d, iters, 1.0000001, 1e-9);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double instr = (double) blocks * 256 * iters * 16;
    double tf = instr * flop_per_instr / (ms * 1e-3) / 1e12;
    printf("%-10s %8.3f ms  %7.2f TFLOP/s  (%.2f T lane-instr/s)\n", name, ms, tf, instr / (ms * 1e-3) / 1e12);
    hipFree(d);
    return tf;
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("device: %s, %d CUs, clock %d MHz\n", p.name, p.multiProcessorCount, p.clockRate / 1000);
    run<0>("fma_f64", 2);
    run<1>("mul_f64", 1);
    run<2>("add_f64", 1);
    return 0;
}
