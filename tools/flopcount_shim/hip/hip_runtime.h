// Stand-in for <hip/hip_runtime.h> when OUR OWN device math headers (cuda-ray-tracer_amd/csrc/rt_math.hpp,
// rt_wavefront_math.hpp) are compiled for the host by tools/count_flops.cpp: the qualifiers become plain C++.
#pragma once
#define __device__
#define __host__
#define __forceinline__ inline
#define __noinline__
