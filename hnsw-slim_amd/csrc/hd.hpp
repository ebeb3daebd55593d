// hd.hpp -- host/device qualifier so the exact-order arithmetic and heap mechanics below are the
// SAME source for the gfx950 kernels and for the CPU-side unit tests / host index builder.
#pragma once
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define HS_HD __host__ __device__ __forceinline__
#else
#define HS_HD inline
#endif
#include <cstddef>
#include <cstdint>
