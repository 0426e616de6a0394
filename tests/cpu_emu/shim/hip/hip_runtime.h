// tests/cpu_emu/shim/hip/hip_runtime.h — TEST INFRASTRUCTURE.
// Minimal stand-in for <hip/hip_runtime.h> so that the product's kernel source (rtc_kernels.hip) compiles
// as plain C++ and runs one "lane" at a time on the CPU.  Purpose: debug kernel logic and run ASan/UBSan
// on it without a GPU (GPU sanitizers are unavailable on the pool).  Never part of librtc_amd.so.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#define __device__
#define __global__
#define __host__
#define __forceinline__ inline
#define __noinline__
#define __shared__ static
#define __launch_bounds__(...)
struct dim3 { unsigned x, y, z; dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {} };
struct uint3_ { unsigned x, y, z; };
static uint3_ blockIdx, threadIdx, blockDim, gridDim;
typedef void* hipStream_t;
template <class T> static inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
// wave intrinsics for a one-lane "wave" (the emulator builds the persistent kernel with RTC_WAVE = 1)
static inline unsigned long long __ballot(int p) { return p ? 1ull : 0ull; }
static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline int __ffsll(long long x) { return __builtin_ffsll(x); }
template <class T> static inline T __shfl(T v, int) { return v; }
using std::fabs; using std::floor; using std::fmax; using std::fmin; using std::pow; using std::sqrt;
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...)          \
  do {                                                                       \
    gridDim.x = (grid).x; blockDim.x = (block).x;                            \
    for (unsigned b_ = 0; b_ < (grid).x; b_++)                               \
      for (unsigned t_ = 0; t_ < (block).x; t_++) {                          \
        blockIdx.x = b_; threadIdx.x = t_;                                   \
        kernel(__VA_ARGS__);                                                 \
      }                                                                      \
  } while (0)
