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
struct float4 { float x, y, z, w; };
struct int4 { int x, y, z, w; };
typedef void* hipStream_t;
typedef void* hipEvent_t;
static inline int hipEventRecord(hipEvent_t, hipStream_t) { return 0; }
static inline int hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return 0; }
using std::fabs; using std::floor;
// fminf/fmaxf/fabsf come from <cmath> (C functions, NaN-ignoring like the device versions)
#define EMU_PLACEHOLDER
// ---- the product source's launch-geometry hooks (csrc/rtc_device.hpp), as this emulation needs them
#ifndef RTC_BLOCK
#define RTC_BLOCK 64
#endif
#ifndef RTC_BVH_STACK
#define RTC_BVH_STACK 64
#endif
#define RTC_LDS_STACK(name) static int name[RTC_BVH_STACK * RTC_BLOCK]
#define RTC_LAUNDER(x) do {} while (0)
#ifndef RTC_EMU_SIMT
#define RTC_LANE_ID 0           // sequential emulation: every lane is its own wave
#define RTC_WF_SHADE_BLOCK 1
#define RTC_WF_LANES 1u
#else
#define RTC_WF_SHADE_BLOCK 64   // thread-per-lane emulation: one-wave blocks
#endif
// The emulation gives the approximate reciprocal the error the hardware instruction may have (+-2^-23, pseudo-random per operand), so
// that the CPU parity tests exercise the margins of every decision built on it.
static inline double rtc_emu_noisy_rcp(double x) {
  unsigned long long bits;
  __builtin_memcpy(&bits, &x, 8);
  bits = (bits ^ (bits >> 29)) * 0x9E3779B97F4A7C15ull;
  const double e = ((double)(bits >> 11) * (1.0 / 9007199254740992.0) * 2.0 - 1.0) * 1.1920928955078125e-07;
  return (1.0 / x) * (1.0 + e);
}
#define RTC_APPROX_RCP(x) rtc_emu_noisy_rcp(x)
using std::fabs; using std::floor; using std::fmax; using std::fmin; using std::pow; using std::sqrt;

#ifndef RTC_EMU_SIMT
// ---- mode 1: lanes run one after another; a "wave" is one lane (build the persistent kernel with RTC_WAVE = 1)
static uint3_ blockIdx, threadIdx, blockDim, gridDim;
template <class T> static inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
template <class T> static inline T atomicOr(T* p, T v) { T o = *p; *p = o | v; return o; }
static inline unsigned long long __ballot(int p) { return p ? 1ull : 0ull; }
static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline int __ffsll(long long x) { return __builtin_ffsll(x); }
template <class T> static inline T __shfl(T v, int) { return v; }
static inline void __syncthreads() {}  // sequential mode: only meaningful for one-thread blocks
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...)          \
  do {                                                                       \
    gridDim.x = (grid).x; blockDim.x = (block).x;                            \
    for (unsigned b_ = 0; b_ < (grid).x; b_++)                               \
      for (unsigned t_ = 0; t_ < (block).x; t_++) {                          \
        blockIdx.x = b_; threadIdx.x = t_;                                   \
        kernel(__VA_ARGS__);                                                 \
      }                                                                      \
  } while (0)
#else
// ---- mode 2 (SIMT): one std::thread per lane of a block, blocks one after another; __ballot / __shfl rendezvous
// through a spin barrier, so wave-cooperative code (votes, wave-aggregated atomics, refill) runs with real
// inter-lane interleavings under ASan/UBSan/TSan-free but data-race-visible conditions.  Blocks are single waves.
#include <atomic>
#include <thread>
#include <vector>
static thread_local uint3_ threadIdx, blockIdx;
static uint3_ blockDim, gridDim;
namespace emu_simt {
static std::atomic<int> g_arrived{0};
static std::atomic<int> g_generation{0};
static int g_lanes = 1;
static std::atomic<unsigned long long> g_bits[3];
static thread_local unsigned g_ballot_no = 0;
static unsigned long long g_shfl[64];
static inline void barrier() {
  int gen = g_generation.load(std::memory_order_acquire);
  if (g_arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == g_lanes) {
    g_arrived.store(0, std::memory_order_relaxed);
    g_generation.fetch_add(1, std::memory_order_acq_rel);
  } else {
    while (g_generation.load(std::memory_order_acquire) == gen) std::this_thread::yield();
  }
}
}  // namespace emu_simt
template <class T> static inline T atomicAdd(T* p, T v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
template <class T> static inline T atomicOr(T* p, T v) { return __atomic_fetch_or(p, v, __ATOMIC_SEQ_CST); }
static inline unsigned long long __ballot(int p) {
  using namespace emu_simt;
  unsigned k = g_ballot_no++;
  if (p) g_bits[k % 3].fetch_or(1ull << threadIdx.x, std::memory_order_acq_rel);
  barrier();
  unsigned long long r = g_bits[k % 3].load(std::memory_order_acquire);
  if (threadIdx.x == 0) g_bits[(k + 2) % 3].store(0ull, std::memory_order_release);  // the slot used by ballot k-1
  return r;
}
static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline int __ffsll(long long x) { return __builtin_ffsll(x); }
static inline void __syncthreads() { emu_simt::barrier(); }  // blocks are single waves here
template <class T> static inline T __shfl(T v, int src) {
  using namespace emu_simt;
  static_assert(sizeof(T) <= 8, "shfl payload");
  unsigned long long raw = 0;
  std::memcpy(&raw, &v, sizeof(T));
  g_shfl[threadIdx.x] = raw;
  barrier();
  raw = g_shfl[src];
  barrier();
  T out;
  std::memcpy(&out, &raw, sizeof(T));
  return out;
}
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...)                                   \
  do {                                                                                                \
    gridDim.x = (grid).x; blockDim.x = (block).x;                                                     \
    emu_simt::g_lanes = (int)(block).x;                                                               \
    for (unsigned b_ = 0; b_ < (grid).x; b_++) {                                                      \
      emu_simt::g_arrived = 0;                                                                        \
      for (auto& w_ : emu_simt::g_bits) w_ = 0ull;                                                    \
      std::vector<std::thread> lanes_;                                                                \
      for (unsigned t_ = 0; t_ < (block).x; t_++)                                                     \
        lanes_.emplace_back([=]() {                                                                   \
          blockIdx.x = b_; threadIdx.x = t_; emu_simt::g_ballot_no = 0;                               \
          kernel(__VA_ARGS__);                                                                        \
        });                                                                                           \
      for (auto& th_ : lanes_) th_.join();                                                            \
    }                                                                                                 \
  } while (0)
#endif
