// rtc_feat.hip — the ray kernels of ONE kernel variant (compiled once per variant with -DRTC_VARIANT=0..4, in parallel: each
// instantiation of the traversal is tens of thousands of instructions and most of the library's build time).
//   variant   feature level (rtc_device.hpp, visit_prim)          program
//   0         0: no gates                                          kernel arguments (DScene.kops)
//   1         1: whole meshes gated                                kernel arguments
//   2         1 (also serves gate-free programs too long for the kernel arguments)   memory (DScene.ops)
//   3         2: per-primitive gates                               memory
//   4         3: + CSG                                             memory
//   5         2: per-primitive gates                               kernel arguments (round 3: grouped scenes on the fast path — scalar op
//                                                                  fetches, LDS-resident tables, three waves per SIMD)
// Exports rtc_launch_trace_v<N> / rtc_launch_wf_ts_v<N> for the dispatchers in rtc_kernels.hip.
#if defined(RTC_VARIANT) && RTC_VARIANT == 5 && !defined(RTC_WF_TS_WAVES_MAXFEAT)
#define RTC_WF_TS_WAVES_MAXFEAT 2   // this variant's traversal kernel at three waves per SIMD like variants 0 and 1
#endif
#include "rtc_device.hpp"
#ifndef RTC_EMU
#include <atomic>
#endif

#ifndef RTC_VARIANT
#error "compile with -DRTC_VARIANT=0..4"
#endif
#ifndef RTC_CAT
#define RTC_CAT2(a, b) a##b
#define RTC_CAT(a, b) RTC_CAT2(a, b)
#endif
#undef RTC_V_FEAT
#undef RTC_V_KOPS
#if RTC_VARIANT == 0
#define RTC_V_FEAT 0
#define RTC_V_KOPS true
#elif RTC_VARIANT == 1
#define RTC_V_FEAT 1
#define RTC_V_KOPS true
#elif RTC_VARIANT == 2
#define RTC_V_FEAT 1
#define RTC_V_KOPS false
#elif RTC_VARIANT == 3
#define RTC_V_FEAT 2
#define RTC_V_KOPS false
#elif RTC_VARIANT == 4
#define RTC_V_FEAT 3
#define RTC_V_KOPS false
#else
#define RTC_V_FEAT 2
#define RTC_V_KOPS true
#endif
#undef RTC_V_LDS
#if RTC_VARIANT <= 1 || RTC_VARIANT == 5
#define RTC_V_LDS 1   // variants with a kernel-argument program can keep the scene tables in LDS
#else
#define RTC_V_LDS 0
#endif

void RTC_CAT(rtc_launch_trace_v, RTC_VARIANT)(bool count, int waves, unsigned grid, hipStream_t stream, const DScene& S, const DCamera& cam, const DPixelMap& pm, int fuel, double* rgb,
                                              double* hit_t, int* hit_prim, int* hit_k, DStats* stats) {
#if RTC_VARIANT == 1 || RTC_VARIANT == 2
  // mesh scenes larger than the L2s: the 3-waves-per-SIMD build (see rtc_trace_kernel)
  if (waves == 3 && !count) {
    hipLaunchKernelGGL((rtc_trace_kernel<false, RTC_V_FEAT, RTC_V_KOPS, 3>), dim3(grid), dim3(RTC_BLOCK), rtc_stack_bytes(S), stream, S, cam, pm, fuel, rgb, hit_t, hit_prim, hit_k, stats);
    return;
  }
#endif
  (void)waves;
#if RTC_VARIANT <= 2
  if (!count && S.all_plain && S.no_glass_mirror) {  // the lean build (see rtc_trace_kernel)
    hipLaunchKernelGGL((rtc_trace_kernel<false, RTC_V_FEAT, RTC_V_KOPS, 0, true>), dim3(grid), dim3(RTC_BLOCK), rtc_stack_bytes(S), stream, S, cam, pm, fuel, rgb, hit_t, hit_prim, hit_k, stats);
    return;
  }
#endif
  if (count) hipLaunchKernelGGL((rtc_trace_kernel<true, RTC_V_FEAT, RTC_V_KOPS>), dim3(grid), dim3(RTC_BLOCK), rtc_stack_bytes(S), stream, S, cam, pm, fuel, rgb, hit_t, hit_prim, hit_k, stats);
  else hipLaunchKernelGGL((rtc_trace_kernel<false, RTC_V_FEAT, RTC_V_KOPS>), dim3(grid), dim3(RTC_BLOCK), rtc_stack_bytes(S), stream, S, cam, pm, fuel, rgb, hit_t, hit_prim, hit_k, stats);
}

void RTC_CAT(rtc_launch_wf_ts_v, RTC_VARIANT)(bool count, unsigned grid, hipStream_t stream, const DScene& S, const DCamera& cam, const DPixelMap& pm, const DWave& W, int tl, int sl,
                                              unsigned n0, int slot, int fuel_left, double* hit_t, int* hit_prim, int* hit_k, DStats* stats) {
  if (count) hipLaunchKernelGGL((wf_ts<true, RTC_V_FEAT, RTC_V_KOPS>), dim3(grid), dim3(RTC_BLOCK), rtc_stack_bytes(S), stream, S, cam, pm, W, tl, sl, n0, slot, fuel_left, hit_t, hit_prim, hit_k, stats);
  else hipLaunchKernelGGL((wf_ts<false, RTC_V_FEAT, RTC_V_KOPS>), dim3(grid), dim3(RTC_BLOCK), rtc_stack_bytes(S), stream, S, cam, pm, W, tl, sl, n0, slot, fuel_left, hit_t, hit_prim, hit_k, stats);
}

#ifndef RTC_EMU
// The same kernel with the scene's accelerator nodes and intersection records copied into LDS by every block (variants with a
// kernel-argument program only: those are the small scenes); one block of RTC_LDS_BLOCK threads per CU, `lds_bytes` of dynamic LDS.
// Returns false when this device refuses the dynamic LDS size (nothing was launched: the caller takes the kernel that reads the tables from memory).
bool RTC_CAT(rtc_launch_wf_ts_lds_v, RTC_VARIANT)(bool count, unsigned grid, unsigned lds_bytes, hipStream_t stream, const DScene& S, const DCamera& cam, const DPixelMap& pm,
                                                  const DWave& W, int tl, int sl, unsigned n0, int slot, int fuel_left, double* hit_t, int* hit_prim, int* hit_k, DStats* stats) {
#if RTC_V_LDS
  // More than 64 KB of dynamic LDS has to be asked for, and the attribute belongs to the function object of the CURRENT device:
  // one bit per device (rtc_multi renders on several from one process), 1 = raised, in the second word 1 = refused.
  static std::atomic<unsigned long long> raised{0ull}, refused{0ull};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
  const unsigned long long bit = 1ull << dev;
  if (refused.load(std::memory_order_acquire) & bit) return false;
  if (!(raised.load(std::memory_order_acquire) & bit)) {
    // (static LDS of the kernel — the RTC_DIAG build has some — comes out of the same 160 KB)
    hipFuncAttributes fa;
    const int st = hipFuncGetAttributes(&fa, (const void*)wf_ts<false, RTC_V_FEAT, RTC_V_KOPS, true>) == hipSuccess ? (int)fa.sharedSizeBytes : 0;
    const hipError_t e1 = hipFuncSetAttribute((const void*)wf_ts<true, RTC_V_FEAT, RTC_V_KOPS, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - st);
    const hipError_t e2 = hipFuncSetAttribute((const void*)wf_ts<false, RTC_V_FEAT, RTC_V_KOPS, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - st);
    if (e1 != hipSuccess || e2 != hipSuccess) {
      (void)hipGetLastError();
      refused.fetch_or(bit, std::memory_order_acq_rel);
      return false;
    }
    raised.fetch_or(bit, std::memory_order_acq_rel);
  }
  if (count) hipLaunchKernelGGL((wf_ts<true, RTC_V_FEAT, RTC_V_KOPS, true>), dim3(grid), dim3(RTC_LDS_BLOCK), lds_bytes, stream, S, cam, pm, W, tl, sl, n0, slot, fuel_left, hit_t, hit_prim, hit_k, stats);
  else hipLaunchKernelGGL((wf_ts<false, RTC_V_FEAT, RTC_V_KOPS, true>), dim3(grid), dim3(RTC_LDS_BLOCK), lds_bytes, stream, S, cam, pm, W, tl, sl, n0, slot, fuel_left, hit_t, hit_prim, hit_k, stats);
  return true;
#else
  (void)count; (void)grid; (void)lds_bytes; (void)stream; (void)S; (void)cam; (void)pm; (void)W; (void)tl; (void)sl; (void)n0; (void)slot; (void)fuel_left;
  (void)hit_t; (void)hit_prim; (void)hit_k; (void)stats;
  return false;
#endif
}

// resident waves per CU of this variant's traversal kernel (the persistent grid of the wavefront path)
int RTC_CAT(rtc_wf_ts_blocks_per_cu_v, RTC_VARIANT)(unsigned lds_bytes) {
  int nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, wf_ts<false, RTC_V_FEAT, RTC_V_KOPS>, RTC_BLOCK, lds_bytes) != hipSuccess || nb <= 0) nb = 8;
  return nb;
}
#endif
