// rtc_feat.hip — the ray kernels of ONE feature level (compiled once per level with -DRTC_FEAT=0..3, in parallel: each
// instantiation of the traversal is ~40 k instructions and most of the library's build time).  Feature levels: rtc_device.hpp,
// visit_prim.  Exports rtc_launch_trace_f<N> / rtc_launch_wf_ts_f<N> for the dispatchers in rtc_kernels.hip.
#include "rtc_device.hpp"

#ifndef RTC_FEAT
#error "compile with -DRTC_FEAT=0..3"
#endif
#ifndef RTC_CAT
#define RTC_CAT2(a, b) a##b
#define RTC_CAT(a, b) RTC_CAT2(a, b)
#endif

void RTC_CAT(rtc_launch_trace_f, RTC_FEAT)(bool count, unsigned grid, hipStream_t stream, const DScene& S, const DCamera& cam, const DPixelMap& pm, int fuel, double* rgb,
                                           double* hit_t, int* hit_prim, int* hit_k, DStats* stats) {
  if (count) hipLaunchKernelGGL((rtc_trace_kernel<true, RTC_FEAT>), dim3(grid), dim3(RTC_BLOCK), rtc_stack_bytes(S), stream, S, cam, pm, fuel, rgb, hit_t, hit_prim, hit_k, stats);
  else hipLaunchKernelGGL((rtc_trace_kernel<false, RTC_FEAT>), dim3(grid), dim3(RTC_BLOCK), rtc_stack_bytes(S), stream, S, cam, pm, fuel, rgb, hit_t, hit_prim, hit_k, stats);
}

void RTC_CAT(rtc_launch_wf_ts_f, RTC_FEAT)(bool count, unsigned grid, hipStream_t stream, const DScene& S, const DCamera& cam, const DPixelMap& pm, const DWave& W, int tl, int sl,
                                           unsigned n0, int slot, double* hit_t, int* hit_prim, int* hit_k, DStats* stats) {
  if (count) hipLaunchKernelGGL((wf_ts<true, RTC_FEAT>), dim3(grid), dim3(RTC_BLOCK), rtc_stack_bytes(S), stream, S, cam, pm, W, tl, sl, n0, slot, hit_t, hit_prim, hit_k, stats);
  else hipLaunchKernelGGL((wf_ts<false, RTC_FEAT>), dim3(grid), dim3(RTC_BLOCK), rtc_stack_bytes(S), stream, S, cam, pm, W, tl, sl, n0, slot, hit_t, hit_prim, hit_k, stats);
}
