// rtc_scene.cpp — the rtc.h C ABI: validates a flattened scene, builds the traversal program + accelerator,
// uploads SoA buffers to HBM and launches the trace kernels.  There is no CPU execution path in this file:
// every entry point that computes pixels needs a HIP device and fails with RTC_ERR_DEVICE otherwise.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rtc.h"
#include "device_scene.h"
#include "scene_build.hpp"

void rtc_launch_trace(const DScene& S, const DCamera& cam, const DPixelMap& pm, int fuel, double* rgb, double* hit_t, int* hit_prim, int* hit_k,
                      DStats* stats, bool count, hipStream_t stream, bool big_scene);
void rtc_launch_quantize(const double* rgb, unsigned char* out, unsigned long long n, hipStream_t stream);
void rtc_launch_pack_hits(const double* t, const int* prim, const int* k, DHit* out, unsigned long long n, hipStream_t stream);
void rtc_launch_deinterleave(const double* slab, double* image, unsigned rowlen, unsigned vsize, unsigned n, unsigned max_rows, unsigned band, hipStream_t stream);
void rtc_launch_deinterleave8(const unsigned char* slab, unsigned char* image, unsigned rowlen, unsigned vsize, unsigned n, unsigned max_rows, unsigned band, hipStream_t stream);
void rtc_launch_wavefront(const DScene& S, const DCamera& cam, const DPixelMap& pm, int fuel, const DWave& W, double* rgb, double* hit_t, int* hit_prim, int* hit_k,
                          DStats* stats, bool count, hipStream_t stream, unsigned blocks, unsigned shade_blocks);
uint64_t rtc_wavefront_work(const DCamera& cam, const DPixelMap& pm);
unsigned rtc_wavefront_grid(const DScene& S, int n_cu);
unsigned rtc_wavefront_lds_bytes(const DScene& S);
int32_t rtc_bvh_build_device(const std::vector<bvh::Item>& items, std::vector<DBvhNode>& n2, std::vector<uint32_t>& order, uint32_t base, int leaf_max, double* frame);

static thread_local std::string g_rtc_err;
static int rtc_fail(int code, const std::string& m) {
  g_rtc_err = m;
  return code;
}
#define HIP_OK(expr)                                                                                     \
  do {                                                                                                   \
    hipError_t e_ = (expr);                                                                              \
    if (e_ != hipSuccess) return rtc_fail(RTC_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

struct rtc_scene {
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  hipEvent_t marker[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  bool marker_recorded[8] = {false, false, false, false, false, false, false, false};
  std::vector<void*> allocs;
  uint64_t bytes = 0;
  DScene d{};
  DStats* d_stats = nullptr;
  // scratch output buffers (grown on demand)
  double* d_rgb = nullptr;
  double* d_hit_t = nullptr;
  int* d_hit_prim = nullptr;
  int* d_hit_k = nullptr;
  DHit* d_hits = nullptr;   // packed {t, prim, push} records for the copy to the host (rtc_hit layout)
  unsigned long long* d_digest = nullptr;    // hit-tree digests of a parity launch (per output slot)
  unsigned long long* d_wave_dig = nullptr;  //   and, wavefront path, the per-level hit hashes they are summed from
  uint64_t cap_digest = 0, cap_wave_dig = 0;
  uint8_t* d_rgb8 = nullptr;
  uint64_t cap_hits = 0, cap_rgb8 = 0;
  uint64_t* d_idx = nullptr;
  double* d_rays = nullptr;
  uint64_t cap_px = 0, cap_idx = 0, cap_rays = 0;
  int kernel_version = 0;  // 0: measured choice between the one-kernel (1) and the wavefront (4) path, per launch signature
  uint64_t tune_sig = 0;
  double tune_ms[2] = {-1.0, -1.0};
  int tune_n[2] = {0, 0};
  int tune_choice = 0;
  // what the scene is made of, for the first-launch path guess (pick_path)
  uint32_t n_analytic = 0;      // bounded analytic primitives (the world BVH's leaves)
  uint32_t n_bouncing = 0;      // primitives whose material reflects or refracts
  uint32_t n_prims_total = 0;
  bool last_wavefront = false;  // path of the most recent launch (for stats read back after an asynchronous launch)
  int last_fuel = 0;
  DCsgHit* csg_slab = nullptr;  // CSG subtrees beyond the per-lane buffer: csg_max_hits rows per thread of the largest launch so far
  uint64_t csg_slab_threads = 0;
  bool wave_alloc_failed = false;  // the device refused the queues once: launches stay on the one-kernel path
  // wavefront path (RTC_KERNEL=4): queues and per-level arrays, grown on demand
  DWave wave{};
  void* wave_mem = nullptr;
  uint64_t wave_cap = 0;
  int wave_levels = 0;
  unsigned wave_blocks = 0, shade_blocks = 0;
  unsigned wave_eighths = 9;  // queue capacity per level, in eighths of the launch's level-0 work ids
  int bvh_depth = 0;
  int built_on_device = 0;  // mesh accelerators of this scene whose tree came from bvh_device.hip
  uint32_t n_bvh_nodes = 0, n_mesh_tris = 0;

  template <class T>
  int upload(const std::vector<T>& v, const T** out) {
    *out = nullptr;
    if (v.empty()) return RTC_OK;
    void* p = nullptr;
    HIP_OK(hipMalloc(&p, v.size() * sizeof(T)));
    allocs.push_back(p);
    bytes += v.size() * sizeof(T);
    HIP_OK(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    *out = (const T*)p;
    return RTC_OK;
  }
};

namespace {

int ensure_px(rtc_scene* s, uint64_t n, bool hits) {
  if (n > s->cap_px) {
    (void)hipFree(s->d_rgb); (void)hipFree(s->d_hit_t); (void)hipFree(s->d_hit_prim); (void)hipFree(s->d_hit_k);  // hipFree(nullptr) is a no-op
    s->d_rgb = nullptr; s->d_hit_t = nullptr; s->d_hit_prim = nullptr; s->d_hit_k = nullptr; s->cap_px = 0;
    HIP_OK(hipMalloc((void**)&s->d_rgb, n * 3 * sizeof(double)));
    HIP_OK(hipMalloc((void**)&s->d_hit_t, n * sizeof(double)));
    HIP_OK(hipMalloc((void**)&s->d_hit_prim, n * sizeof(int)));
    HIP_OK(hipMalloc((void**)&s->d_hit_k, n * sizeof(int)));
    s->cap_px = n;
  }
  if (hits && n > s->cap_hits) {
    (void)hipFree(s->d_hits);
    s->d_hits = nullptr; s->cap_hits = 0;
    HIP_OK(hipMalloc((void**)&s->d_hits, n * sizeof(DHit)));
    s->cap_hits = n;
  }
  return RTC_OK;
}

// ---- device -> host for the caller's output buffers (Image::par_render returns host pixels, src/image.rs:76-80) ----------------
// Measured on the GPU box (scripts/d2h_probe.hip, profiles/r3_d2h_probe.txt): a pageable hipMemcpy runs at PCIe speed (52-56 GB/s,
// 1.0 ms per 50 MB) when the destination's pages exist and at 2.4 ms per 50 MB into a freshly allocated buffer (Vec / calloc /
// numpy.empty: the runtime pins the range, which faults its pages in in bulk).  What made round 2's rtc_render 13x slower than its
// kernels was everything around that copy: three temporary vectors and a scalar host loop for the hit records, and a host sync
// between the kernels and the first copy.  So: (1) hit records are packed on the device into the caller's layout; (2) one copy per
// output array, straight into the caller's memory, queued on the scene's stream behind the kernels.
// Touching the destination's pages from host threads while the device renders was measured too and is OFF by default
// (RTC_PRETOUCH_THREADS=n turns it on): 3.5 ms per 50 MB with 8 threads writing one byte per page (page faults of one address space
// serialise in the kernel) — and 22 ms when the page was read first (zero page mapped, then a second, copy-on-write fault).
void pretouch_pages(void* p, size_t bytes, std::vector<std::thread>* pool) {
  if (!p || bytes < (4u << 20)) return;
  static const int T = [] { const char* e = std::getenv("RTC_PRETOUCH_THREADS"); int t = e ? std::atoi(e) : 0; return t < 0 ? 0 : (t > 32 ? 32 : t); }();
  if (T == 0) return;
  char* base = (char*)p;
  const size_t per = ((bytes / (size_t)T) + 4095) & ~(size_t)4095;
  for (int t = 0; t < T; t++) {
    const size_t b = std::min(bytes, per * (size_t)t), e = std::min(bytes, per * (size_t)(t + 1));
    if (e > b) pool->emplace_back([base, b, e] { for (size_t o = b; o < e; o += 4096) { volatile char* q = base + o; *q = 0; } });
  }
}
void join_all(std::vector<std::thread>* pool) {
  for (auto& t : *pool) t.join();
  pool->clear();
}

// Sizes the wavefront arrays for `n_work` level-0 work ids and fuel + 1 levels: every level may hold up to wave_eighths / 8 x
// n_work rays (1.125 x to start with: a level larger than the frame's work ids needs most hits to spawn two rays).  A level that
// needs more sets the overflow flag; a synchronous launch then doubles the factor and renders again while the arrays stay
// under RTC_WF_MAX_BYTES (default 24 GiB), else falls back to the one-kernel path.
uint64_t wave_cap(const rtc_scene* s, uint64_t n_work, unsigned eighths) { (void)s; return std::max<uint64_t>((n_work * eighths + 7) / 8, 4096); }
uint64_t wave_bytes(uint64_t cap, int lv) { return dwave_bytes(cap, lv); }
uint64_t wave_budget() {
  const char* e = std::getenv("RTC_WF_MAX_BYTES");
  return e ? std::strtoull(e, nullptr, 10) : (24ull << 30);
}
int ensure_wave(rtc_scene* s, uint64_t n_work, int fuel) {
  const int levels = fuel + 1;
  uint64_t cap = wave_cap(s, n_work, s->wave_eighths);
  if (cap > 0x7fffff00ull || wave_bytes(cap, levels) > wave_budget()) return rtc_fail(RTC_ERR_UNSUPPORTED, "launch too large for the wavefront path");
  if (cap <= s->wave_cap && levels <= s->wave_levels) return RTC_OK;
  cap = std::max(cap, s->wave_cap);
  const int lv = std::max(levels, s->wave_levels);
  HIP_OK(hipStreamSynchronize(s->stream));
  if (s->wave_mem) (void)hipFree(s->wave_mem);
  s->wave_mem = nullptr; s->wave_cap = 0; s->wave_levels = 0;
  {
    // other owners of the device's memory (a host framework's allocator, other scenes) are not in the budget: ask the device
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && wave_bytes(cap, lv) > (uint64_t)free_b)
      return rtc_fail(RTC_ERR_UNSUPPORTED, "the wavefront queues do not fit the device's free memory");
    const hipError_t e = std::getenv("RTC_WF_FAIL_ALLOC") ? hipErrorOutOfMemory : hipMalloc(&s->wave_mem, wave_bytes(cap, lv));  // env: test hook
    if (e != hipSuccess) {
      (void)hipGetLastError();  // clear the sticky error: the launch continues on the one-kernel path
      s->wave_mem = nullptr;
      if (e == hipErrorOutOfMemory) return rtc_fail(RTC_ERR_UNSUPPORTED, "hipMalloc of the wavefront queues: out of memory");
      return rtc_fail(RTC_ERR_DEVICE, std::string("hipMalloc of the wavefront queues: ") + hipGetErrorString(e));
    }
  }
  dwave_carve(&s->wave, s->wave_mem, cap, lv);
  s->wave_cap = cap;
  s->wave_levels = lv;
  return RTC_OK;
}

void to_dcam(const rtc_camera& c, DCamera* d) {
  d->hsize = c.hsize; d->vsize = c.vsize;
  d->half_width = c.half_width; d->half_height = c.half_height; d->pixel_size = c.pixel_size;
  std::memcpy(d->inv, c.transform_inv, sizeof(d->inv));
}

// Signature of a launch for the path choice below: everything the ray counts of a frame depend on besides the scene.
uint64_t launch_signature(const DCamera& cam, const DPixelMap& pm, int fuel) {
  uint64_t h = 1469598103934665603ull;
  auto mix = [&](const void* p, size_t n) { for (size_t i = 0; i < n; i++) { h ^= ((const unsigned char*)p)[i]; h *= 1099511628211ull; } };
  mix(&cam, sizeof(cam));
  mix(&pm.n, sizeof(pm.n)); mix(&pm.mode, sizeof(pm.mode)); mix(&pm.row_first, sizeof(pm.row_first)); mix(&pm.row_step, sizeof(pm.row_step)); mix(&pm.band, sizeof(pm.band));
  mix(&fuel, sizeof(fuel));
  return h ? h : 1;
}

// Which path renders a launch (RTC_KERNEL unset).  The one-kernel path wins where rays are cheap (planes + one gated mesh:
// the teapot scenes, 0.85 vs 1.5 ms), the wavefront path where a ray walks a BVH of analytic primitives and the ray trees
// are deep (config 2: 2.3 vs 4.2 ms); which one depends on scene, camera and fuel, so the choice is measured where it can be:
// for a given launch signature the first four SYNCHRONOUS launches alternate between the paths, starting with the guess below (the
// smaller of a path's two times counts), and every later launch — synchronous or not — takes the faster.  Until a signature is
// measured, and for a caller that renders one frame per scene (the reference's only call pattern, src/bin/*.rs), the guess decides:
// first_guess() — wavefront iff the scene has an analytic BVH worth walking (>= 32 bounded analytic primitives), a tenth of its
// primitives reflect or refract, the frame is large enough to fill per-level launches and fuel allows bounces; else one kernel,
// which also never allocates ray queues.  (Calibrated on the nine reference scenes and the five BASELINE configs:
// profiles/r3_path_choice.txt.)  A wavefront launch whose queues overflowed is rendered again by the one-kernel path and never chosen for
// that signature, so an unsynchronised wavefront launch only ever repeats a launch that is known to fit.
int first_guess(const rtc_scene* s, uint64_t n_work, int fuel) {
  if (const char* e = std::getenv("RTC_FIRST_GUESS")) { const int v = std::atoi(e); if (v == 1 || v == 4) return v; }
  const bool deep = fuel >= 2 && (uint64_t)s->n_bouncing * 10u >= (uint64_t)s->n_prims_total && s->n_bouncing > 0;
  return (s->n_analytic >= 32 && deep && n_work >= (256u << 10)) ? 4 : 1;
}
int pick_path(rtc_scene* s, uint64_t sig, uint64_t n_work, int fuel, bool will_sync, bool pixel_list) {
  if (sig != s->tune_sig) { s->tune_sig = sig; s->tune_ms[0] = s->tune_ms[1] = -1.0; s->tune_n[0] = s->tune_n[1] = 0; s->tune_choice = 0; }
  if (pixel_list) return 1;  // index lists and explicit rays: small, irregular launches
  if (s->tune_choice) return s->tune_choice;
  const int guess = first_guess(s, n_work, fuel);
  if (!will_sync) return guess;
  // two samples per path, alternating, the smaller one counts: a path's first launch pays for code loading and scratch
  const int other = guess == 4 ? 1 : 4;
  const int ng = s->tune_n[guess == 4 ? 1 : 0], no = s->tune_n[other == 4 ? 1 : 0];
  return ng <= no ? guess : other;
}

// Sticky error state of the scene (DStats tail): read and cleared together.
int read_clear_sticky(rtc_scene* s, DStats* h) {
  HIP_OK(hipMemcpy(h, s->d_stats, sizeof(*h), hipMemcpyDeviceToHost));
  if (h->nan_ts || h->guard || h->wf_overflow)
    HIP_OK(hipMemset((char*)s->d_stats + RTC_STATS_LAUNCH_BYTES, 0, sizeof(DStats) - RTC_STATS_LAUNCH_BYTES));
  return RTC_OK;
}

// rtc_stats of the scene's most recent launch from its device counters `h` (read after the stream went idle).
int fill_stats(rtc_scene* s, const DStats& h, uint64_t pixels, rtc_stats* stats) {
  float ms = 0.f;
  HIP_OK(hipEventElapsedTime(&ms, s->ev0, s->ev1));
  std::memset(stats, 0, sizeof(*stats));
  stats->pixels = pixels;
  stats->rays_primary = h.rays_primary; stats->rays_shadow = h.rays_shadow; stats->rays_reflect = h.rays_reflect; stats->rays_refract = h.rays_refract;
  stats->rays_container = h.rays_container; stats->accel_nodes = h.accel_nodes; stats->group_tests = h.group_tests; stats->tri_tests = h.tri_tests;
  stats->analytic_tests = h.analytic_tests; stats->nan_ts = h.nan_ts;
  stats->accel_nodes_kernarg = h.knodes; stats->analytic_tests_kernarg = h.kplanes; stats->light_grid_cells = h.light_cells;
  stats->group_tests_uniform = h.kgroups;
  stats->kernel_ms = ms;
  stats->n_launches = s->last_wavefront ? 2u * (uint32_t)s->last_fuel + 4u : 1u;
  return RTC_OK;
}

// `after` (optional): queued behind the kernels of EVERY attempt of this launch, before the host waits (the copies of a host-pixel
// render); a launch that is rendered again (queue overflow -> larger queues / the other path) calls it again.
typedef std::function<int()> AfterLaunch;
int run(rtc_scene* s, const DCamera& cam, DPixelMap pm, int fuel, double* d_rgb, bool want_hits, rtc_stats* stats, bool count, bool sync, int force = 0,
        const AfterLaunch* after = nullptr) {
  if (fuel < 0) fuel = 0;  // fuel <= 0 spawns nothing (src/world.rs:90,110)
  if (fuel > RTC_MAX_FUEL) return rtc_fail(RTC_ERR_UNSUPPORTED, "fuel exceeds RTC_MAX_FUEL (16): the reference recurses without a limit, the device keeps its pending rays per lane");
  // World::shade_hit calls reflected_color / refracted_color inside its loop over the lights (src/world.rs:58-79): a world without
  // lights traces no secondary ray at all (and shades every hit black)
  if (s->d.n_lights == 0) fuel = 0;
  HIP_OK(hipSetDevice(s->device));
  // only the per-launch counters are zeroed: the error fields behind them accumulate until somebody reads them
  HIP_OK(hipMemsetAsync(s->d_stats, 0, RTC_STATS_LAUNCH_BYTES, s->stream));
  HIP_OK(hipEventRecord(s->ev0, s->stream));
  const bool will_sync = sync || stats != nullptr;
  const bool tuned = force == 0 && s->kernel_version == 0;
  int path = force ? force : s->kernel_version;
  if (tuned) path = pick_path(s, launch_signature(cam, pm, fuel), rtc_wavefront_work(cam, pm), fuel, will_sync, pm.mode != 2);
  if (path == 4 && s->wave_alloc_failed && force == 0) path = 1;
  const bool wavefront = path == 4 && pm.n > 0;
  if (s->d.csg_max_hits > RTC_CSG_MAX_HITS) {
    // threads of the launch that walk CSG sub-programs: the one-kernel grid covers every work id, the wavefront traversal grid is persistent
    const uint64_t threads = wavefront ? (uint64_t)s->wave_blocks * 64ull : ((rtc_wavefront_work(cam, pm) + 63) / 64) * 64;
    if (threads > s->csg_slab_threads) {
      const uint64_t bytes = threads * (uint64_t)s->d.csg_max_hits * sizeof(DCsgHit);
      const char* e = std::getenv("RTC_CSG_MAX_BYTES");
      if (bytes > (e ? std::strtoull(e, nullptr, 10) : (16ull << 30)))
        return rtc_fail(RTC_ERR_UNSUPPORTED, "the CSG intersection slab of this launch exceeds RTC_CSG_MAX_BYTES: render fewer pixels per call");
      HIP_OK(hipStreamSynchronize(s->stream));
      (void)hipFree(s->csg_slab);
      s->csg_slab = nullptr; s->csg_slab_threads = 0;
      HIP_OK(hipMalloc((void**)&s->csg_slab, bytes));
      s->csg_slab_threads = threads;
    }
    s->d.csg_slab = s->csg_slab;
  }
  if (wavefront) {
    int rc = ensure_wave(s, rtc_wavefront_work(cam, pm), fuel);
    if (rc == RTC_ERR_UNSUPPORTED && force == 0) {
      // does not fit the memory budget / the device's free memory: this launch shape stays on the one-kernel path (same bits)
      if (tuned) { s->tune_ms[1] = 1e30; s->tune_n[1] = 2; s->tune_choice = 1; }
      else s->wave_alloc_failed = true;
      return run(s, cam, pm, fuel, d_rgb, want_hits, stats, count, sync, 1, after);
    }
    if (rc != RTC_OK) return rc;
    s->wave.dig = nullptr;
    if (pm.digest && count) {  // parity launch: (levels) x cap hit hashes beside the queues
      const uint64_t need = (uint64_t)s->wave_cap * (uint64_t)(fuel + 1);
      if (need > s->cap_wave_dig) {
        HIP_OK(hipStreamSynchronize(s->stream));
        (void)hipFree(s->d_wave_dig);
        s->d_wave_dig = nullptr; s->cap_wave_dig = 0;
        HIP_OK(hipMalloc((void**)&s->d_wave_dig, need * sizeof(unsigned long long)));
        s->cap_wave_dig = need;
      }
      s->wave.dig = s->d_wave_dig;
    }
    HIP_OK(hipMemsetAsync(s->wave.counts, 0, RTC_WF_COUNTS * sizeof(uint32_t), s->stream));
    HIP_OK(hipEventRecord(s->ev0, s->stream));
    rtc_launch_wavefront(s->d, cam, pm, fuel, s->wave, d_rgb, want_hits ? s->d_hit_t : nullptr, s->d_hit_prim, s->d_hit_k, s->d_stats, count, s->stream, s->wave_blocks, s->shade_blocks);
  } else {
    rtc_launch_trace(s->d, cam, pm, fuel, d_rgb, want_hits ? s->d_hit_t : nullptr, s->d_hit_prim, s->d_hit_k, s->d_stats, count, s->stream, s->bytes > (32ull << 20));
  }
  HIP_OK(hipGetLastError());
  HIP_OK(hipEventRecord(s->ev1, s->stream));
  s->last_wavefront = wavefront;
  s->last_fuel = fuel;
  if (after) { int rca = (*after)(); if (rca != RTC_OK) return rca; }
  if (!sync && !stats) return RTC_OK;
  HIP_OK(hipStreamSynchronize(s->stream));
  DStats h;
  int rcs = read_clear_sticky(s, &h);
  if (rcs != RTC_OK) return rcs;
  if (wavefront) {
    if (std::getenv("RTC_WF_DUMP_COUNTS")) {  // debug aid: rays and shade records per level of this frame
      uint32_t c[64];
      HIP_OK(hipMemcpy(c, s->wave.counts, sizeof(c), hipMemcpyDeviceToHost));
      std::fprintf(stderr, "[rtc-wf] level: rays / shade records:");
      for (int l = 0; l <= fuel; l++) std::fprintf(stderr, " %d: %u / %u;", l, l ? c[l] : (unsigned)rtc_wavefront_work(cam, pm), c[RTC_WF_SHADE_COUNT + l]);
      std::fprintf(stderr, "\n");
    }
    uint32_t overflow = 0;
    HIP_OK(hipMemcpy(&overflow, s->wave.counts + RTC_WF_OVERFLOW, sizeof(overflow), hipMemcpyDeviceToHost));
    if (overflow) {  // a level outgrew its queue: larger queues if they fit the budget, else the one-kernel path (always fits)
      const uint64_t n_work = rtc_wavefront_work(cam, pm);
      if (s->wave_eighths < 512 && wave_bytes(wave_cap(s, n_work, 2 * s->wave_eighths), fuel + 1) <= wave_budget() && wave_cap(s, n_work, 2 * s->wave_eighths) <= 0x7fffff00ull) {
        s->wave_eighths *= 2;
        const int rc2 = run(s, cam, pm, fuel, d_rgb, want_hits, stats, count, true, 4, after);
        if (rc2 != RTC_ERR_UNSUPPORTED) return rc2;  // (the larger queues were refused by the device: fall through)
      }
      if (tuned) { s->tune_ms[1] = 1e30; s->tune_n[1] = 2; s->tune_choice = 1; }
      return run(s, cam, pm, fuel, d_rgb, want_hits, stats, count, true, 1, after);
    }
  }
  if (tuned && !count && pm.mode == 2 && !s->tune_choice) {
    float ms = 0.f;
    HIP_OK(hipEventElapsedTime(&ms, s->ev0, s->ev1));
    const int k = wavefront ? 1 : 0;
    s->tune_ms[k] = s->tune_n[k] ? std::min(s->tune_ms[k], (double)ms) : (double)ms;
    s->tune_n[k]++;
    if (s->tune_n[0] >= 2 && s->tune_n[1] >= 2) s->tune_choice = s->tune_ms[1] < s->tune_ms[0] ? 4 : 1;
  }
  if (stats) {
    int rcf = fill_stats(s, h, pm.n, stats);
    if (rcf != RTC_OK) return rcf;
  }
  if (std::getenv("RTC_DIAG_DUMP")) {  // RTC_DIAG builds: raw region / utilisation counters for scripts/diag_report.py
    std::fprintf(stderr, "[rtc-diag]");
    for (int i = 0; i < 64; i++) std::fprintf(stderr, " %llu", (unsigned long long)h.diag[i]);
    std::fprintf(stderr, "\n");
  }
  if (h.guard) return rtc_fail(RTC_ERR_DEVICE, "traversal guard tripped (mask " + std::to_string(h.guard) + "): an index left its array; no pixel since the last check is trustworthy");
  if (h.nan_ts) return rtc_fail(RTC_ERR_NAN, "a NaN intersection t reached a sort of two or more intersections (the reference panics in Intersection::sort, src/intersection.rs:124)");
  return RTC_OK;
}

// The pixel map of a host-side launch over pixels first .. first + n - 1 or the n listed indices (uploaded to the scene's index buffer).
int make_pixel_map(rtc_scene* s, const rtc_camera* cam, const uint64_t* pixel_indices, uint64_t first, uint64_t n, DPixelMap* pm, std::vector<uint64_t>* range_idx) {
  const uint64_t total = cam->hsize * cam->vsize;
  *pm = DPixelMap{};
  pm->n = n;
  // A contiguous range is expressed through the two pixel maps the kernels are validated with on hardware: whole rows ->
  // interleaved-row map with step 1; anything else -> an explicit index list.
  if (!pixel_indices) {
    if (first % cam->hsize == 0 && n % cam->hsize == 0) {
      pm->mode = 2; pm->row_first = (uint32_t)(first / cam->hsize); pm->row_step = 1;
    } else {
      range_idx->resize(n);
      for (uint64_t i = 0; i < n; i++) (*range_idx)[i] = first + i;
      pixel_indices = range_idx->data();
    }
  }
  if (pixel_indices) {
    for (uint64_t i = 0; i < n; i++)
      if (pixel_indices[i] >= total) return rtc_fail(RTC_ERR_INVALID, "pixel index exceeds the image");
    if (n > s->cap_idx) {
      if (s->d_idx) (void)hipFree(s->d_idx);
      s->d_idx = nullptr; s->cap_idx = 0;
      HIP_OK(hipMalloc((void**)&s->d_idx, n * sizeof(uint64_t)));
      s->cap_idx = n;
    }
    HIP_OK(hipMemcpy(s->d_idx, pixel_indices, n * sizeof(uint64_t), hipMemcpyHostToDevice));
    pm->mode = 1; pm->indices = s->d_idx;
  }
  return RTC_OK;
}

// One launch whose results go to the caller's host buffers: rgb (n x 3 doubles) or rgb8 (n x 3 bytes, Color::clamp on the device),
// and optionally the primary-hit records.  The destination's pages are touched by host threads while the device renders, the
// copies are queued behind the kernels (see pretouch_pages above).
int render_to_host(rtc_scene* s, const DCamera& dc, const DPixelMap& pm, int fuel, double* rgb, uint8_t* rgb8, rtc_hit* hits, rtc_stats* stats) {
  static_assert(sizeof(DHit) == sizeof(rtc_hit) && offsetof(DHit, prim) == offsetof(rtc_hit, prim) && offsetof(DHit, k) == offsetof(rtc_hit, push_idx), "hit layout");
  const uint64_t n = pm.n;
  std::vector<std::thread> pool;
  bool touched = false;
  const AfterLaunch after = [&]() -> int {
    if (!touched) {  // (a launch that is rendered again finds the pages present)
      touched = true;
      if (rgb) pretouch_pages(rgb, n * 3 * sizeof(double), &pool);
      if (rgb8) pretouch_pages(rgb8, n * 3, &pool);
      if (hits) pretouch_pages(hits, n * sizeof(rtc_hit), &pool);
    }
    if (rgb8) rtc_launch_quantize(s->d_rgb, s->d_rgb8, n * 3, s->stream);
    if (hits) rtc_launch_pack_hits(s->d_hit_t, s->d_hit_prim, s->d_hit_k, s->d_hits, n, s->stream);
    hipError_t e = hipGetLastError();
    join_all(&pool);
    if (e == hipSuccess && rgb) e = hipMemcpyAsync(rgb, s->d_rgb, n * 3 * sizeof(double), hipMemcpyDeviceToHost, s->stream);
    if (e == hipSuccess && rgb8) e = hipMemcpyAsync(rgb8, s->d_rgb8, n * 3, hipMemcpyDeviceToHost, s->stream);
    if (e == hipSuccess && hits) e = hipMemcpyAsync(hits, s->d_hits, n * sizeof(rtc_hit), hipMemcpyDeviceToHost, s->stream);
    if (e != hipSuccess) return rtc_fail(RTC_ERR_DEVICE, std::string("copy to the host: ") + hipGetErrorString(e));
    return RTC_OK;
  };
  const int rc = run(s, dc, pm, fuel, s->d_rgb, hits != nullptr, stats, stats != nullptr, true, 0, &after);
  join_all(&pool);  // (an error before the hook ran its join)
  return rc;
}

}  // namespace

extern "C" {

const char* rtc_last_error(void) { return g_rtc_err.c_str(); }

int rtc_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int rtc_scene_create(const rtc_scene_desc* desc, int device, rtc_scene** out) {
  if (!desc || !out) return rtc_fail(RTC_ERR_INVALID, "NULL argument");
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return rtc_fail(RTC_ERR_DEVICE, "no HIP device available (this library has no CPU path)");
  if (device < 0 || device >= ndev) return rtc_fail(RTC_ERR_DEVICE, "device index out of range");
  HIP_OK(hipSetDevice(device));
  rtb::HostArrays H;
  std::string err;
  // Mesh accelerators of at least 100 000 triangles are built on the device (LBVH over extent-aware bisection keys, bvh_device.hip)
  // instead of by the host's binned SAH: same pixels (the accelerator is results-neutral), a quarter of the build time, frames
  // within 8 % of the SAH tree's (config 5: 11.8 vs 10.9 ms; profiles/r3_device_bvh_build.txt).  RTC_DEVICE_BVH=0: always the host
  // build; =1: the device build from 4 096 triangles up (RTC_DEVICE_BVH_MIN overrides either threshold).
  const char* dbe = std::getenv("RTC_DEVICE_BVH");
  const bool device_bvh = !(dbe && dbe[0] == '0');
  const size_t device_min = (dbe && dbe[0] == '1') ? 4096 : 100000;
  int rc = rtb::build_arrays(*desc, &H, &err, device_bvh ? rtc_bvh_build_device : nullptr, device_min);
  if (rc != RTC_OK) return rtc_fail(rc, err);
  const bool timing = std::getenv("RTC_TIMING") != nullptr;
  const auto t_up0 = std::chrono::steady_clock::now();

  // (every early return below releases what was created so far: streams, events, uploaded tables)
  struct SceneDeleter { void operator()(rtc_scene* p) const { rtc_scene_destroy(p); } };
  std::unique_ptr<rtc_scene, SceneDeleter> s(new rtc_scene());
  s->device = device;
  HIP_OK(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
  HIP_OK(hipEventCreate(&s->ev0));
  HIP_OK(hipEventCreate(&s->ev1));
  for (auto& m : s->marker) HIP_OK(hipEventCreate(&m));

  DScene& d = s->d;
#define UP(field)                                                       \
  do {                                                                  \
    int rc_ = s->upload(H.field, &d.field);                             \
    if (rc_ != RTC_OK) return rc_;                                      \
  } while (0)
  UP(ops); UP(group_box); UP(group_parent); UP(bvh); UP(mtri); UP(mtri_prim); {
    int rc_ = s->upload(H.items, &d.item_prim);
    if (rc_ != RTC_OK) return rc_;
    d.quirk_prim = d.item_prim;
    d.qitem = d.item_prim;
  }
  UP(qgrids); UP(qcell); UP(bvh_frame); UP(csg); UP(prims); UP(pisect); UP(xf_inv); UP(xf_matinv); UP(limits);
  UP(tri_geo); UP(tri_nrm); UP(mat); UP(mat_pattern); UP(pats); UP(lights);
#undef UP
  d.n_ops = (int32_t)H.ops.size();
  d.n_prims = (int32_t)H.prims.size();
  d.n_recs = (int32_t)H.pisect.size();
  d.n_lights = H.n_lights;
  d.all_cast_shadow = H.all_cast_shadow;
  {
    DScene hv = H.view();
    d.bvh_stack = hv.bvh_stack;
    d.has_mesh = hv.has_mesh;
    d.has_csg = hv.has_csg;
    d.has_groups = hv.has_groups;
    d.has_recs = hv.has_recs;
    d.all_plain = hv.all_plain;
    d.no_glass_mirror = hv.no_glass_mirror;
    d.backface_skip = hv.backface_skip;
    d.light_grid_first = hv.light_grid_first;
    d.quirk_reach2 = hv.quirk_reach2;
    std::memcpy(d.abvh_frame, hv.abvh_frame, sizeof(d.abvh_frame));
    d.light_grid_n = hv.light_grid_n; d.light_grid_cell_off = hv.light_grid_cell_off;
    d.csg_max_hits = hv.csg_max_hits;
    d.csg_slab = nullptr;
    d.n_kops = hv.n_kops; d.n_kplanes = hv.n_kplanes;
    std::memcpy(d.kops, hv.kops, sizeof(d.kops));
    std::memcpy(d.kplanes, hv.kplanes, sizeof(d.kplanes));
    std::memcpy(d.kaux, hv.kaux, sizeof(d.kaux));
    d.kqgrid = hv.kqgrid;
    d.n_bvh = hv.n_bvh; d.n_items = hv.n_items; d.n_mtri = hv.n_mtri; d.n_quirk = hv.n_quirk;
    d.n_qitem = hv.n_qitem; d.n_qcell = hv.n_qcell; d.n_groups = hv.n_groups; d.n_qgrids = hv.n_qgrids;
  }
  s->n_prims_total = desc->n_prims;
  for (uint32_t i = 0; i < desc->n_prims; i++) {
    const rtc_prim& P = desc->prims[i];
    if (P.geometry != RTC_PLANE && P.geometry != RTC_TRIANGLE && P.geometry != RTC_SMOOTH_TRIANGLE) s->n_analytic++;
    if (P.material >= 0 && (uint32_t)P.material < desc->n_materials) {
      const rtc_material& M = desc->materials[P.material];
      if (M.reflective != 0.0 || M.transparency != 0.0) s->n_bouncing++;
    }
  }
  s->bvh_depth = H.bvh_depth;
  s->built_on_device = H.built_on_device;
  s->n_bvh_nodes = (uint32_t)H.bvh.size();
  s->n_mesh_tris = (uint32_t)H.mtri_prim.size();
  if (std::getenv("RTC_VERIFY_UPLOAD")) {  // debug aid: read every table back and compare with the host copy
    auto verify = [&](const char* name, const void* dev, const void* host, size_t bytes) {
      if (!bytes) return true;
      std::vector<unsigned char> back(bytes);
      if (hipMemcpy(back.data(), dev, bytes, hipMemcpyDeviceToHost) != hipSuccess) return false;
      if (std::memcmp(back.data(), host, bytes) != 0) { std::fprintf(stderr, "[rtc] upload mismatch in %s (%zu bytes)\n", name, bytes); return false; }
      return true;
    };
    bool ok = verify("items", d.item_prim, H.items.data(), H.items.size() * 4);
    ok = verify("qcell", d.qcell, H.qcell.data(), H.qcell.size() * 4) && ok;
    ok = verify("ops", d.ops, H.ops.data(), H.ops.size() * sizeof(DOp)) && ok;
    std::fprintf(stderr, "[rtc] upload verify: %s; n_items=%zu n_qcell=%zu n_prims=%zu\n", ok ? "ok" : "MISMATCH", H.items.size(), H.qcell.size(), H.prims.size());
  }
  if (timing) std::fprintf(stderr, "[rtc-timing] %-28s %.3f s (%.1f MB)\n", "upload", std::chrono::duration<double>(std::chrono::steady_clock::now() - t_up0).count(), (double)s->bytes / 1e6);
  HIP_OK(hipMalloc((void**)&s->d_stats, sizeof(DStats)));
  HIP_OK(hipMemset(s->d_stats, 0, sizeof(DStats)));
  {
    // RTC_KERNEL pins a device path for A/B runs and for the parity tests: 1 = one kernel per frame, 4 = wavefront kernels (every
    // launch, pixel lists and explicit rays included); unset = measured choice per launch shape.
    const char* kv = std::getenv("RTC_KERNEL");
    const int kvi = kv ? std::atoi(kv) : 0;
    s->kernel_version = (kvi == 1 || kvi == 4) ? kvi : 0;
    // hipDeviceGetAttribute, not hipGetDeviceProperties: the property struct's layout differs between ROCm releases and
    // this library may run on the HIP runtime PyTorch loaded first.
    int n_cu = 0;
    HIP_OK(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, device));
    // traversal kernel: a persistent grid of one-wave blocks, as many as the chip holds; shading kernel: two 512-thread blocks per CU
    s->wave_blocks = rtc_wavefront_grid(s->d, n_cu);
    if (const char* w = std::getenv("RTC_WF_BLOCKS_PER_CU")) s->wave_blocks = (unsigned)std::max(1, n_cu * std::atoi(w));
    s->shade_blocks = (unsigned)std::max(1, n_cu * 2);
    if (const char* w = std::getenv("RTC_WF_SHADE_BLOCKS_PER_CU")) s->shade_blocks = (unsigned)std::max(1, n_cu * std::atoi(w));
  }
  *out = s.release();
  return RTC_OK;
}

void rtc_scene_destroy(rtc_scene* s) {
  if (!s) return;
  (void)hipSetDevice(s->device);
  if (s->stream) (void)hipStreamSynchronize(s->stream);
  for (void* p : s->allocs) (void)hipFree(p);
  if (s->d_stats) (void)hipFree(s->d_stats);
  (void)hipFree(s->d_rgb); (void)hipFree(s->d_hit_t); (void)hipFree(s->d_hit_prim); (void)hipFree(s->d_hit_k);
  (void)hipFree(s->d_hits); (void)hipFree(s->d_rgb8); (void)hipFree(s->d_digest); (void)hipFree(s->d_wave_dig);
  if (s->wave_mem) (void)hipFree(s->wave_mem);
  (void)hipFree(s->csg_slab);
  if (s->d_idx) (void)hipFree(s->d_idx);
  if (s->d_rays) (void)hipFree(s->d_rays);
  for (auto& m : s->marker) if (m) (void)hipEventDestroy(m);
  if (s->ev0) (void)hipEventDestroy(s->ev0);
  if (s->ev1) (void)hipEventDestroy(s->ev1);
  if (s->stream) (void)hipStreamDestroy(s->stream);
  delete s;
}

uint64_t rtc_scene_device_bytes(const rtc_scene* s) { return s ? s->bytes : 0; }

int rtc_render(rtc_scene* s, const rtc_camera* cam, int32_t fuel, const uint64_t* pixel_indices, uint64_t first, uint64_t n, double* rgb, rtc_hit* hits,
               rtc_stats* stats) {
  if (!s || !cam || (!rgb && n)) return rtc_fail(RTC_ERR_INVALID, "NULL argument");
  if (cam->hsize == 0 || cam->vsize == 0) return rtc_fail(RTC_ERR_INVALID, "empty camera");
  const uint64_t total = cam->hsize * cam->vsize;
  if (!pixel_indices && first + n > total) return rtc_fail(RTC_ERR_INVALID, "pixel range exceeds the image");
  if (n == 0) { if (stats) std::memset(stats, 0, sizeof(*stats)); return RTC_OK; }
  HIP_OK(hipSetDevice(s->device));
  int rc = ensure_px(s, n, hits != nullptr);
  if (rc != RTC_OK) return rc;
  DPixelMap pm{};
  std::vector<uint64_t> range_idx;
  rc = make_pixel_map(s, cam, pixel_indices, first, n, &pm, &range_idx);
  if (rc != RTC_OK) return rc;
  DCamera dc;
  to_dcam(*cam, &dc);
  return render_to_host(s, dc, pm, fuel, rgb, nullptr, hits, stats);
}

int rtc_render_hit_digest(rtc_scene* s, const rtc_camera* cam, int32_t fuel, const uint64_t* pixel_indices, uint64_t first, uint64_t n, uint64_t* digest) {
  if (!s || !cam || (!digest && n)) return rtc_fail(RTC_ERR_INVALID, "NULL argument");
  if (cam->hsize == 0 || cam->vsize == 0) return rtc_fail(RTC_ERR_INVALID, "empty camera");
  const uint64_t total = cam->hsize * cam->vsize;
  if (!pixel_indices && first + n > total) return rtc_fail(RTC_ERR_INVALID, "pixel range exceeds the image");
  if (n == 0) return RTC_OK;
  HIP_OK(hipSetDevice(s->device));
  int rc = ensure_px(s, n, false);
  if (rc != RTC_OK) return rc;
  if (n > s->cap_digest) {
    (void)hipFree(s->d_digest);
    s->d_digest = nullptr; s->cap_digest = 0;
    HIP_OK(hipMalloc((void**)&s->d_digest, n * sizeof(unsigned long long)));
    s->cap_digest = n;
  }
  DPixelMap pm{};
  pm.n = n;
  std::vector<uint64_t> range_idx;
  rc = make_pixel_map(s, cam, pixel_indices, first, n, &pm, &range_idx);
  if (rc != RTC_OK) return rc;
  pm.digest = s->d_digest;
  DCamera dc;
  to_dcam(*cam, &dc);
  rtc_stats st;
  rc = run(s, dc, pm, fuel, s->d_rgb, false, &st, true, true);   // the counting variants carry the digest code
  if (rc != RTC_OK) return rc;
  HIP_OK(hipMemcpy(digest, s->d_digest, n * sizeof(uint64_t), hipMemcpyDeviceToHost));
  return RTC_OK;
}

int rtc_render_rgb8(rtc_scene* s, const rtc_camera* cam, int32_t fuel, uint8_t* rgb8, rtc_stats* stats) {
  if (!s || !cam || !rgb8) return rtc_fail(RTC_ERR_INVALID, "NULL argument");
  if (cam->hsize == 0 || cam->vsize == 0) return rtc_fail(RTC_ERR_INVALID, "empty camera");
  HIP_OK(hipSetDevice(s->device));
  const uint64_t n = cam->hsize * cam->vsize;
  int rc = ensure_px(s, n, false);
  if (rc != RTC_OK) return rc;
  if (3 * n > s->cap_rgb8) {
    (void)hipFree(s->d_rgb8);
    s->d_rgb8 = nullptr; s->cap_rgb8 = 0;
    HIP_OK(hipMalloc((void**)&s->d_rgb8, 3 * n));
    s->cap_rgb8 = 3 * n;
  }
  DPixelMap pm{};
  pm.n = n; pm.mode = 2; pm.row_first = 0; pm.row_step = 1;
  DCamera dc;
  to_dcam(*cam, &dc);
  return render_to_host(s, dc, pm, fuel, nullptr, rgb8, nullptr, stats);
}

// Rows of the image owned by part `first` of `step` when bands of `band` rows are dealt out round-robin (the last band of the image
// may be short).
static uint64_t band_rows_owned(uint64_t vsize, uint32_t band, uint32_t first, uint32_t step) {
  const uint64_t n_bands = (vsize + band - 1) / band;
  if (first >= n_bands) return 0;
  const uint64_t mine = (n_bands - first + step - 1) / step;
  uint64_t rows = mine * band;
  const uint64_t last = first + (mine - 1) * (uint64_t)step;  // my last band; short only if it is the image's last
  if (last == n_bands - 1) rows -= n_bands * band - vsize;
  return rows;
}

int rtc_render_bands_device(rtc_scene* s, const rtc_camera* cam, int32_t fuel, uint32_t band_rows, uint32_t band_first, uint32_t band_step, uint32_t n_rows,
                            double* rgb_dev, rtc_stats* stats, int count_stats, int sync) {
  if (!s || !cam || (!rgb_dev && n_rows)) return rtc_fail(RTC_ERR_INVALID, "NULL argument");
  if (band_step == 0 || band_rows == 0) return rtc_fail(RTC_ERR_INVALID, "band_rows and band_step must be >= 1");
  if (n_rows) {
    // the launch covers the first n_rows rows of this part's dense tile: whole bands, then possibly part of one
    const uint64_t j = n_rows - 1, last_row = ((uint64_t)band_first + (j / band_rows) * band_step) * band_rows + j % band_rows;
    if (last_row >= cam->vsize) return rtc_fail(RTC_ERR_INVALID, "rows exceed the image");
  }
  DPixelMap pm{};
  pm.n = (uint64_t)n_rows * cam->hsize;
  pm.mode = 2; pm.row_first = band_first; pm.row_step = band_step; pm.band = band_rows;
  if (pm.n == 0) { if (stats) std::memset(stats, 0, sizeof(*stats)); return RTC_OK; }
  DCamera dc;
  to_dcam(*cam, &dc);
  return run(s, dc, pm, fuel, rgb_dev, false, stats, count_stats != 0, sync != 0);
}

int rtc_render_rows_device(rtc_scene* s, const rtc_camera* cam, int32_t fuel, uint32_t row_first, uint32_t row_step, uint32_t n_rows, double* rgb_dev,
                           rtc_stats* stats, int count_stats, int sync) {
  return rtc_render_bands_device(s, cam, fuel, 1, row_first, row_step, n_rows, rgb_dev, stats, count_stats, sync);
}

uint64_t rtc_band_rows_owned(uint64_t vsize, uint32_t band_rows, uint32_t band_first, uint32_t band_step) {
  if (band_rows == 0 || band_step == 0) return 0;
  return band_rows_owned(vsize, band_rows, band_first, band_step);
}

int rtc_trace_rays(rtc_scene* s, const double* rays, uint64_t n, int32_t fuel, double* rgb, rtc_hit* hits, rtc_stats* stats) {
  if (!s || (!rays && n) || (!rgb && n)) return rtc_fail(RTC_ERR_INVALID, "NULL argument");
  if (n == 0) return RTC_OK;
  HIP_OK(hipSetDevice(s->device));
  int rc = ensure_px(s, n, hits != nullptr);
  if (rc != RTC_OK) return rc;
  if (n > s->cap_rays) {
    if (s->d_rays) (void)hipFree(s->d_rays);
    s->d_rays = nullptr; s->cap_rays = 0;
    HIP_OK(hipMalloc((void**)&s->d_rays, n * 6 * sizeof(double)));
    s->cap_rays = n;
  }
  HIP_OK(hipMemcpyAsync(s->d_rays, rays, n * 6 * sizeof(double), hipMemcpyHostToDevice, s->stream));
  DPixelMap pm{};
  pm.n = n; pm.mode = 3; pm.rays = s->d_rays;
  DCamera dc{};
  dc.hsize = 1; dc.vsize = 1;
  return render_to_host(s, dc, pm, fuel, rgb, nullptr, hits, stats);
}


int rtc_quantize_device(rtc_scene* s, const double* rgb_dev, uint64_t n_values, uint8_t* out_dev, int sync) {
  if (!s || (n_values && (!rgb_dev || !out_dev))) return rtc_fail(RTC_ERR_INVALID, "NULL argument");
  HIP_OK(hipSetDevice(s->device));
  rtc_launch_quantize(rgb_dev, out_dev, n_values, s->stream);
  HIP_OK(hipGetLastError());
  if (sync) HIP_OK(hipStreamSynchronize(s->stream));
  return RTC_OK;
}

int rtc_quantize(rtc_scene* s, const double* rgb, uint64_t n_values, uint8_t* out) {
  if (!s || (n_values && (!rgb || !out))) return rtc_fail(RTC_ERR_INVALID, "NULL argument");
  if (n_values == 0) return RTC_OK;
  HIP_OK(hipSetDevice(s->device));
  double* d_in = nullptr;
  uint8_t* d_out = nullptr;
  HIP_OK(hipMalloc((void**)&d_in, n_values * sizeof(double)));
  if (hipMalloc((void**)&d_out, n_values) != hipSuccess) { (void)hipFree(d_in); return rtc_fail(RTC_ERR_DEVICE, "hipMalloc failed"); }
  int rc = RTC_OK;
  if (hipMemcpy(d_in, rgb, n_values * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) rc = rtc_fail(RTC_ERR_DEVICE, "H2D copy failed");
  if (rc == RTC_OK) rc = rtc_quantize_device(s, d_in, n_values, d_out, 1);
  if (rc == RTC_OK && hipMemcpy(out, d_out, n_values, hipMemcpyDeviceToHost) != hipSuccess) rc = rtc_fail(RTC_ERR_DEVICE, "D2H copy failed");
  (void)hipFree(d_in);
  (void)hipFree(d_out);
  return rc;
}

uint64_t rtc_ppm(uint64_t hsize, uint64_t vsize, const uint8_t* rgb8, char* out, uint64_t cap) {
  // two passes: size, then fill (src/image.rs:93-112)
  auto digits = [](unsigned v) { return v >= 100 ? 3u : (v >= 10 ? 2u : 1u); };
  std::string head = "P3\n" + std::to_string(hsize) + " " + std::to_string(vsize) + "\n255";
  uint64_t size = head.size() + 1;  // + trailing newline
  const uint64_t n = hsize * vsize;
  for (uint64_t i = 0; i < n; i++) size += 1 + digits(rgb8[3 * i]) + 1 + digits(rgb8[3 * i + 1]) + 1 + digits(rgb8[3 * i + 2]);
  if (!out || cap < size + 1) return size;
  char* p = out;
  std::memcpy(p, head.data(), head.size());
  p += head.size();
  auto put = [&](unsigned v) {
    if (v >= 100) *p++ = (char)('0' + v / 100);
    if (v >= 10) *p++ = (char)('0' + (v / 10) % 10);
    *p++ = (char)('0' + v % 10);
  };
  uint64_t j = 0;
  for (uint64_t i = 0; i < n; i++) {
    if (i % hsize == 0 || j % 5 == 0) { *p++ = '\n'; j = 1; } else { *p++ = ' '; j += 1; }
    put(rgb8[3 * i]); *p++ = ' '; put(rgb8[3 * i + 1]); *p++ = ' '; put(rgb8[3 * i + 2]);
  }
  *p++ = '\n';
  *p = 0;
  return (uint64_t)(p - out);
}

// Error state of every launch since the last check (or the last synchronous render, which reports and clears it too):
// asynchronous launches (sync == 0, stats == NULL) do not read it back themselves.
int rtc_scene_check(rtc_scene* s) {
  if (!s) return rtc_fail(RTC_ERR_INVALID, "NULL scene");
  HIP_OK(hipSetDevice(s->device));
  HIP_OK(hipStreamSynchronize(s->stream));
  DStats h;
  int rc = read_clear_sticky(s, &h);
  if (rc != RTC_OK) return rc;
  if (h.guard) return rtc_fail(RTC_ERR_DEVICE, "traversal guard tripped (mask " + std::to_string(h.guard) + ")");
  if (h.nan_ts) return rtc_fail(RTC_ERR_NAN, "a NaN intersection t reached a sort of two or more intersections (the reference panics in Intersection::sort, src/intersection.rs:124)");
  if (h.wf_overflow) return rtc_fail(RTC_ERR_UNSUPPORTED, "a wavefront ray queue overflowed in an unsynchronised launch: render this launch synchronously (falls back by itself)");
  return RTC_OK;
}

// Stream markers.  Every event is created with the scene and lives as long as it; a marker that was never recorded is an
// error, not an undefined wait (hipEventSynchronize / hipEventElapsedTime on a never-recorded event).
static int marker_slot(rtc_scene* s, int slot, bool need_recorded) {
  if (!s) return rtc_fail(RTC_ERR_INVALID, "NULL scene");
  if (slot < 0 || slot >= 8) return rtc_fail(RTC_ERR_INVALID, "marker slot out of range (0..7)");
  if (need_recorded && !s->marker_recorded[slot]) return rtc_fail(RTC_ERR_INVALID, "marker slot " + std::to_string(slot) + " was never recorded");
  return RTC_OK;
}
int rtc_scene_record(rtc_scene* s, int slot) {
  int rc = marker_slot(s, slot, false);
  if (rc != RTC_OK) return rc;
  HIP_OK(hipSetDevice(s->device));
  HIP_OK(hipEventRecord(s->marker[slot], s->stream));
  s->marker_recorded[slot] = true;
  return RTC_OK;
}
int rtc_scene_wait(rtc_scene* s, int slot) {
  int rc = marker_slot(s, slot, true);
  if (rc != RTC_OK) return rc;
  HIP_OK(hipSetDevice(s->device));
  HIP_OK(hipEventSynchronize(s->marker[slot]));
  return RTC_OK;
}
int rtc_scene_elapsed_ms(rtc_scene* s, int from, int to, double* ms) {
  int rc = marker_slot(s, from, true);
  if (rc == RTC_OK) rc = marker_slot(s, to, true);
  if (rc != RTC_OK) return rc;
  if (!ms) return rtc_fail(RTC_ERR_INVALID, "NULL argument");
  HIP_OK(hipSetDevice(s->device));
  HIP_OK(hipEventSynchronize(s->marker[to]));  // both must have completed for hipEventElapsedTime
  float f = 0.f;
  HIP_OK(hipEventElapsedTime(&f, s->marker[from], s->marker[to]));
  *ms = f;
  return RTC_OK;
}

int rtc_scene_sync(rtc_scene* s) {
  if (!s) return rtc_fail(RTC_ERR_INVALID, "NULL scene");
  HIP_OK(hipSetDevice(s->device));
  HIP_OK(hipStreamSynchronize(s->stream));
  return RTC_OK;
}

// ---- N GPUs of one process (include/rtc.h) ---------------------------------------------------------------------
struct rtc_multi {
  std::vector<rtc_scene*> scenes;   // one replica per listed device
  std::vector<double*> tiles;       // replica k's dense rows k, k + n, ... on ITS device
  std::vector<uint64_t> tile_cap;
  std::vector<hipEvent_t> done;     // replica k's tile is complete (recorded on its stream)
  std::vector<hipEvent_t> copied;   // the first device has pulled replica k's tile (recorded on the first replica's stream): the
  std::vector<char> copied_valid;   //   replica's next frame waits for it before it overwrites the tile
  double* slab = nullptr;           // first device: n x max_rows x hsize x 3
  double* image = nullptr;          // first device: vsize x hsize x 3
  uint64_t slab_cap = 0, image_cap = 0;
  // rtc_render_multi_rgb8: the same buffers for quantised pixels (a replica quantises its tile on its own device)
  std::vector<uint8_t*> tiles8;
  std::vector<uint64_t> tile8_cap;
  uint8_t* slab8 = nullptr;
  uint8_t* image8 = nullptr;
  uint64_t slab8_cap = 0, image8_cap = 0;
  uint32_t band = 8;                // rows per band of the partition (rtc_multi_set_band_rows)
};

namespace {
int multi_fail(rtc_multi* m, int rc) { (void)m; return rc; }

int render_multi(rtc_multi* m, const rtc_camera* cam, int32_t fuel, double* rgb_dev_out, double* rgb_host, rtc_stats* stats, bool sync, uint8_t* rgb8_host = nullptr) {
  const bool q8 = rgb8_host != nullptr;
  if (!m || !cam) return rtc_fail(RTC_ERR_INVALID, "NULL argument");
  if (cam->hsize == 0 || cam->vsize == 0) return rtc_fail(RTC_ERR_INVALID, "empty camera");
  const uint32_t n = (uint32_t)m->scenes.size();
  const uint64_t H = cam->hsize, V = cam->vsize, rowlen = H * 3;
  if (rowlen > 0xffffffffull || V > 0xffffffffull) return rtc_fail(RTC_ERR_INVALID, "image too large");
  const uint32_t B = m->band;
  const uint64_t max_rows = (((V + B - 1) / B + n - 1) / n) * B;
  rtc_scene* s0 = m->scenes[0];
  DCamera dc;
  to_dcam(*cam, &dc);
  // buffers: a tile per replica on its own device; slab and image on the first device
  for (uint32_t k = 0; k < n; k++) {
    rtc_scene* s = m->scenes[k];
    if (max_rows * rowlen > m->tile_cap[k]) {
      HIP_OK(hipSetDevice(s->device));
      HIP_OK(hipStreamSynchronize(s->stream));
      (void)hipFree(m->tiles[k]);
      m->tiles[k] = nullptr; m->tile_cap[k] = 0;
      HIP_OK(hipMalloc((void**)&m->tiles[k], max_rows * rowlen * sizeof(double)));
      m->tile_cap[k] = max_rows * rowlen;
    }
    if (q8 && max_rows * rowlen > m->tile8_cap[k]) {
      HIP_OK(hipSetDevice(s->device));
      HIP_OK(hipStreamSynchronize(s->stream));
      (void)hipFree(m->tiles8[k]);
      m->tiles8[k] = nullptr; m->tile8_cap[k] = 0;
      HIP_OK(hipMalloc((void**)&m->tiles8[k], max_rows * rowlen));
      m->tile8_cap[k] = max_rows * rowlen;
    }
  }
  HIP_OK(hipSetDevice(s0->device));
  if (q8) {
    if ((uint64_t)n * max_rows * rowlen > m->slab8_cap) {
      HIP_OK(hipStreamSynchronize(s0->stream));
      (void)hipFree(m->slab8);
      m->slab8 = nullptr; m->slab8_cap = 0;
      HIP_OK(hipMalloc((void**)&m->slab8, (uint64_t)n * max_rows * rowlen));
      m->slab8_cap = (uint64_t)n * max_rows * rowlen;
    }
    if (V * rowlen > m->image8_cap) {
      HIP_OK(hipStreamSynchronize(s0->stream));
      (void)hipFree(m->image8);
      m->image8 = nullptr; m->image8_cap = 0;
      HIP_OK(hipMalloc((void**)&m->image8, V * rowlen));
      m->image8_cap = V * rowlen;
    }
  }
  if (!q8 && (uint64_t)n * max_rows * rowlen > m->slab_cap) {
    HIP_OK(hipStreamSynchronize(s0->stream));
    (void)hipFree(m->slab);
    m->slab = nullptr; m->slab_cap = 0;
    HIP_OK(hipMalloc((void**)&m->slab, (uint64_t)n * max_rows * rowlen * sizeof(double)));
    m->slab_cap = (uint64_t)n * max_rows * rowlen;
  }
  if (!q8 && !rgb_dev_out && V * rowlen > m->image_cap) {
    HIP_OK(hipStreamSynchronize(s0->stream));
    (void)hipFree(m->image);
    m->image = nullptr; m->image_cap = 0;
    HIP_OK(hipMalloc((void**)&m->image, V * rowlen * sizeof(double)));
    m->image_cap = V * rowlen;
  }
  double* image = rgb_dev_out ? rgb_dev_out : m->image;
  if (stats) std::memset(stats, 0, sizeof(*stats));
  // every replica traces its interleaved rows on its own device and stream; nothing is exchanged while tracing
  for (uint32_t k = 0; k < n; k++) {
    rtc_scene* s = m->scenes[k];
    DPixelMap pm{};
    const uint64_t rows = band_rows_owned(V, B, k, n);
    pm.n = rows * H; pm.mode = 2; pm.row_first = k; pm.row_step = n; pm.band = B;
    if (pm.n == 0) continue;
    if (m->copied_valid[k]) {  // frames queued back to back: the previous frame's gather still reads this tile
      HIP_OK(hipSetDevice(s->device));
      HIP_OK(hipStreamWaitEvent(s->stream, m->copied[k], 0));
    }
    // asynchronous on the replica's own stream (the counting variant when stats are wanted: they are read back after the gather)
    int rc = run(s, dc, pm, fuel, m->tiles[k], false, nullptr, stats != nullptr, false);
    if (rc != RTC_OK) return rc;
    HIP_OK(hipSetDevice(s->device));
    if (q8) {  // Color::clamp on the replica's own device, behind its trace kernels
      rtc_launch_quantize(m->tiles[k], m->tiles8[k], rows * rowlen, s->stream);
      HIP_OK(hipGetLastError());
    }
    HIP_OK(hipEventRecord(m->done[k], s->stream));
  }
  // gather: the first device's stream waits for each tile and pulls it over xGMI into its slot of the slab, then one
  // de-interleave pass writes the image (row k + n j  <-  slab[k][j])
  HIP_OK(hipSetDevice(s0->device));
  for (uint32_t k = 0; k < n; k++) {
    const uint64_t rows = band_rows_owned(V, B, k, n);
    if (rows == 0) continue;
    rtc_scene* s = m->scenes[k];
    HIP_OK(hipStreamWaitEvent(s0->stream, m->done[k], 0));
    void* dst = q8 ? (void*)(m->slab8 + (uint64_t)k * max_rows * rowlen) : (void*)(m->slab + (uint64_t)k * max_rows * rowlen);
    const void* src = q8 ? (const void*)m->tiles8[k] : (const void*)m->tiles[k];
    const uint64_t bytes = rows * rowlen * (q8 ? 1 : sizeof(double));
    if (s->device == s0->device) HIP_OK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s0->stream));
    else HIP_OK(hipMemcpyPeerAsync(dst, s0->device, src, s->device, bytes, s0->stream));
    HIP_OK(hipEventRecord(m->copied[k], s0->stream));
    m->copied_valid[k] = 1;
  }
  if (q8) rtc_launch_deinterleave8(m->slab8, m->image8, (unsigned)rowlen, (unsigned)V, n, (unsigned)max_rows, B, s0->stream);
  else rtc_launch_deinterleave(m->slab, image, (unsigned)rowlen, (unsigned)V, n, (unsigned)max_rows, B, s0->stream);
  HIP_OK(hipGetLastError());
  {  // the caller's pages, while the devices render (see pretouch_pages)
    std::vector<std::thread> pool;
    if (q8) pretouch_pages(rgb8_host, V * rowlen, &pool);
    if (rgb_host) pretouch_pages(rgb_host, V * rowlen * sizeof(double), &pool);
    join_all(&pool);
  }
  if (q8) HIP_OK(hipMemcpyAsync(rgb8_host, m->image8, V * rowlen, hipMemcpyDeviceToHost, s0->stream));
  if (rgb_host) HIP_OK(hipMemcpyAsync(rgb_host, image, V * rowlen * sizeof(double), hipMemcpyDeviceToHost, s0->stream));
  if (!sync && !rgb_host && !q8 && !stats) return RTC_OK;
  HIP_OK(hipStreamSynchronize(s0->stream));
  for (uint32_t k = 0; k < n; k++) {
    rtc_scene* s = m->scenes[k];
    const uint64_t rows = band_rows_owned(V, B, k, n);
    HIP_OK(hipSetDevice(s->device));
    HIP_OK(hipStreamSynchronize(s->stream));
    DStats h;
    int rc = read_clear_sticky(s, &h);   // counters of the replica's launch + error state of every launch since the last check
    if (rc != RTC_OK) return rc;
    if (h.guard) return rtc_fail(RTC_ERR_DEVICE, "traversal guard tripped (mask " + std::to_string(h.guard) + ")");
    if (h.nan_ts) return rtc_fail(RTC_ERR_NAN, "a NaN intersection t reached a sort of two or more intersections (the reference panics in Intersection::sort, src/intersection.rs:124)");
    if (h.wf_overflow) {  // an unsynchronised wavefront launch overflowed its queues: once more, synchronously (falls back by itself)
      DPixelMap pm{};
      pm.n = rows * H; pm.mode = 2; pm.row_first = k; pm.row_step = n; pm.band = B;
      rc = run(s, dc, pm, fuel, m->tiles[k], false, nullptr, false, true);
      if (rc != RTC_OK) return rc;
      return render_multi(m, cam, fuel, rgb_dev_out, rgb_host, stats, true, rgb8_host);
    }
    if (stats && rows) {
      rtc_stats st;
      rc = fill_stats(s, h, rows * H, &st);
      if (rc != RTC_OK) return rc;
      stats->pixels += st.pixels; stats->rays_primary += st.rays_primary; stats->rays_shadow += st.rays_shadow; stats->rays_reflect += st.rays_reflect;
      stats->rays_refract += st.rays_refract; stats->rays_container += st.rays_container; stats->accel_nodes += st.accel_nodes; stats->group_tests += st.group_tests;
      stats->tri_tests += st.tri_tests; stats->analytic_tests += st.analytic_tests; stats->nan_ts += st.nan_ts; stats->n_launches += st.n_launches;
      stats->accel_nodes_kernarg += st.accel_nodes_kernarg; stats->analytic_tests_kernarg += st.analytic_tests_kernarg;
      stats->light_grid_cells += st.light_grid_cells; stats->group_tests_uniform += st.group_tests_uniform;
      stats->kernel_ms = std::max(stats->kernel_ms, st.kernel_ms);  // replicas run side by side: the slowest one
    }
  }
  return RTC_OK;
}
}  // namespace

int rtc_multi_create(const rtc_scene_desc* desc, const int* devices, int n_devices, rtc_multi** out) {
  if (!desc || !devices || !out || n_devices <= 0) return rtc_fail(RTC_ERR_INVALID, "NULL argument / no devices");
  *out = nullptr;
  std::unique_ptr<rtc_multi> m(new rtc_multi());
  for (int k = 0; k < n_devices; k++) {
    rtc_scene* s = nullptr;
    int rc = rtc_scene_create(desc, devices[k], &s);
    if (rc != RTC_OK) { rtc_multi_destroy(m.release()); return rc; }
    m->scenes.push_back(s);
    m->tiles.push_back(nullptr);
    m->tile_cap.push_back(0);
    m->tiles8.push_back(nullptr);
    m->tile8_cap.push_back(0);
    hipEvent_t e = nullptr;
    if (hipSetDevice(devices[k]) != hipSuccess || hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
      rtc_multi_destroy(m.release());
      return rtc_fail(RTC_ERR_DEVICE, "hipEventCreate failed");
    }
    m->done.push_back(e);
    hipEvent_t c = nullptr;
    if (hipSetDevice(devices[0]) != hipSuccess || hipEventCreateWithFlags(&c, hipEventDisableTiming) != hipSuccess) {
      rtc_multi_destroy(m.release());
      return rtc_fail(RTC_ERR_DEVICE, "hipEventCreate failed");
    }
    m->copied.push_back(c);
    m->copied_valid.push_back(0);
  }
  // direct xGMI copies into the first device where the platform allows them (hipMemcpyPeerAsync stages through the host otherwise)
  (void)hipSetDevice(devices[0]);
  for (int k = 1; k < n_devices; k++) {
    int can = 0;
    if (devices[k] != devices[0] && hipDeviceCanAccessPeer(&can, devices[0], devices[k]) == hipSuccess && can) {
      if (hipDeviceEnablePeerAccess(devices[k], 0) != hipSuccess) (void)hipGetLastError();  // already enabled: fine
    }
  }
  *out = m.release();
  return RTC_OK;
}

void rtc_multi_destroy(rtc_multi* m) {
  if (!m) return;
  for (size_t k = 0; k < m->scenes.size(); k++) {
    rtc_scene* s = m->scenes[k];
    (void)hipSetDevice(s->device);
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    if (k < m->tiles.size()) (void)hipFree(m->tiles[k]);
    if (k < m->tiles8.size()) (void)hipFree(m->tiles8[k]);
    if (k < m->done.size() && m->done[k]) (void)hipEventDestroy(m->done[k]);
  }
  for (hipEvent_t c : m->copied) if (c) (void)hipEventDestroy(c);
  if (!m->scenes.empty()) {
    (void)hipSetDevice(m->scenes[0]->device);
    (void)hipFree(m->slab);
    (void)hipFree(m->image);
    (void)hipFree(m->slab8);
    (void)hipFree(m->image8);
  }
  for (rtc_scene* s : m->scenes) rtc_scene_destroy(s);
  delete m;
}

int rtc_multi_device_count(const rtc_multi* m) { return m ? (int)m->scenes.size() : 0; }

int rtc_multi_set_band_rows(rtc_multi* m, uint32_t band_rows) {
  if (!m || band_rows == 0) return rtc_fail(RTC_ERR_INVALID, "NULL argument / band_rows must be >= 1");
  for (rtc_scene* s : m->scenes) {  // queued frames still use the old layout of tiles and slab
    HIP_OK(hipSetDevice(s->device));
    HIP_OK(hipStreamSynchronize(s->stream));
  }
  m->band = band_rows;
  return RTC_OK;
}

int rtc_render_multi(rtc_multi* m, const rtc_camera* cam, int32_t fuel, double* rgb, rtc_stats* stats) {
  if (!rgb) return rtc_fail(RTC_ERR_INVALID, "NULL argument");
  return render_multi(m, cam, fuel, nullptr, rgb, stats, true);
}

int rtc_render_multi_rgb8(rtc_multi* m, const rtc_camera* cam, int32_t fuel, uint8_t* rgb8, rtc_stats* stats) {
  if (!rgb8) return rtc_fail(RTC_ERR_INVALID, "NULL argument");
  return render_multi(m, cam, fuel, nullptr, nullptr, stats, true, rgb8);
}

int rtc_render_multi_device(rtc_multi* m, const rtc_camera* cam, int32_t fuel, double* rgb_dev, int sync) {
  if (!rgb_dev) return rtc_fail(RTC_ERR_INVALID, "NULL argument");
  return render_multi(m, cam, fuel, rgb_dev, nullptr, nullptr, sync != 0);
}

int rtc_multi_sync(rtc_multi* m) {
  if (!m) return rtc_fail(RTC_ERR_INVALID, "NULL argument");
  for (rtc_scene* s : m->scenes) {
    int rc = rtc_scene_check(s);
    if (rc != RTC_OK) return rc;
  }
  return RTC_OK;
}

// Accelerator facts for reports (not part of the reference-facing surface).
void rtc_scene_path_info(const rtc_scene* s, int32_t* choice, double* one_kernel_ms, double* wavefront_ms) {
  if (choice) *choice = s ? (s->kernel_version ? s->kernel_version : s->tune_choice) : 0;
  if (one_kernel_ms) *one_kernel_ms = s ? s->tune_ms[0] : -1.0;
  if (wavefront_ms) *wavefront_ms = s ? s->tune_ms[1] : -1.0;
}

int rtc_scene_bvh_built_on_device(const rtc_scene* s) { return s ? s->built_on_device : 0; }

uint32_t rtc_scene_wavefront_lds_bytes(const rtc_scene* s) { return s ? rtc_wavefront_lds_bytes(s->d) : 0u; }

void rtc_scene_accel_info(const rtc_scene* s, uint32_t* n_ops, uint32_t* n_bvh_nodes, uint32_t* n_mesh_tris, uint32_t* bvh_depth) {
  if (n_ops) *n_ops = s ? (uint32_t)s->d.n_ops : 0;
  if (n_bvh_nodes) *n_bvh_nodes = s ? s->n_bvh_nodes : 0;
  if (n_mesh_tris) *n_mesh_tris = s ? s->n_mesh_tris : 0;
  if (bvh_depth) *bvh_depth = s ? (uint32_t)s->bvh_depth : 0;
}

}  // extern "C"
