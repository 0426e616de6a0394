// tests/cpu_emu/emu_rtc.cpp — TEST INFRASTRUCTURE.
// Implements the rtc.h entry points by running the product's kernel source lane-by-lane on the CPU
// (see shim/hip/hip_runtime.h).  Linked with the product's rtw_capi.cpp into tests/cpu_emu/_build/librtc_emu.so
// so the same Python harness can drive it.  Used to debug kernel logic and to run sanitizers; it is never
// loaded by the raytracer_challenge_amd package and proves nothing about the GPU build by itself.
#include "shim/hip/hip_runtime.h"

#include "../../raytracer_challenge_amd/csrc/rtc_kernels.hip"

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <memory>
#include <string>
#include <vector>

#include "../../include/rtc.h"
#include "../../raytracer_challenge_amd/csrc/scene_build.hpp"

struct rtc_scene {
  rtb::HostArrays H;
  DScene d;
  double marker_ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // wall clock at rtc_scene_record (launches are synchronous here)
};
static thread_local std::string g_err;
static int efail(int c, const std::string& m) { g_err = m; return c; }

static int run(rtc_scene* s, const DCamera& cam, DPixelMap pm, int fuel, double* rgb, rtc_hit* hits, rtc_stats* stats, unsigned long long* digest = nullptr) {
  pm.digest = digest;
  if (fuel < 0) fuel = 0;
  if (fuel > RTC_MAX_FUEL) return efail(RTC_ERR_UNSUPPORTED, "fuel exceeds RTC_MAX_FUEL");
  if (s->d.n_lights == 0) fuel = 0;  // as rtc_scene.cpp run(): no lights, no secondary rays (src/world.rs:58-79)
  std::vector<double> t(hits ? pm.n : 0);
  std::vector<int> p(hits ? pm.n : 0), k(hits ? pm.n : 0);
  DStats st;
  std::memset(&st, 0, sizeof(st));
  unsigned n_launches = 1;
  std::vector<DCsgHit> slab;
  if (s->d.csg_max_hits > RTC_CSG_MAX_HITS) {  // as the product: rows per thread of the largest grid this call launches
    const uint64_t threads = std::max<uint64_t>(((rtc_wavefront_work(cam, pm) + 63) / 64) * 64, 5 * 64);
    slab.resize(threads * (uint64_t)s->d.csg_max_hits);
    s->d.csg_slab = slab.data();
  }
  const char* kv = std::getenv("RTC_KERNEL");
  if (kv && kv[0] == '4' && pm.n > 0) {  // wavefront path, host-allocated queues (grown on overflow, as the product does)
    const uint64_t n_work = rtc_wavefront_work(cam, pm);
    const int lv = fuel + 1;
    bool done = false;
    for (uint64_t eighths = 9; eighths <= 512 && !done; eighths *= 2) {
      const uint64_t cap = std::max<uint64_t>((n_work * eighths + 7) / 8, 256);
      std::vector<unsigned char> mem(dwave_bytes(cap, lv), 0xCD);  // hipMalloc does not zero either: nothing may rely on it
      DWave W{};
      dwave_carve(&W, mem.data(), cap, lv);
      std::vector<unsigned long long> dig(digest ? cap * (uint64_t)lv : 0, 0xCDCDCDCDCDCDCDCDull);
      if (digest) W.dig = dig.data();
      std::memset(W.counts, 0, RTC_WF_COUNTS * sizeof(uint32_t));
      std::memset(&st, 0, sizeof(st));
      rtc_launch_wavefront(s->d, cam, pm, fuel, W, rgb, hits ? t.data() : nullptr, p.data(), k.data(), &st, true, nullptr, 5, 3);
      done = W.counts[RTC_WF_OVERFLOW] == 0;
      if (done) n_launches = 2u * (unsigned)fuel + 4u;
    }
    if (!done) {  // as the product does beyond its memory budget: render again with the one-kernel path
      std::memset(&st, 0, sizeof(st));
      rtc_launch_trace(s->d, cam, pm, fuel, rgb, hits ? t.data() : nullptr, p.data(), k.data(), &st, true, nullptr, false);
    }
  } else {
    rtc_launch_trace(s->d, cam, pm, fuel, rgb, hits ? t.data() : nullptr, p.data(), k.data(), &st, true, nullptr, false);
  }
  if (hits) for (uint64_t i = 0; i < pm.n; i++) hits[i] = {t[i], p[i], k[i]};
  if (stats) {
    std::memset(stats, 0, sizeof(*stats));
    stats->pixels = pm.n;
    stats->rays_primary = st.rays_primary; stats->rays_shadow = st.rays_shadow; stats->rays_reflect = st.rays_reflect; stats->rays_refract = st.rays_refract;
    stats->rays_container = st.rays_container; stats->accel_nodes = st.accel_nodes; stats->group_tests = st.group_tests; stats->tri_tests = st.tri_tests;
    stats->analytic_tests = st.analytic_tests; stats->nan_ts = st.nan_ts;
    stats->accel_nodes_kernarg = st.knodes; stats->analytic_tests_kernarg = st.kplanes; stats->light_grid_cells = st.light_cells;
    stats->group_tests_uniform = st.kgroups;
    stats->n_launches = n_launches;
  }
  if (st.guard) return efail(RTC_ERR_DEVICE, "traversal guard tripped (mask " + std::to_string(st.guard) + ")");
  if (st.nan_ts) return efail(RTC_ERR_NAN, "NaN intersection t");
  return RTC_OK;
}
static void to_dcam(const rtc_camera& c, DCamera* d) {
  d->hsize = c.hsize; d->vsize = c.vsize; d->half_width = c.half_width; d->half_height = c.half_height; d->pixel_size = c.pixel_size;
  std::memcpy(d->inv, c.transform_inv, sizeof(d->inv));
}

extern "C" {

// test hooks: device functions of the kernel source, called directly (tests/test_group_gate.py)
int rtc_emu_group_box_hit(const double* box6, const double* ray6) {
  Ray r{ray6[0], ray6[1], ray6[2], ray6[3], ray6[4], ray6[5]};
  return group_box_hit(box6, r) ? 1 : 0;
}

const char* rtc_last_error(void) { return g_err.c_str(); }
int rtc_device_count(void) { return 0; }
int rtc_scene_create(const rtc_scene_desc* desc, int, rtc_scene** out) {
  std::unique_ptr<rtc_scene> s(new rtc_scene());
  std::string err;
  int rc = rtb::build_arrays(*desc, &s->H, &err);
  if (rc != RTC_OK) return efail(rc, err);
  s->d = s->H.view();
  *out = s.release();
  return RTC_OK;
}
void rtc_scene_destroy(rtc_scene* s) { delete s; }
uint64_t rtc_scene_device_bytes(const rtc_scene*) { return 0; }
int rtc_render(rtc_scene* s, const rtc_camera* cam, int32_t fuel, const uint64_t* idx, uint64_t first, uint64_t n, double* rgb, rtc_hit* hits, rtc_stats* stats) {
  DPixelMap pm{};
  pm.n = n;
  std::vector<uint64_t> range_idx;
  if (!idx) {
    if (first % cam->hsize == 0 && n % cam->hsize == 0) { pm.mode = 2; pm.row_first = (uint32_t)(first / cam->hsize); pm.row_step = 1; }
    else { range_idx.resize(n); for (uint64_t i = 0; i < n; i++) range_idx[i] = first + i; idx = range_idx.data(); }
  }
  if (idx) { pm.mode = 1; pm.indices = idx; }
  DCamera dc;
  to_dcam(*cam, &dc);
  return run(s, dc, pm, fuel, rgb, hits, stats);
}
int rtc_render_bands_device(rtc_scene* s, const rtc_camera* cam, int32_t fuel, uint32_t band_rows, uint32_t band_first, uint32_t band_step, uint32_t n_rows, double* rgb,
                            rtc_stats* stats, int, int) {
  if (band_rows == 0 || band_step == 0) return efail(RTC_ERR_INVALID, "band_rows and band_step must be >= 1");
  DPixelMap pm{};
  pm.n = (uint64_t)n_rows * cam->hsize; pm.mode = 2; pm.row_first = band_first; pm.row_step = band_step; pm.band = band_rows;
  DCamera dc;
  to_dcam(*cam, &dc);
  return run(s, dc, pm, fuel, rgb, nullptr, stats);
}
int rtc_render_rows_device(rtc_scene* s, const rtc_camera* cam, int32_t fuel, uint32_t row_first, uint32_t row_step, uint32_t n_rows, double* rgb, rtc_stats* stats, int a, int b) {
  return rtc_render_bands_device(s, cam, fuel, 1, row_first, row_step, n_rows, rgb, stats, a, b);
}
uint64_t rtc_band_rows_owned(uint64_t vsize, uint32_t band, uint32_t first, uint32_t step) {  // as the product (rtc_scene.cpp)
  if (band == 0 || step == 0) return 0;
  const uint64_t n_bands = (vsize + band - 1) / band;
  if (first >= n_bands) return 0;
  const uint64_t mine = (n_bands - first + step - 1) / step;
  uint64_t rows = mine * band;
  if (first + (mine - 1) * (uint64_t)step == n_bands - 1) rows -= n_bands * band - vsize;
  return rows;
}
int rtc_render_hit_digest(rtc_scene* s, const rtc_camera* cam, int32_t fuel, const uint64_t* idx, uint64_t first, uint64_t n, uint64_t* digest) {
  DPixelMap pm{};
  pm.n = n;
  std::vector<uint64_t> range_idx;
  if (!idx) {
    if (first % cam->hsize == 0 && n % cam->hsize == 0) { pm.mode = 2; pm.row_first = (uint32_t)(first / cam->hsize); pm.row_step = 1; }
    else { range_idx.resize(n); for (uint64_t i = 0; i < n; i++) range_idx[i] = first + i; idx = range_idx.data(); }
  }
  if (idx) { pm.mode = 1; pm.indices = idx; }
  DCamera dc;
  to_dcam(*cam, &dc);
  std::vector<double> rgb(3 * n);
  static_assert(sizeof(unsigned long long) == sizeof(uint64_t), "digest type");
  return run(s, dc, pm, fuel, rgb.data(), nullptr, nullptr, (unsigned long long*)digest);
}
int rtc_quantize_device(rtc_scene*, const double* rgb, uint64_t n, uint8_t* out, int);
int rtc_render_rgb8(rtc_scene* s, const rtc_camera* cam, int32_t fuel, uint8_t* rgb8, rtc_stats* stats) {
  if (!s || !cam || !rgb8) return efail(RTC_ERR_INVALID, "NULL argument");
  const uint64_t n = cam->hsize * cam->vsize;
  std::vector<double> rgb(3 * n);
  int rc = rtc_render(s, cam, fuel, nullptr, 0, n, rgb.data(), nullptr, stats);
  if (rc != RTC_OK) return rc;
  return rtc_quantize_device(s, rgb.data(), 3 * n, rgb8, 1);
}
int rtc_trace_rays(rtc_scene* s, const double* rays, uint64_t n, int32_t fuel, double* rgb, rtc_hit* hits, rtc_stats* stats) {
  DPixelMap pm{};
  pm.n = n; pm.mode = 3; pm.rays = rays;
  DCamera dc{};
  dc.hsize = 1; dc.vsize = 1;
  return run(s, dc, pm, fuel, rgb, hits, stats);
}
int rtc_quantize_device(rtc_scene*, const double* rgb, uint64_t n, uint8_t* out, int) { rtc_launch_quantize(rgb, out, n, nullptr); return RTC_OK; }
int rtc_quantize(rtc_scene* s, const double* rgb, uint64_t n, uint8_t* out) { return rtc_quantize_device(s, rgb, n, out, 1); }
int rtc_scene_sync(rtc_scene*) { return RTC_OK; }
int rtc_scene_check(rtc_scene*) { return RTC_OK; }
int rtc_scene_record(rtc_scene* s, int slot) {
  if (!s || slot < 0 || slot >= 8) return efail(RTC_ERR_INVALID, "bad marker slot");
  s->marker_ms[slot] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
  return RTC_OK;
}
int rtc_scene_wait(rtc_scene*, int) { return RTC_OK; }
int rtc_scene_elapsed_ms(rtc_scene* s, int from, int to, double* ms) {
  if (!s || !ms || from < 0 || from >= 8 || to < 0 || to >= 8) return efail(RTC_ERR_INVALID, "bad marker slot");
  *ms = s->marker_ms[to] - s->marker_ms[from];
  return RTC_OK;
}
void rtc_scene_path_info(const rtc_scene*, int32_t* choice, double* one_kernel_ms, double* wavefront_ms) {
  const char* kv = std::getenv("RTC_KERNEL");
  if (choice) *choice = kv ? std::atoi(kv) : 1;
  if (one_kernel_ms) *one_kernel_ms = -1.0;
  if (wavefront_ms) *wavefront_ms = -1.0;
}
uint32_t rtc_scene_wavefront_lds_bytes(const rtc_scene*) { return 0; }
int rtc_scene_bvh_built_on_device(const rtc_scene*) { return 0; }
void rtc_scene_accel_info(const rtc_scene* s, uint32_t* n_ops, uint32_t* n_bvh_nodes, uint32_t* n_mesh_tris, uint32_t* bvh_depth) {
  if (n_ops) *n_ops = (uint32_t)s->H.ops.size();
  if (n_bvh_nodes) *n_bvh_nodes = (uint32_t)s->H.bvh.size();
  if (n_mesh_tris) *n_mesh_tris = (uint32_t)s->H.mtri_prim.size();
  if (bvh_depth) *bvh_depth = (uint32_t)s->H.bvh_depth;
}
}
