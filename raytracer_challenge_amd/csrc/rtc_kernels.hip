// rtc_kernels.hip — hand-written HIP for gfx950 (MI355X): the kernels of the hot path that do not depend on the scene's feature
// level (wf_shade, wf_gather, the quantiser) and the host-callable launchers.  The ray kernels (rtc_trace_kernel, wf_ts) are
// templates in rtc_device.hpp, instantiated per feature level by rtc_feat.hip (one translation unit per level, built in parallel).
#include <cstdio>
#include <cstdlib>

#include "rtc_device.hpp"

#define RTC_VARIANT_DECL(N)                                                                                                                                   \
  void rtc_launch_trace_v##N(bool count, int waves, unsigned grid, hipStream_t stream, const DScene& S, const DCamera& cam, const DPixelMap& pm, int fuel, double* rgb, \
                             double* hit_t, int* hit_prim, int* hit_k, DStats* stats);                                                                       \
  void rtc_launch_wf_ts_v##N(bool count, unsigned grid, hipStream_t stream, const DScene& S, const DCamera& cam, const DPixelMap& pm, const DWave& W, int tl, \
                             int sl, unsigned n0, int slot, int fuel_left, double* hit_t, int* hit_prim, int* hit_k, DStats* stats);                         \
  int rtc_wf_ts_blocks_per_cu_v##N(unsigned lds_bytes);                                                                                                        \
  bool rtc_launch_wf_ts_lds_v##N(bool count, unsigned grid, unsigned lds_bytes, hipStream_t stream, const DScene& S, const DCamera& cam, const DPixelMap& pm,   \
                                 const DWave& W, int tl, int sl, unsigned n0, int slot, int fuel_left, double* hit_t, int* hit_prim, int* hit_k, DStats* stats);
RTC_VARIANT_DECL(0) RTC_VARIANT_DECL(1) RTC_VARIANT_DECL(2) RTC_VARIANT_DECL(3) RTC_VARIANT_DECL(4) RTC_VARIANT_DECL(5)
#undef RTC_VARIANT_DECL
#ifdef RTC_EMU
// the CPU emulator (tests/cpu_emu) compiles everything as one translation unit
#define RTC_VARIANT 0
#include "rtc_feat.hip"
#undef RTC_VARIANT
#define RTC_VARIANT 1
#include "rtc_feat.hip"
#undef RTC_VARIANT
#define RTC_VARIANT 2
#include "rtc_feat.hip"
#undef RTC_VARIANT
#define RTC_VARIANT 3
#include "rtc_feat.hip"
#undef RTC_VARIANT
#define RTC_VARIANT 4
#include "rtc_feat.hip"
#undef RTC_VARIANT
#define RTC_VARIANT 5
#include "rtc_feat.hip"
#undef RTC_VARIANT
#endif

// kernel variant a scene needs (rtc_feat.hip): its feature level (rtc_device.hpp, visit_prim) and where its program lives
static int rtc_variant(const DScene& S) {
  const int feat = S.has_csg ? 3 : (S.has_groups == 2 ? 2 : (S.has_groups ? 1 : 0));
  if (feat <= 1 && S.n_kops > 0) return feat;
  if (feat == 2 && S.n_kops > 0) return 5;
  return feat <= 1 ? 2 : feat + 1;
}

#ifndef RTC_WF_SHADE_WAVES
#define RTC_WF_SHADE_WAVES 4  // <= 128 VGPRs: two 512-thread blocks per CU, so one block's wait for its queue atomics is covered by the other
#endif
// PAT = false: every pattern of the scene is a Plain colour (DScene.all_plain): no pattern-tree walk and none of its 672 B of scratch per lane.
template <bool COUNT, bool PAT = true>
__global__ void __launch_bounds__(RTC_WF_SHADE_BLOCK, RTC_WF_SHADE_WAVES) wf_shade(DScene S, DCamera cam, DPixelMap pm, DWave W, int level, unsigned n0, int fuel0, DStats* __restrict__ stats) {
  // per wave and class: its count, then its base index in the queue (double-buffered by iteration parity: no barrier needed before
  // the next iteration writes).  Classes keep like with like inside a block's span of the queues, so that most 64-item chunks of the
  // next traversal launch hold one kind of ray: shade records on planes / on other primitives; reflected rays off planes (mirror
  // images of their coherent parents) / off other primitives / refracted rays.
  __shared__ unsigned s_rec2[2][2][16], s_child2[2][3][16];
  unsigned parity = 0;
  const WorkMap wm = make_workmap(pm, cam);
  const unsigned count = wf_count(W, level, n0);
  const size_t cap = W.cap;
  const double L = (double)S.n_lights;
  const int fuel = fuel0 - level;
  unsigned n_reflect = 0, n_refract = 0;
  int32_t* ch = W.child + (size_t)level * 2 * cap;
  double* nq = W.rq[(level + 1) & 1];
  const int lane = RTC_LANE_ID;
  const int wave = (int)(threadIdx.x / (RTC_WF_SHADE_BLOCK >= 64 ? 64 : 1));
  const int n_waves = RTC_WF_SHADE_BLOCK >= 64 ? RTC_WF_SHADE_BLOCK / 64 : 1;
  for (unsigned base = blockIdx.x * RTC_WF_SHADE_BLOCK; base < count; base += gridDim.x * RTC_WF_SHADE_BLOCK) {  // block-uniform bound: barriers inside
    const unsigned i = base + threadIdx.x;
    int prim = -1;
    if (i < count) prim = W.h_prim[i];
    const bool hit = prim >= 0;
    State st;
    double cr = 0.0, cg = 0.0, cbl = 0.0, weight = 1.0, n1 = 1.0, n2 = 1.0;
    int mat = 0, geom = 0;
    double reflective = 0.0, transparency = 0.0;
    if (hit) {
      Ray ray;
      if (level == 0) {
        uint64_t q = 0;
        (void)work_to_slot(wm, i, q);
        ray = slot_ray(pm, cam, q);
      } else {
        ray = wf_load_ray(W, level, i, weight);
      }
      const DPrim P = S.prims[prim];
      mat = P.mat;
      geom = P.geom;
      const double* M = S.mat + 8 * P.mat;
      reflective = M[4]; transparency = M[5];
      double hu, hv;
      hit_uv(S, P, ray, hu, hv);
      prepare_state(S, P, ray, W.h_t[i], hu, hv, st);
      if (transparency != 0.0 && fuel > 0) { n1 = W.h_n12[i]; n2 = W.h_n12[cap + i]; }  // stored under the same condition
      // Pattern::color_at(material_inv * over_point) — identical for every light (src/shape.rs:437)
      const double* mi = S.xf_matinv + 16 * P.xform;
      double x = mi[0] * st.px + mi[1] * st.py + mi[2] * st.pz + mi[3] * 1.0;
      double y = mi[4] * st.px + mi[5] * st.py + mi[6] * st.pz + mi[7] * 1.0;
      double z = mi[8] * st.px + mi[9] * st.py + mi[10] * st.pz + mi[11] * 1.0;
      double w = mi[12] * st.px + mi[13] * st.py + mi[14] * st.pz + mi[15] * 1.0;
      const DPat& root = S.pats[S.mat_pattern[P.mat]];
      if (!PAT || root.tag == 1) { cr = root.color[0]; cg = root.color[1]; cbl = root.color[2]; }
      else pattern_color(S, S.mat_pattern[P.mat], x, y, z, w, cr, cg, cbl);
    }
    const bool blend = hit && reflective > 0.0 && transparency > 0.0;
    double R = 0.0;
    if (blend) R = blend_reflectance(st, n1, n2, fuel, cr, cg, cbl);  // (a NaN reflectance: the record's colour becomes NaN)
    // reflected_color / refracted_color (src/world.rs:84-132), once per light in the reference -> factor L
    bool do_refl = false, do_refr = false;
    double wr = 0.0, wt = 0.0, tdx = 0.0, tdy = 0.0, tdz = 0.0;
    if (hit && fuel > 0) {
      do_refl = reflective != 0.0;
      do_refr = transparency != 0.0;
      wr = weight * L * reflective; wt = weight * L * transparency;
      if (blend) {
        wr *= R;
        wt *= (1.0 - R);
      }
      if (do_refr) {
        double n_ratio = n1 / n2;
        double cos_i = st.ex * st.nx + st.ey * st.ny + st.ez * st.nz;
        double sin2_t = (n_ratio * n_ratio) * (1.0 - cos_i * cos_i);
        if (sin2_t > 1.0) do_refr = false;
        else {
          double cos_t = sqrt(1.0 - sin2_t);
          double kk = n_ratio * cos_i - cos_t;
          tdx = st.nx * kk - st.ex * n_ratio; tdy = st.ny * kk - st.ey * n_ratio; tdz = st.nz * kk - st.ez * n_ratio;
        }
      }
    }
    // queue space: shade records and child rays (a wave's reflected rays first, then its refracted ones); one pair of
    // atomics per block and iteration
    const unsigned long long lt = (1ull << lane) - 1ull;
    const bool on_plane = hit && geom == 1;
    const unsigned long long m_rec0 = __ballot(hit && on_plane ? 1 : 0), m_rec1 = __ballot(hit && !on_plane ? 1 : 0);
    const unsigned long long m_c0 = __ballot(do_refl && on_plane ? 1 : 0), m_c1 = __ballot(do_refl && !on_plane ? 1 : 0), m_c2 = __ballot(do_refr ? 1 : 0);
    unsigned (*s_rec)[16] = s_rec2[parity];
    unsigned (*s_child)[16] = s_child2[parity];
    parity ^= 1u;
    if (lane == 0) {
      s_rec[0][wave] = (unsigned)__popcll(m_rec0); s_rec[1][wave] = (unsigned)__popcll(m_rec1);
      s_child[0][wave] = (unsigned)__popcll(m_c0); s_child[1][wave] = (unsigned)__popcll(m_c1); s_child[2][wave] = (unsigned)__popcll(m_c2);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned tr = 0, tc = 0;
      for (int w = 0; w < n_waves; w++) { tr += s_rec[0][w] + s_rec[1][w]; tc += s_child[0][w] + s_child[1][w] + s_child[2][w]; }
      unsigned br = tr ? atomicAdd(&W.counts[RTC_WF_SHADE_COUNT + level], tr) : 0u;
      unsigned bc = tc ? atomicAdd(&W.counts[level + 1], tc) : 0u;
      if ((unsigned long long)br + tr > W.cap || (unsigned long long)bc + tc > W.cap) { W.counts[RTC_WF_OVERFLOW] = 1u; stats->wf_overflow = 1ull; }
      for (int k = 0; k < 2; k++)
        for (int w = 0; w < n_waves; w++) { unsigned r = s_rec[k][w]; s_rec[k][w] = br; br += r; }
      for (int k = 0; k < 3; k++)
        for (int w = 0; w < n_waves; w++) { unsigned c = s_child[k][w]; s_child[k][w] = bc; bc += c; }
    }
    __syncthreads();
    const unsigned s = on_plane ? s_rec[0][wave] + (unsigned)__popcll(m_rec0 & lt) : s_rec[1][wave] + (unsigned)__popcll(m_rec1 & lt);
    const unsigned jr = on_plane ? s_child[0][wave] + (unsigned)__popcll(m_c0 & lt) : s_child[1][wave] + (unsigned)__popcll(m_c1 & lt);
    const unsigned jt = s_child[2][wave] + (unsigned)__popcll(m_c2 & lt);
    if (hit && s < W.cap) {
      double* r = W.sr;
      r[s] = st.px; r[cap + s] = st.py; r[2 * cap + s] = st.pz;
      r[3 * cap + s] = st.nx; r[4 * cap + s] = st.ny; r[5 * cap + s] = st.nz;
      r[6 * cap + s] = cr; r[7 * cap + s] = cg; r[8 * cap + s] = cbl;
      W.sr_mat[s] = mat;
      W.sr_node[s] = (int32_t)i;
    }
    if (do_refl && jr < W.cap) {
      nq[jr] = st.px; nq[cap + jr] = st.py; nq[2 * cap + jr] = st.pz; nq[3 * cap + jr] = st.rx; nq[4 * cap + jr] = st.ry; nq[5 * cap + jr] = st.rz;
      nq[6 * cap + jr] = wr;
      ch[i] = (int32_t)jr;
      n_reflect++;
    }
    if (do_refr && jt < W.cap) {
      nq[jt] = st.ux; nq[cap + jt] = st.uy; nq[2 * cap + jt] = st.uz; nq[3 * cap + jt] = tdx; nq[4 * cap + jt] = tdy; nq[5 * cap + jt] = tdz;
      nq[6 * cap + jt] = wt;
      ch[cap + i] = (int32_t)jt;
      n_refract++;
    }
  }
  if (COUNT) {
    atomicAdd(&stats->rays_reflect, (unsigned long long)n_reflect);
    atomicAdd(&stats->rays_refract, (unsigned long long)n_refract);
  }
}

// Pixel = the contributions of its ray tree added in the order the one-kernel path adds them (a ray, then its reflected
// subtree, then its refracted subtree: rtc_trace_kernel's depth-first loop), so both paths produce the same bits and a
// pixel's value does not depend on what else was rendered with it.  One lane per level-0 work id walks the child links.
__global__ void __launch_bounds__(256) wf_gather(DCamera cam, DPixelMap pm, DWave W, unsigned n0, double* __restrict__ rgb) {
  const WorkMap wm = make_workmap(pm, cam);
  const size_t cap = W.cap;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n0; i += gridDim.x * blockDim.x) {
    uint64_t q;
    if (!work_to_slot(wm, i, q)) continue;
    double r = 0.0, g = 0.0, b = 0.0;
    int wait_idx[RTC_MAX_FUEL + 1];  // refracted children waiting for their turn; entry k belongs to level wait_lvl[k]
    int wait_lvl[RTC_MAX_FUEL + 1];
    int sp = 0, lvl = 0, idx = (int)i;
    const bool digest = W.dig != nullptr && pm.digest != nullptr;  // parity channel: the same walk sums the rays' hit hashes
    unsigned long long dg = 0ull;
    int kind = 0;
    for (;;) {
      const double* cb = W.contrib + (size_t)lvl * 3 * cap;
      const int32_t* ch = W.child + (size_t)lvl * 2 * cap;
      const int a = ch[idx], c = ch[cap + idx];
      if (a != RTC_WF_MISS) { r += cb[idx]; g += cb[cap + idx]; b += cb[2 * cap + idx]; }  // a miss wrote no contribution
      if (digest) dg += rtc_hit_hash(W.dig[(size_t)lvl * cap + idx], lvl, kind);
      if (c >= 0) { wait_idx[sp] = c; wait_lvl[sp] = lvl + 1; sp++; }
      if (a >= 0) { idx = a; lvl++; kind = 1; continue; }
      if (sp == 0) break;
      sp--;
      idx = wait_idx[sp]; lvl = wait_lvl[sp]; kind = 2;
    }
    rgb[3 * q + 0] = r; rgb[3 * q + 1] = g; rgb[3 * q + 2] = b;
    if (digest) pm.digest[q] = dg;
  }
}


static void launch_wf_ts(int v, bool count, unsigned grid, hipStream_t stream, const DScene& S, const DCamera& cam, const DPixelMap& pm, const DWave& W, int tl, int sl,
                         unsigned n0, int slot, int fuel_left, double* hit_t, int* hit_prim, int* hit_k, DStats* stats) {
  switch (v) {
    case 0: rtc_launch_wf_ts_v0(count, grid, stream, S, cam, pm, W, tl, sl, n0, slot, fuel_left, hit_t, hit_prim, hit_k, stats); break;
    case 1: rtc_launch_wf_ts_v1(count, grid, stream, S, cam, pm, W, tl, sl, n0, slot, fuel_left, hit_t, hit_prim, hit_k, stats); break;
    case 2: rtc_launch_wf_ts_v2(count, grid, stream, S, cam, pm, W, tl, sl, n0, slot, fuel_left, hit_t, hit_prim, hit_k, stats); break;
    case 3: rtc_launch_wf_ts_v3(count, grid, stream, S, cam, pm, W, tl, sl, n0, slot, fuel_left, hit_t, hit_prim, hit_k, stats); break;
    case 5: rtc_launch_wf_ts_v5(count, grid, stream, S, cam, pm, W, tl, sl, n0, slot, fuel_left, hit_t, hit_prim, hit_k, stats); break;
    default: rtc_launch_wf_ts_v4(count, grid, stream, S, cam, pm, W, tl, sl, n0, slot, fuel_left, hit_t, hit_prim, hit_k, stats); break;
  }
}

#ifndef RTC_EMU
// LDS-resident scene (rtc_device.hpp, LdsScene): bytes of dynamic LDS a block of the LDSC traversal kernel needs, or 0 when the
// scene does not qualify (program not in the kernel arguments, tables + stacks beyond a CU's 160 KB) or RTC_WF_LDS=0.
unsigned rtc_wavefront_lds_bytes(const DScene& S) {
  static const bool off = [] { const char* e = std::getenv("RTC_WF_LDS"); return e && e[0] == '0'; }();
  if (off || (rtc_variant(S) > 1 && rtc_variant(S) != 5)) return 0;
  const unsigned long long need = rtc_lds_table_bytes(S) + (unsigned long long)RTC_LDS_BLOCK * (unsigned)S.bvh_stack * sizeof(int);
  return need <= 158ull * 1024 ? (unsigned)need : 0u;
}
static bool launch_wf_ts_lds(int v, bool count, unsigned grid, unsigned lds, hipStream_t stream, const DScene& S, const DCamera& cam, const DPixelMap& pm, const DWave& W, int tl,
                             int sl, unsigned n0, int slot, int fuel_left, double* hit_t, int* hit_prim, int* hit_k, DStats* stats) {
  if (v == 0) return rtc_launch_wf_ts_lds_v0(count, grid, lds, stream, S, cam, pm, W, tl, sl, n0, slot, fuel_left, hit_t, hit_prim, hit_k, stats);
  if (v == 5) return rtc_launch_wf_ts_lds_v5(count, grid, lds, stream, S, cam, pm, W, tl, sl, n0, slot, fuel_left, hit_t, hit_prim, hit_k, stats);
  return rtc_launch_wf_ts_lds_v1(count, grid, lds, stream, S, cam, pm, W, tl, sl, n0, slot, fuel_left, hit_t, hit_prim, hit_k, stats);
}

// Grid of the wavefront traversal kernel: as many one-wave blocks as the chip holds at once (the kernel hands out chunks
// itself), from the occupancy the runtime reports for this scene's variant and LDS stack size.
unsigned rtc_wavefront_grid(const DScene& S, int n_cu) {
  const unsigned lds = rtc_stack_bytes(S);
  int per_cu = 8;
  switch (rtc_variant(S)) {
    case 0: per_cu = rtc_wf_ts_blocks_per_cu_v0(lds); break;
    case 1: per_cu = rtc_wf_ts_blocks_per_cu_v1(lds); break;
    case 2: per_cu = rtc_wf_ts_blocks_per_cu_v2(lds); break;
    case 3: per_cu = rtc_wf_ts_blocks_per_cu_v3(lds); break;
    case 5: per_cu = rtc_wf_ts_blocks_per_cu_v5(lds); break;
    default: per_cu = rtc_wf_ts_blocks_per_cu_v4(lds); break;
  }
  return (unsigned)std::max(1, n_cu * per_cu);
}
#endif

// work ids of a launch (tile padding included): the minimum DWave.cap
uint64_t rtc_wavefront_work(const DCamera& cam, const DPixelMap& pm) {
  if (pm.mode == 2 && cam.hsize >= 8 && pm.n % cam.hsize == 0) return (uint64_t)((cam.hsize + 7) / 8) * ((pm.n / cam.hsize + 7) / 8) * 64;
  return pm.n;
}

// One frame through the wavefront kernels.  The caller zeroed W.counts (RTC_WF_COUNTS entries) on the stream and sized the
// arrays for W.cap >= the work ids of the launch and fuel + 1 levels; `blocks` / `shade_blocks` = grid sizes of the traversal /
// shading kernels.
void rtc_launch_wavefront(const DScene& S, const DCamera& cam, const DPixelMap& pm, int fuel, const DWave& W, double* rgb, double* hit_t, int* hit_prim, int* hit_k,
                          DStats* stats, bool count, hipStream_t stream, unsigned blocks, unsigned shade_blocks) {
  if (pm.n == 0) return;
  const int v = rtc_variant(S);
  const unsigned n0 = (unsigned)rtc_wavefront_work(cam, pm);
#ifndef RTC_EMU
  const unsigned lds = rtc_wavefront_lds_bytes(S);
  const unsigned lds_blocks = std::max(1u, shade_blocks / 2u);  // one block per CU (the shading grid is two per CU)
#endif
  const dim3 sgrid(std::max(1u, shade_blocks)), sblock(RTC_WF_SHADE_BLOCK);
  // trace_0; shade_0; [shadow_0 + trace_1]; shade_1; ... [shadow_{fuel-1} + trace_fuel]; shade_fuel; shadow_fuel; sums
  for (int level = 0; level <= fuel + 1; level++) {
    const int tl = level <= fuel ? level : -1, sl = level - 1;
#ifndef RTC_EMU
    // (a device that refuses the LDS size — the opt-in is per device — runs the kernel that reads the tables from memory)
    if (!lds || !launch_wf_ts_lds(v, count, lds_blocks, lds, stream, S, cam, pm, W, tl, sl, n0, level, fuel - level, hit_t, hit_prim, hit_k, stats))
#endif
    launch_wf_ts(v, count, blocks, stream, S, cam, pm, W, tl, sl, n0, level, fuel - level, hit_t, hit_prim, hit_k, stats);
    if (level <= fuel) {
      if (count) hipLaunchKernelGGL((wf_shade<true>), sgrid, sblock, 0, stream, S, cam, pm, W, level, n0, fuel, stats);
      else if (S.all_plain) hipLaunchKernelGGL((wf_shade<false, false>), sgrid, sblock, 0, stream, S, cam, pm, W, level, n0, fuel, stats);
      else hipLaunchKernelGGL((wf_shade<false>), sgrid, sblock, 0, stream, S, cam, pm, W, level, n0, fuel, stats);
    }
  }
  hipLaunchKernelGGL(wf_gather, dim3(std::max(1u, std::min(blocks, (n0 + 255u) / 256u))), dim3(256), 0, stream, cam, pm, W, n0, rgb);
}

// ---- host-callable launcher (C++ linkage, used by rtc_scene.cpp) ------------------------------------------
// Color::clamp (src/color.rs:42-46) over a flat array of channel values.
__global__ void __launch_bounds__(256) rtc_quantize_kernel(const double* __restrict__ rgb, unsigned char* __restrict__ out, unsigned long long n) {
  unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    double x = rgb[i];
    double c = (x != x) ? 1.0 : (x < 1.0 ? x : 1.0);  // x.min(1.0): NaN -> 1.0
    c = c > 0.0 ? c : 0.0;                            // .max(0.0)
    double r = round(c * 255.0);                      // half away from zero
    out[i] = (unsigned char)(r <= 0.0 ? 0.0 : (r >= 255.0 ? 255.0 : r));
  }
}
// The kernels' SoA primary-hit rows -> the C ABI's 16-byte records (include/rtc.h rtc_hit), so that the host copy is one transfer.
__global__ void __launch_bounds__(256) rtc_pack_hits_kernel(const double* __restrict__ t, const int* __restrict__ prim, const int* __restrict__ k, DHit* __restrict__ out, unsigned long long n) {
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) {
    DHit h;
    h.t = t[i]; h.prim = prim[i]; h.k = k[i];
    out[i] = h;
  }
}
void rtc_launch_pack_hits(const double* t, const int* prim, const int* k, DHit* out, unsigned long long n, hipStream_t stream) {
  if (n == 0) return;
  unsigned long long blocks = (n + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(rtc_pack_hits_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, t, prim, k, out, n);
}
void rtc_launch_quantize(const double* rgb, unsigned char* out, unsigned long long n, hipStream_t stream) {
  if (n == 0) return;
  unsigned long long blocks = (n + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(rtc_quantize_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, rgb, out, n);
}


// Multi-GPU gather, last step: image row k + n j  <-  row j of replica k's dense tile in the slab.
// (bands of `band` rows: image row y is row ((y / band) / n) * band + y % band of replica (y / band) % n's tile)
__global__ void __launch_bounds__(256) rtc_deinterleave_kernel(const double* __restrict__ slab, double* __restrict__ image, unsigned rowlen, unsigned vsize, unsigned n,
                                                               unsigned max_rows, unsigned band) {
  const unsigned long long total = (unsigned long long)vsize * rowlen;
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (unsigned long long)gridDim.x * blockDim.x) {
    const unsigned y = (unsigned)(i / rowlen), x = (unsigned)(i % rowlen);
    const unsigned b = y / band;
    image[i] = slab[((unsigned long long)(b % n) * max_rows + (b / n) * band + y % band) * rowlen + x];
  }
}
// The same for quantised tiles (rtc_render_multi_rgb8: every replica quantises its own rows, so 3 bytes per pixel cross xGMI, not 24).
__global__ void __launch_bounds__(256) rtc_deinterleave8_kernel(const unsigned char* __restrict__ slab, unsigned char* __restrict__ image, unsigned rowlen, unsigned vsize,
                                                                unsigned n, unsigned max_rows, unsigned band) {
  const unsigned long long total = (unsigned long long)vsize * rowlen;
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (unsigned long long)gridDim.x * blockDim.x) {
    const unsigned y = (unsigned)(i / rowlen), x = (unsigned)(i % rowlen);
    const unsigned b = y / band;
    image[i] = slab[((unsigned long long)(b % n) * max_rows + (b / n) * band + y % band) * rowlen + x];
  }
}
void rtc_launch_deinterleave8(const unsigned char* slab, unsigned char* image, unsigned rowlen, unsigned vsize, unsigned n, unsigned max_rows, unsigned band, hipStream_t stream) {
  const unsigned long long total = (unsigned long long)vsize * rowlen;
  if (total == 0) return;
  unsigned long long blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(rtc_deinterleave8_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, slab, image, rowlen, vsize, n, max_rows, band ? band : 1u);
}
void rtc_launch_deinterleave(const double* slab, double* image, unsigned rowlen, unsigned vsize, unsigned n, unsigned max_rows, unsigned band, hipStream_t stream) {
  const unsigned long long total = (unsigned long long)vsize * rowlen;
  if (total == 0) return;
  unsigned long long blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(rtc_deinterleave_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, slab, image, rowlen, vsize, n, max_rows, band ? band : 1u);
}

// One-kernel path: one lane per work id (tile padding included).  big_scene: the accelerator does not fit the L2s.
void rtc_launch_trace(const DScene& S, const DCamera& cam, const DPixelMap& pm, int fuel, double* rgb, double* hit_t, int* hit_prim, int* hit_k,
                      DStats* stats, bool count, hipStream_t stream, bool big_scene) {
  if (pm.n == 0) return;
  const unsigned grid = (unsigned)((rtc_wavefront_work(cam, pm) + RTC_BLOCK - 1) / RTC_BLOCK);
  const int waves = big_scene ? 3 : 0;
  switch (rtc_variant(S)) {
    case 0: rtc_launch_trace_v0(count, waves, grid, stream, S, cam, pm, fuel, rgb, hit_t, hit_prim, hit_k, stats); break;
    case 1: rtc_launch_trace_v1(count, waves, grid, stream, S, cam, pm, fuel, rgb, hit_t, hit_prim, hit_k, stats); break;
    case 2: rtc_launch_trace_v2(count, waves, grid, stream, S, cam, pm, fuel, rgb, hit_t, hit_prim, hit_k, stats); break;
    case 3: rtc_launch_trace_v3(count, waves, grid, stream, S, cam, pm, fuel, rgb, hit_t, hit_prim, hit_k, stats); break;
    case 5: rtc_launch_trace_v5(count, waves, grid, stream, S, cam, pm, fuel, rgb, hit_t, hit_prim, hit_k, stats); break;
    default: rtc_launch_trace_v4(count, waves, grid, stream, S, cam, pm, fuel, rgb, hit_t, hit_prim, hit_k, stats); break;
  }
}
