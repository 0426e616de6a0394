// scene_build.hpp — pure host code (no HIP calls): validates an rtc_scene_desc and turns it into the arrays of
// device_scene.h: the traversal program, the exact group boxes, the accelerator BVHs and the SoA tables.
// rtc_scene.cpp uploads these arrays verbatim; tests/cpu_emu runs the kernel source on them on the CPU.
#pragma once
#include <chrono>
#include <cstdio>
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rtc.h"
#include "bvh_build.hpp"
#include "device_scene.h"
#include "host_scene.hpp"

namespace rtb {

// fn(begin, end) over [0, n) on a few threads (RTC_BUILD_THREADS, default: the hardware's, at most 8); small n runs inline.
template <class F>
inline void parallel_for(size_t n, size_t grain, F fn) {
  unsigned t = std::thread::hardware_concurrency();
  if (const char* e = std::getenv("RTC_BUILD_THREADS")) t = (unsigned)std::max(1, std::atoi(e));
  t = std::min<unsigned>(std::max(1u, t), 8u);
  if (n < 2 * grain || t < 2) { fn((size_t)0, n); return; }
  t = (unsigned)std::min<size_t>(t, n / grain);
  std::vector<std::thread> th;
  const size_t per = (n + t - 1) / t;
  for (unsigned k = 0; k < t; k++) {
    const size_t b = std::min(n, per * k), e = std::min(n, per * (k + 1));
    if (e > b) th.emplace_back([=] { fn(b, e); });
  }
  for (auto& x : th) x.join();
}


struct ProgramBuilder {
  const rtc_scene_desc& D;
  std::vector<DOp> ops;
  std::vector<double> group_box;
  std::vector<DBvhNode4> bvh_nodes;
  std::vector<double> mtri;
  // ONE item array serves BVH leaves, linear quirk lists and direction-grid cells (absolute indices): a traversal
  // step never has to choose between arrays.
  std::vector<int32_t> mtri_prim, items;
  std::vector<int32_t>&item_prim = items, &quirk_prim = items, &qitem = items;
  std::vector<int32_t> bvh_prims;  // primitives the analytic BVH reaches (their leaf refs carry the index)
  int max_stack = 0;  // worst-case traversal stack entries over all trees
  std::vector<DQuirkGrid> qgrids;
  std::vector<uint32_t> qcell;
  std::vector<double> bvh_frame;  // per BVH: centre xyz + inf-norm radius (DOp.c indexes it)
  std::vector<DCsg> csg;
  int max_depth = 0;
  std::string error;
  int status = RTC_OK;

  static constexpr size_t kMinAccel = 6;  // fewer bounded children than this stay a linear list

  // Cubes in the analytic BVH are padded by cube_pad object units: a ray with an object-space direction component below EPSILON is
  // treated as parallel to that slab pair (src/shape.rs:641-648), so the points it reports drift out of the cube by up to |t| EPSILON
  // along that axis.  With the pad, every such point of a ray whose reach |t| stays below cube_pad / EPSILON is inside the leaf's box:
  // those rays (all rays of a scene a few hundred units across) need no separate quirk scan for cubes (rtc_device.hpp, cubes_in_leaf).
  double cube_pad = 0.015625;
  bool local_bounds(const rtc_prim& p, double lo[3], double hi[3], bool for_accel = false) const {
    // bounds of every point the exact test can report, in object space (tighter than Geometry::bbox for open
    // cylinders, whose walls only exist for min < y < max: src/shape.rs:745,756)
    switch (p.geometry) {
      case RTC_SPHERE:
        for (int a = 0; a < 3; a++) { lo[a] = -1.0; hi[a] = 1.0; }
        return true;
      case RTC_CUBE:
        for (int a = 0; a < 3; a++) { lo[a] = -1.0 - (for_accel ? cube_pad : 0.0); hi[a] = 1.0 + (for_accel ? cube_pad : 0.0); }
        return true;
      case RTC_PLANE: return false;
      case RTC_CYLINDER:
      case RTC_CONE: {
        double mn = D.limits[2 * p.data], mx = D.limits[2 * p.data + 1];
        if (!std::isfinite(mn) || !std::isfinite(mx)) return false;
        double r = p.geometry == RTC_CYLINDER ? 1.0 : std::fmax(std::fabs(mn), std::fabs(mx));
        lo[0] = -r; hi[0] = r; lo[1] = std::fmin(mn, mx); hi[1] = std::fmax(mn, mx); lo[2] = -r; hi[2] = r;
        return true;
      }
      default: {
        const double* g = D.tri_p1e1e2 + 9 * (size_t)p.data;
        for (int a = 0; a < 3; a++) {
          double v0 = g[a], v1 = g[a] + g[3 + a], v2 = g[a] + g[6 + a];
          lo[a] = std::fmin(v0, std::fmin(v1, v2));
          hi[a] = std::fmax(v0, std::fmax(v1, v2));
          if (!std::isfinite(lo[a]) || !std::isfinite(hi[a])) return false;
        }
        return true;
      }
    }
  }

  bool world_bounds(const rtc_prim& p, bvh::Item* out) const {
    double lo[3], hi[3];
    if (!local_bounds(p, lo, hi, true)) return false;
    rth::M4 inv = rth::M4::from(D.xforms[p.xform].transform_inv), fwd;
    if (!inv.invert(&fwd)) return false;
    const double(*F)[4] = fwd.a;
    bool tight = false;
    if (p.geometry == RTC_SPHERE || p.geometry == RTC_CYLINDER || p.geometry == RTC_CONE) {
      // Exact AABB of the image of a solid of revolution about the object y axis: along world axis i the unit disc in the
      // object xz plane spans +-sqrt(F_i0^2 + F_i2^2) (times the largest radius), the sphere +-|row i of the 3x3 part|, and the
      // axis segment y in [ylo, yhi] spans F_i1 * y.  Tighter than the 8 corners of the object-space box by up to sqrt(3).
      double r = 1.0, ylo = 0.0, yhi = 0.0;
      if (p.geometry != RTC_SPHERE) { ylo = lo[1]; yhi = hi[1]; r = hi[0]; }
      tight = true;
      for (int a = 0; a < 3; a++) {
        double c = F[a][3], e;
        if (p.geometry == RTC_SPHERE) e = std::sqrt(F[a][0] * F[a][0] + F[a][1] * F[a][1] + F[a][2] * F[a][2]);
        else e = r * std::sqrt(F[a][0] * F[a][0] + F[a][2] * F[a][2]);
        double s0 = F[a][1] * ylo, s1 = F[a][1] * yhi;
        out->lo[a] = c + std::fmin(s0, s1) - e;
        out->hi[a] = c + std::fmax(s0, s1) + e;
        if (!std::isfinite(out->lo[a]) || !std::isfinite(out->hi[a])) tight = false;
      }
    }
    if (!tight) {
      for (int a = 0; a < 3; a++) { out->lo[a] = rth::kInf; out->hi[a] = -rth::kInf; }
      for (int k = 0; k < 8; k++) {
        rth::V4 c = fwd.apply(rth::pt((k & 4) ? hi[0] : lo[0], (k & 2) ? hi[1] : lo[1], (k & 1) ? hi[2] : lo[2]));
        double q[3] = {c.x, c.y, c.z};
        for (int a = 0; a < 3; a++) {
          if (!std::isfinite(q[a])) return false;
          out->lo[a] = std::fmin(out->lo[a], q[a]);
          out->hi[a] = std::fmax(out->hi[a], q[a]);
        }
      }
    }
    // the forward matrix is a numerical inverse of an inverse: widen by its conditioning-independent slop
    for (int a = 0; a < 3; a++) {
      double pad = 1e-9 * (std::fabs(out->lo[a]) + std::fabs(out->hi[a]) + (out->hi[a] - out->lo[a]));
      out->lo[a] -= pad;
      out->hi[a] += pad;
    }
    return true;
  }

  static constexpr int32_t kMinQuirkGrid = 8;

  // Direction grid for quirk_prim[q0, q0+qn).  A ray is a quirk ray for
  //   a cube iff |row_i(M) . d| < EPSILON for some object axis i          (src/shape.rs:641),
  //   a cone iff |(r0.d)^2 - (r1.d)^2 + (r2.d)^2| < EPSILON               (src/shape.rs:784),
  // with M = transform_inv's 3x3 part: conditions on the direction only.  For a unit direction u in a cell with centre c
  // and chord radius rho, |r.u| >= |r.c| - |r| rho, so the primitive is listed in the cell iff the bound cannot exclude
  // it for any ray of length >= RTC_QGRID_MIN_LEN; shorter rays take the linear scan.
  int32_t build_quirk_grid(int32_t q0, int32_t qn) {
    int bands = 0;
    for (int32_t i = 0; i < qn; i++) bands += D.prims[quirk_prim[q0 + i]].geometry == RTC_CUBE ? 3 : 1;
    // 256 x 256 cells per face whatever the list's length (~1.1 * bands / n candidates per cell; 1.5 MB of cell offsets): a short
    // list — the 51 cones of config2_cones — then has mostly empty cells and its scan is a lookup (n = 31: 4.98 ms, 64: 4.45, 128: 4.12, 256: 3.97)
    (void)bands;
    const int n = 256;
    DQuirkGrid g{n, (int32_t)qcell.size(), q0, qn};
    const double eps = 0.00001, lmin = RTC_QGRID_MIN_LEN;
    struct Rows { double r[3][3], len[3]; bool cube; int32_t prim; };
    std::vector<Rows> rows(qn);
    for (int32_t i = 0; i < qn; i++) {
      const rtc_prim& p = D.prims[quirk_prim[q0 + i]];
      const double* m = D.xforms[p.xform].transform_inv;
      Rows& R = rows[i];
      R.cube = p.geometry == RTC_CUBE;
      R.prim = quirk_prim[q0 + i];
      for (int a = 0; a < 3; a++) {
        for (int c = 0; c < 3; c++) R.r[a][c] = m[4 * a + c];
        R.len[a] = std::sqrt(R.r[a][0] * R.r[a][0] + R.r[a][1] * R.r[a][1] + R.r[a][2] * R.r[a][2]);
      }
    }
    auto dir = [](int face, double u, double v, double* out) {
      double s = (face & 1) ? -1.0 : 1.0;
      switch (face >> 1) {
        case 0: out[0] = s; out[1] = u; out[2] = v; break;
        case 1: out[0] = u; out[1] = s; out[2] = v; break;
        default: out[0] = u; out[1] = v; out[2] = s; break;
      }
      double l = std::sqrt(out[0] * out[0] + out[1] * out[1] + out[2] * out[2]);
      out[0] /= l; out[1] /= l; out[2] /= l;
    };
    for (int face = 0; face < 6; face++)
      for (int iv = 0; iv < n; iv++)
        for (int iu = 0; iu < n; iu++) {
          qcell.push_back((uint32_t)qitem.size());
          double c[3], rho = 0.0;
          dir(face, (iu + 0.5) / n * 2.0 - 1.0, (iv + 0.5) / n * 2.0 - 1.0, c);
          for (int k = 0; k < 4; k++) {
            double q[3];
            dir(face, (iu + (k & 1)) / (double)n * 2.0 - 1.0, (iv + (k >> 1)) / (double)n * 2.0 - 1.0, q);
            double dx = q[0] - c[0], dy = q[1] - c[1], dz = q[2] - c[2];
            rho = std::max(rho, std::sqrt(dx * dx + dy * dy + dz * dz));
          }
          rho = rho * (1.0 + 1e-6) + 1e-9;  // device and host may disagree on the cell of a direction on an edge
          for (const Rows& R : rows) {
            bool in = false;
            if (R.cube) {
              for (int a = 0; a < 3 && !in; a++) {
                double f = std::fabs(R.r[a][0] * c[0] + R.r[a][1] * c[1] + R.r[a][2] * c[2]);
                in = !(f > eps / lmin + R.len[a] * rho);
              }
            } else {
              double e0 = R.r[0][0] * c[0] + R.r[0][1] * c[1] + R.r[0][2] * c[2];
              double e1 = R.r[1][0] * c[0] + R.r[1][1] * c[1] + R.r[1][2] * c[2];
              double e2 = R.r[2][0] * c[0] + R.r[2][1] * c[1] + R.r[2][2] * c[2];
              double gq = std::fabs(e0 * e0 - e1 * e1 + e2 * e2);
              double lip = 2.0 * rho * (R.len[0] * R.len[0] + R.len[1] * R.len[1] + R.len[2] * R.len[2]);
              in = !(gq > eps / (lmin * lmin) + lip);
            }
            if (in) qitem.push_back(R.prim);
          }
        }
    qcell.push_back((uint32_t)qitem.size());
    qgrids.push_back(g);
    return (int32_t)qgrids.size() - 1;
  }

  // Light grids.  Every shadow ray towards light l runs along a line through the light's position, so which
  // primitives its segment can touch depends on its DIRECTION only: primitive B is a candidate for direction d iff some point x
  // of B's bounds has (l - x) / |l - x| = d.  Per light a cube map of n x n cells per face (the quirk grid's cell function) lists,
  // per cell, the primitives of the analytic BVH whose bounds have such a point for some direction of the cell, as the leaf
  // references the BVH itself would hand the walk: a shadow ray looks its cell up, puts the list on its traversal stack and runs
  // the walk's leaf loop — no frame, no node steps.  The lists are supersets of what the walk would reach (the walk also culls by
  // distance), the exact tests are the same, so the results are.  Returns 1 + index of light 0's grid in qgrids, 0 = not built.
  int32_t build_light_grids(const std::vector<bvh::Item>& boxes, const std::vector<int32_t>& ids) {
    const char* e = std::getenv("RTC_LIGHT_GRID");
    if ((e && e[0] == '0') || D.n_lights == 0 || boxes.size() < 16) return 0;
    const char* en = std::getenv("RTC_LIGHT_GRID_N");
    // cells per face edge: 256 (config 2: 12 % fewer candidates than 128, -4 % frame time; 64: +10 %) while the cell arrays of all
    // lights stay small beside the caches (1.5 MB per light at 256)
    // ... and while the build stays cheap: a scene of 10^4-10^5 bounded primitives gets coarser grids
    int n = en ? std::min(512, std::max(2, std::atoi(en))) : (D.n_lights <= 4 ? 256 : (D.n_lights <= 16 ? 128 : 64));
    if (!en && boxes.size() > 8192) n = std::min(n, boxes.size() > 65536 ? 64 : 128);
    // Work budget: every box is counted into (and later written to) every cell its projection covers, so a light inside a cloud of
    // large boxes costs boxes x cells; beyond 64 entries per cell on average the lists would be thrown away as RTC_LIGHT_CELL_WALK
    // anyway: no grids then (shadow rays walk the BVH, as they did before round 2), whatever the scene.
    const unsigned long long area_budget = 64ull * 6ull * (unsigned long long)n * (unsigned long long)n;
    // the first candidate is walked at once, the others wait on the traversal stack, whose depth the BVH decided: a cell with more
    // candidates than that holds the one reference RTC_LIGHT_CELL_WALK instead — its rays walk the BVH like any other ray
    const int max_list = std::max(8, max_stack + 1);
    const size_t g0 = qgrids.size(), c0 = qcell.size(), i0 = items.size();
    for (uint32_t l = 0; l < D.n_lights; l++) {
      const double* o = D.lights[l].origin;
      std::vector<uint32_t> count((size_t)6 * n * n + 1, 0u);
      struct Rect { int32_t face, u0, u1, v0, v1, ref; float dmin; };
      std::vector<Rect> rects;
      auto cell = [&](double u) {
        int i = (int)((u + 1.0) * 0.5 * (double)n);
        return i < 0 ? 0 : (i >= n ? n - 1 : i);
      };
      bool ok = std::isfinite(o[0]) && std::isfinite(o[1]) && std::isfinite(o[2]);
      unsigned long long area = 0;
      for (size_t k = 0; k < boxes.size() && ok; k++) {
        double a[3], b[3];  // w = l - x over the box
        for (int c = 0; c < 3; c++) { a[c] = o[c] - boxes[k].hi[c]; b[c] = o[c] - boxes[k].lo[c]; if (!(a[c] <= b[c])) ok = false; }
        const int32_t ref = ~(int32_t)((uint32_t)ids[k] << 3);
        // no point of the box is nearer to the light than this (rounded down): a shadow ray shorter than that starts between the
        // light and the box, which then lies behind its origin
        double d2 = 0.0;
        for (int c = 0; c < 3; c++) { const double g = a[c] > 0.0 ? a[c] : (b[c] < 0.0 ? -b[c] : 0.0); d2 += g * g; }
        float dmin = (float)(std::sqrt(d2) * (1.0 - 1e-6));
        dmin = dmin > 0.0f ? std::nextafterf(dmin, 0.0f) : 0.0f;
        for (int face = 0; face < 6 && ok; face++) {
          const int ax = face >> 1;
          const double m_lo = (face & 1) ? -b[ax] : a[ax], m_hi = (face & 1) ? -a[ax] : b[ax];  // range of the dominant |component|
          if (!(m_hi > 0.0)) continue;
          const int iu = ax == 0 ? 1 : 0, iv = ax == 2 ? 1 : 2;  // the cell function's (u, v): the other two axes in x, y, z order
          double lo2[2], hi2[2];
          bool none = false;
          for (int c = 0; c < 2; c++) {
            const int q = c == 0 ? iu : iv;
            double hi = b[q] > 0.0 ? (m_lo > 0.0 ? b[q] / m_lo : rth::kInf) : b[q] / m_hi;
            double lo = a[q] < 0.0 ? (m_lo > 0.0 ? a[q] / m_lo : -rth::kInf) : a[q] / m_hi;
            const double tol = 1e-9;  // device and host may disagree on the cell of a direction on an edge
            lo -= tol * (1.0 + std::fabs(lo)); hi += tol * (1.0 + std::fabs(hi));
            if (lo > 1.0 || hi < -1.0) none = true;
            lo2[c] = std::fmax(lo, -1.0); hi2[c] = std::fmin(hi, 1.0);
          }
          if (none) continue;
          Rect r{face, cell(lo2[0]), cell(hi2[0]), cell(lo2[1]), cell(hi2[1]), ref, dmin};
          area += (unsigned long long)(r.u1 - r.u0 + 1) * (unsigned long long)(r.v1 - r.v0 + 1);
          if (area > area_budget) { ok = false; break; }
          rects.push_back(r);
          for (int v = r.v0; v <= r.v1; v++)
            for (int u = r.u0; u <= r.u1; u++) count[((size_t)face * n + v) * n + u]++;
        }
      }
      if (!ok) {
        if (std::getenv("RTC_TIMING") && area > area_budget) std::fprintf(stderr, "[rtc-timing]   light grids dropped: light %u's boxes cover more than 64 entries per cell on average\n", l);
        qgrids.resize(g0); qcell.resize(c0); items.resize(i0);
        return 0;
      }
      uint32_t longest = 0, walks = 0;
      unsigned long long total = 0;
      for (uint32_t& c : count) {
        longest = std::max(longest, c);
        if ((int)c > max_list) { c = 0x80000001u; walks++; } else total += c;
      }
      if ((unsigned long long)walks * 2ull > (unsigned long long)count.size()) {  // most cells would send their rays through the BVH anyway
        qgrids.resize(g0); qcell.resize(c0); items.resize(i0);
        return 0;
      }
      if (std::getenv("RTC_TIMING"))
        std::fprintf(stderr, "[rtc-timing]   light grid %u: n %d, %llu items, longest list %u, %u cells left to the BVH walk (more than %d candidates)\n", l, n, total, longest, walks, max_list);
      // a cell's items: pairs {leaf reference, dmin as f32 bits}, nearest to the light first (8-byte aligned for one load each)
      if (items.size() & 1u) items.push_back(0);
      DQuirkGrid g{n, (int32_t)qcell.size(), 0, 0};
      std::vector<uint32_t> at(count.size());
      uint32_t run = (uint32_t)items.size();
      for (size_t c = 0; c < count.size(); c++) { at[c] = run; qcell.push_back(run); run += 2u * (count[c] & 0x7fffffffu); }
      items.resize(run);
      for (size_t c = 0; c < count.size(); c++) if (count[c] & 0x80000000u) { items[at[c]] = RTC_LIGHT_CELL_WALK; items[at[c] + 1] = 0; at[c] = 0xffffffffu; }
      std::sort(rects.begin(), rects.end(), [](const Rect& x, const Rect& y) { return x.dmin < y.dmin || (x.dmin == y.dmin && x.ref > y.ref); });
      for (const Rect& r : rects)
        for (int v = r.v0; v <= r.v1; v++)
          for (int u = r.u0; u <= r.u1; u++) {
            uint32_t& w = at[((size_t)r.face * n + v) * n + u];
            if (w == 0xffffffffu) continue;
            items[w] = r.ref;
            std::memcpy(&items[w + 1], &r.dmin, 4);
            w += 2;
          }
      qgrids.push_back(g);
    }
    return (int32_t)g0 + 1;
  }

  // Leaf sizes: an analytic primitive test (own matrix, geometry switch) is ~3x an inner node and runs at poor lane
  // utilisation, so analytic leaves hold one primitive; packed triangles are cheap and uniform, so mesh leaves hold four.
  static int leaf_size(bool mesh) {
    const char* e = std::getenv(mesh ? "RTC_LEAF_MESH" : "RTC_LEAF_ANALYTIC");
    return e ? std::atoi(e) : (mesh ? 4 : 1);
  }
  int32_t build_tree(const std::vector<bvh::Item>& items, std::vector<uint32_t>& order, uint32_t base, bool mesh, int32_t* frame_index) {
    // binned-SAH binary tree, then every other level folded away (4-wide nodes); if the worst-case traversal stack of
    // the result does not fit, a median-split tree (balanced) is used instead
    int depth = 0, depth2 = 0, need = 0;
    size_t n0 = bvh_nodes.size(), o0 = order.size();
    double frame[4];
    int32_t root = 0;
    for (int attempt = 0; attempt < 2; attempt++) {
      std::vector<DBvhNode> n2;
      bvh_nodes.resize(n0);
      order.resize(o0);
      auto t0_ = std::chrono::steady_clock::now();
      int32_t root2 = -1;
      // large meshes can have their binary tree built on the device (LBVH, bvh_device.hip); the rest of the
      // pipeline (4-wide collapse, stack bound, leaf packing) is shared, and a tree too deep for the stack falls to the median build
      const bool on_device = attempt == 0 && mesh && device_build && items.size() >= device_build_min;
      if (on_device) root2 = device_build(items, n2, order, base, leaf_size(mesh), frame);
      if (root2 >= 0) built_on_device++;
      else { n2.clear(); order.resize(o0); root2 = bvh::build(items, n2, order, base, &depth2, attempt == 1, leaf_size(mesh), frame); }
      auto t1_ = std::chrono::steady_clock::now();
      root = bvh::collapse4(n2, root2, bvh_nodes, &depth, &need);
      if (std::getenv("RTC_TIMING") && items.size() > 100000)
        std::fprintf(stderr, "[rtc-timing]   %s build %.3f s, collapse4 %.3f s (%zu items)\n", on_device && built_on_device ? "device LBVH" : "host SAH", std::chrono::duration<double>(t1_ - t0_).count(),
                     std::chrono::duration<double>(std::chrono::steady_clock::now() - t1_).count(), items.size());
      if (need <= RTC_BVH_STACK - 1) break;
    }
    if (need > RTC_BVH_STACK - 1) { error = "BVH too deep for the traversal stack"; status = RTC_ERR_UNSUPPORTED; }
    *frame_index = (int32_t)(bvh_frame.size() / 4);
    bvh_frame.insert(bvh_frame.end(), frame, frame + 4);
    max_depth = std::max(max_depth, depth);
    max_stack = std::max(max_stack, need);
    return root;
  }

  // Most intersections a primitive can push (src/shape.rs:608-619, :667-678, :746-767 + caps :689-722).
  static int max_hits(int geometry) { return geometry == RTC_CYLINDER || geometry == RTC_CONE ? 4 : (geometry == RTC_SPHERE || geometry == RTC_CUBE ? 2 : 1); }

  // Linear sub-program of a subtree that lives inside a CSG group: every node in DFS order, no accelerator.
  bool emit_linear(uint32_t b, uint32_t e, int depth, int* hits) {
    for (uint32_t i = b; i < e;) {
      const rtc_node& n = D.nodes[i];
      if (n.skip <= (int32_t)i || (uint32_t)n.skip > e) { error = "node skip out of range"; status = RTC_ERR_INVALID; return false; }
      if (n.kind == RTC_NODE_PRIM) {
        ops.push_back({OP_PRIM, n.ref, 0, 0, -1, {0, 0, 0}});
        *hits += max_hits(D.prims[n.ref].geometry);
      } else if (n.kind == RTC_NODE_AGGREGATION) {
        int32_t gi = (int32_t)(group_box.size() / 6);
        group_box.insert(group_box.end(), n.bbox_min, n.bbox_min + 3);
        group_box.insert(group_box.end(), n.bbox_max, n.bbox_max + 3);
        group_parent.push_back(-1);
        size_t at = ops.size();
        ops.push_back({OP_GROUP, gi, 0, 0, -1, {0, 0, 0}});
        if (!emit_linear(i + 1, (uint32_t)n.skip, depth, hits)) return false;
        ops[at].b = (int32_t)ops.size();
      } else {
        if (!emit_csg(i, depth + 1, hits)) return false;
      }
      i = (uint32_t)n.skip;
    }
    return true;
  }

  // A Union / Intersection / Difference group (src/shape.rs:74-101 asserts exactly two children).
  bool emit_csg(uint32_t i, int depth, int* outer_hits = nullptr, int32_t gate = -1) {
    const rtc_node& n = D.nodes[i];
    if (depth >= RTC_CSG_MAX_DEPTH) { error = "CSG groups nested deeper than RTC_CSG_MAX_DEPTH"; status = RTC_ERR_UNSUPPORTED; return false; }
    // children: first child = node i+1, second = the node after the first child's subtree
    uint32_t c0 = i + 1, end = (uint32_t)n.skip;
    if (c0 >= end) { error = "CSG group without children"; status = RTC_ERR_INVALID; return false; }
    uint32_t c1 = (uint32_t)D.nodes[c0].skip;
    if (c1 >= end || (uint32_t)D.nodes[c1].skip != end) { error = "CSG group kinds take exactly two children (src/shape.rs:82)"; status = RTC_ERR_INVALID; return false; }
    auto first_prim = [&](uint32_t from) {  // sequence number of the first primitive at or after node `from`
      for (uint32_t k = from; k < D.n_nodes; k++)
        if (D.nodes[k].kind == RTC_NODE_PRIM) return D.nodes[k].ref;
      return (int32_t)D.n_prims;
    };
    DCsg rec{n.kind, first_prim(c0), first_prim(c1), 0};
    // an empty left subtree (groups without primitives) has left_first == left_end: nothing is "in the left child"
    int32_t ci = (int32_t)csg.size();
    csg.push_back(rec);
    int32_t gi = (int32_t)(group_box.size() / 6);
    group_box.insert(group_box.end(), n.bbox_min, n.bbox_min + 3);
    group_box.insert(group_box.end(), n.bbox_max, n.bbox_max + 3);
    size_t at = ops.size();
    group_parent.push_back(-1);
    ops.push_back({OP_CSG, gi, 0, ci, gate, {0, 0, 0}});
    int hits = 0;
    if (!emit_linear(c0, end, depth, &hits)) return false;
    ops[at].b = (int32_t)ops.size();
    ops.push_back({OP_CSG_END, 0, 0, ci, -1, {0, 0, 0}});
    csg_max_hits = std::max(csg_max_hits, hits);  // beyond RTC_CSG_MAX_HITS the launches use a slab in device memory
    if (outer_hits) *outer_hits += hits;
    return true;
  }

  int csg_max_hits = 0;
  bool have_abvh = false;        // an analytic BVH with padded cubes exists: its frame (centre, inf-norm radius) bounds a ray's reach
  double abvh_frame[4] = {0, 0, 0, 0};
  size_t n_plain_items = (size_t)-1;  // items[0, n_plain_items) are primitive indices (quirk lists, quirk-grid cells); the rest light-grid pairs
  std::vector<int32_t>& items_member() { return this->items; }
  bvh::DeviceBuildFn device_build = nullptr;  // set by rtc_scene_create when the accelerator is to be built on the device
  size_t device_build_min = 4096;
  int built_on_device = 0;

  // ---- whole-scene emission: aggregation groups are dissolved into per-primitive gates -------------------------------------
  std::vector<int32_t> prim_gcond;   // per primitive
  std::vector<int32_t> group_parent; // per group box index
  struct PendingCsg { uint32_t node; int32_t g; };
  std::vector<PendingCsg> pending_csg;
  std::vector<int32_t> top_prims;    // primitives outside any CSG, in DFS order

  bool collect(uint32_t b, uint32_t e, int32_t g) {
    for (uint32_t i = b; i < e;) {
      const rtc_node& n = D.nodes[i];
      if (n.skip <= (int32_t)i || (uint32_t)n.skip > e) { error = "node skip out of range"; status = RTC_ERR_INVALID; return false; }
      if (n.kind == RTC_NODE_PRIM) {
        prim_gcond[n.ref] = g;
        top_prims.push_back(n.ref);
      } else if (n.kind == RTC_NODE_AGGREGATION) {
        int32_t gi = (int32_t)(group_box.size() / 6);
        group_box.insert(group_box.end(), n.bbox_min, n.bbox_min + 3);
        group_box.insert(group_box.end(), n.bbox_max, n.bbox_max + 3);
        group_parent.push_back(g);
        if (!collect(i + 1, (uint32_t)n.skip, gi)) return false;
      } else {
        pending_csg.push_back({i, g});
      }
      i = (uint32_t)n.skip;
    }
    return true;
  }

  // Emits the program for the whole world.
  bool emit(uint32_t b, uint32_t e) {
    prim_gcond.assign(D.n_prims, -1);
    if (!collect(b, e, -1)) return false;
    // 1. triangles that share one matrix AND one gate -> object-space mesh BVHs
    std::map<std::pair<int32_t, int32_t>, std::vector<int32_t>> meshes;
    std::vector<int32_t> rest;
    for (int32_t pi : top_prims) {
      const rtc_prim& p = D.prims[pi];
      if (p.geometry >= RTC_TRIANGLE) meshes[{p.xform, prim_gcond[pi]}].push_back(pi); else rest.push_back(pi);
    }
    for (auto& kv : meshes) {
      std::vector<bvh::Item> items;
      std::vector<int32_t> ids;
      for (int32_t pi : kv.second) {
        bvh::Item it;
        if (local_bounds(D.prims[pi], it.lo, it.hi)) { items.push_back(it); ids.push_back(pi); }
        else rest.push_back(pi);
      }
      if (items.size() < kMinAccel) { for (int32_t pi : ids) rest.push_back(pi); continue; }
      std::vector<uint32_t> order;
      uint32_t base = (uint32_t)mtri_prim.size();
      int32_t fi = 0;
      int32_t root = build_tree(items, order, base, true, &fi);
      {  // the mesh's triangles in leaf order (a gather of 72 B records: threads for the large meshes)
        const size_t at = mtri_prim.size(), cnt = order.size();
        mtri.resize((at + cnt) * 9);
        mtri_prim.resize(at + cnt);
        parallel_for(cnt, 65536, [&](size_t b, size_t e) {
          for (size_t j = b; j < e; j++) {
            const int32_t pi = ids[order[j]];
            std::memcpy(&mtri[(at + j) * 9], D.tri_p1e1e2 + 9 * (size_t)D.prims[pi].data, 9 * sizeof(double));
            mtri_prim[at + j] = pi;
          }
        });
      }
      ops.push_back({OP_MESH, root, kv.first.first, fi, kv.first.second, {0, 0, 0}});
    }
    // 2. every other bounded primitive of the scene -> ONE world-space BVH; unbounded ones stay linear
    {
      std::vector<bvh::Item> items;
      std::vector<int32_t> ids, linear;
      for (int32_t pi : rest) {
        bvh::Item it;
        if (world_bounds(D.prims[pi], &it)) { items.push_back(it); ids.push_back(pi); } else linear.push_back(pi);
      }
      if (items.size() < kMinAccel) { for (int32_t pi : ids) linear.push_back(pi); ids.clear(); items.clear(); }
      std::sort(linear.begin(), linear.end());
      for (int32_t pi : linear) ops.push_back({OP_PRIM, pi, 0, 0, -1, {0, 0, 0}});
      if (!items.empty()) {
        std::vector<uint32_t> order;
        uint32_t base = (uint32_t)item_prim.size();
        int32_t fi = 0;
        size_t first_node = bvh_nodes.size();
        int32_t root = build_tree(items, order, base, false, &fi);
        // analytic leaves hold ONE primitive: the leaf ref carries the primitive index itself (no item indirection)
        // bit 0 of the (otherwise unused) item-count field: the primitive is a sphere or a cube — 0 or 2 intersections, both inside
        // its bounds — which lets a container pass skip it unless the hit point lies inside the leaf's box (rtc_device.hpp, make_frame_point)
        auto direct = [&](int32_t ref) {
          if (ref >= 0) return ref;
          const int32_t pi = ids[order[((uint32_t)~ref >> 3) - base]];
          const uint32_t two = (D.prims[pi].geometry == RTC_SPHERE || D.prims[pi].geometry == RTC_CUBE) ? 1u : 0u;
          return ~(int32_t)(((uint32_t)pi << 3) | two);
        };
        for (size_t ni = first_node; ni < bvh_nodes.size(); ni++)
          for (int32_t& c : bvh_nodes[ni].c) c = direct(c);
        for (int32_t pi : ids) bvh_prims.push_back(pi);
        root = direct(root);
        const size_t bvh_op = ops.size();
        ops.push_back({OP_BVH, root, 0, fi, -1, {0, 0, 0}});
        // quirk lists: cubes and cones apart — c = 1 marks the cubes' op, which a ray of short reach skips (cube_pad above)
        for (int kind = 0; kind < 2; kind++) {
          int32_t q0 = (int32_t)quirk_prim.size();
          for (int32_t pi : ids)
            if (D.prims[pi].geometry == (kind == 0 ? RTC_CUBE : RTC_CONE)) quirk_prim.push_back(pi);
          int32_t qn = (int32_t)quirk_prim.size() - q0;
          const int32_t tag = (kind == 0 && cube_pad > 0.0) ? 1 : 0;
          if (qn >= kMinQuirkGrid) ops.push_back({OP_QGRID, build_quirk_grid(q0, qn), 0, tag, -1, {0, 0, 0}});
          else if (qn > 0) ops.push_back({OP_QUIRK, q0, qn, tag, -1, {0, 0, 0}});
        }
        if (cube_pad > 0.0) {
          have_abvh = true;
          std::memcpy(abvh_frame, &bvh_frame[(size_t)fi * 4], sizeof(abvh_frame));
        }
        // light grids last: their {reference, distance} pairs form the tail of the item array, after every plain primitive index
        n_plain_items = items_member().size();
        if (root >= 0) ops[bvh_op].b = build_light_grids(items, ids);  // (a root that is itself a leaf needs no help)
      }
    }
    // 3. CSG groups: linear sub-programs, gated by their enclosing aggregation groups
    for (const PendingCsg& c : pending_csg)
      if (!emit_csg(c.node, 0, nullptr, c.g)) return false;
    return true;
  }
};

inline int validate(const rtc_scene_desc& D, std::string* err) {
  auto bad = [&](const char* m) { *err = m; return RTC_ERR_INVALID; };
  if (D.n_lights > 0 && !D.lights) return bad("lights is NULL");
  for (uint32_t i = 0; i < D.n_prims; i++) {
    const rtc_prim& p = D.prims[i];
    if (p.geometry < RTC_SPHERE || p.geometry > RTC_SMOOTH_TRIANGLE) return bad("primitive geometry out of range");
    if (p.material < 0 || (uint32_t)p.material >= D.n_materials) return bad("primitive material index out of range");
    if (p.xform < 0 || (uint32_t)p.xform >= D.n_xforms) return bad("primitive xform index out of range");
    if ((p.geometry == RTC_CYLINDER || p.geometry == RTC_CONE) && (p.data < 0 || (uint32_t)p.data >= D.n_limits)) return bad("limits index out of range");
    if (p.geometry >= RTC_TRIANGLE && (p.data < 0 || (uint32_t)p.data >= D.n_tris)) return bad("triangle index out of range");
  }
  uint32_t seen = 0;
  for (uint32_t i = 0; i < D.n_nodes; i++) {
    const rtc_node& n = D.nodes[i];
    if (n.kind == RTC_NODE_PRIM) {
      if (n.ref != (int32_t)seen) return bad("primitive nodes must reference primitives in DFS order");
      seen++;
      if (n.skip != (int32_t)i + 1) return bad("primitive node skip must be index+1");
    } else if (n.kind < RTC_NODE_UNION || n.kind > RTC_NODE_AGGREGATION) return bad("node kind out of range");
    if (n.skip <= (int32_t)i || (uint32_t)n.skip > D.n_nodes) return bad("node skip out of range");
  }
  if (seen != D.n_prims) return bad("node array does not cover every primitive exactly once");
  for (uint32_t i = 0; i < D.n_materials; i++) {
    if (D.materials[i].pattern < 0 || (uint32_t)D.materials[i].pattern >= D.n_pattern_nodes) return bad("material pattern index out of range");
    // the device decides at fuel 0 whether the Schlick reflectance is NaN from eye . normal alone (rtc_device.hpp schlick_or_nan): that
    // holds for indices the formula's r0 = ((n1 - n2) / (n1 + n2))^2 is finite for
    const double ri = D.materials[i].refractive_index;
    if (!(ri > 1e-70 && ri < 1e70)) {
      *err = "a material's refractive_index is not a positive finite number (1e-70 .. 1e70)";
      return RTC_ERR_UNSUPPORTED;
    }
  }
  // pattern nodes: children need not precede parents; check indices, that the graph is a forest without cycles (tree depth can be
  // anything: the reference's Box tree is unbounded, src/material.rs:60-65) and the number of colour frames the device's walk
  // keeps on one path (rtc_device.hpp pattern_color: Blend / RingGradient / Gradient mixtures and colour jitters): <= RTC_MAX_PATTERN_DEPTH
  std::vector<int> depth(D.n_pattern_nodes, 0), frames(D.n_pattern_nodes, 0);
  for (uint32_t pass = 0; pass <= D.n_pattern_nodes + 1; pass++) {
    bool changed = false;
    for (uint32_t i = 0; i < D.n_pattern_nodes; i++) {
      const rtc_pattern_node& p = D.pattern_nodes[i];
      if (p.tag < RTC_PAT_DEBUG || p.tag > RTC_PAT_MIXTURE) return bad("pattern tag out of range");
      int d = 1, f = 0;
      if (p.tag >= RTC_PAT_JITTER) {
        if (p.left < 0 || (uint32_t)p.left >= D.n_pattern_nodes) return bad("pattern child out of range");
        d = std::max(d, 1 + depth[p.left]);
        f = std::max(f, frames[p.left]);
      }
      if (p.tag == RTC_PAT_MIXTURE) {
        if (p.right < 0 || (uint32_t)p.right >= D.n_pattern_nodes) return bad("pattern child out of range");
        if (p.kind < RTC_MIX_BLEND || p.kind > RTC_MIX_STRIPES) return bad("mixture kind out of range");
        d = std::max(d, 1 + depth[p.right]);
        f = std::max(f, frames[p.right]);
      }
      const bool keeps = (p.tag == RTC_PAT_JITTER && p.kind == RTC_JITTER_COLOR) ||
                         (p.tag == RTC_PAT_MIXTURE && (p.kind == RTC_MIX_BLEND || p.kind == RTC_MIX_RING_GRADIENT || p.kind == RTC_MIX_GRADIENT));
      f += keeps ? 1 : 0;
      if (d != depth[i] || f != frames[i]) { depth[i] = d; frames[i] = f; changed = true; }
      if ((uint32_t)d > D.n_pattern_nodes) return bad("pattern nodes form a cycle");
      if (f > RTC_MAX_PATTERN_DEPTH) {
        *err = "a pattern keeps more than RTC_MAX_PATTERN_DEPTH colour frames on one path (blends / gradients / colour jitters nested deeper than 8)";
        return RTC_ERR_UNSUPPORTED;
      }
    }
    if (!changed) break;
  }
  return RTC_OK;
}


// All arrays of one scene, host side.  `view()` gives a DScene whose pointers address these vectors.
struct HostArrays {
  std::vector<DOp> ops;
  std::vector<double> group_box;
  std::vector<int32_t> group_parent;
  std::vector<DBvhNode4> bvh;
  std::vector<double> mtri;
  std::vector<int32_t> mtri_prim, items;  // items: BVH leaf items + quirk lists + grid cells, absolute indices
  size_t n_plain_items = (size_t)-1;       // items from here on are light-grid {leaf reference, distance} pairs, not primitive indices
  std::vector<DQuirkGrid> qgrids;
  std::vector<uint32_t> qcell;
  std::vector<double> bvh_frame;
  std::vector<DCsg> csg;
  std::vector<DPrim> prims;
  std::vector<DPrimI> pisect;
  std::vector<int32_t> bvh_prims;  // host only: primitives the analytic BVH reaches
  std::vector<double> xf_inv, xf_matinv, limits, tri_geo, tri_nrm, mat;
  std::vector<int32_t> mat_pattern;
  std::vector<DPat> pats;
  std::vector<double> lights;
  mutable int backface_cached = -1;  // view(): the magnitude scan behind DScene.backface_skip, done once
  int32_t n_lights = 0, all_cast_shadow = 1, bvh_depth = 0, bvh_stack = 8, csg_max_hits = 0, built_on_device = 0;
  double quirk_reach2 = 0.0, abvh_frame[4] = {0, 0, 0, 0};  // see DScene

  // Shape::normal (src/shape.rs:419-427) of a plane — local normal vector(0, 1, 0), src/shape.rs:896 — exactly as the device's
  // prepare_state evaluates it (rtc_device.hpp: the same expressions in the same order; this translation unit is compiled with
  // -ffp-contract=off like the kernels, sqrt and / are correctly rounded on both sides): the bits a lane would compute per hit.
  static void plane_world_normal(const double* m, double n[3]) {
    const double lx = 0.0, ly = 1.0, lz = 0.0;
    const double wx = m[0] * lx + m[4] * ly + m[8] * lz + 0.0;
    const double wy = m[1] * lx + m[5] * ly + m[9] * lz + 0.0;
    const double wz = m[2] * lx + m[6] * ly + m[10] * lz + 0.0;
    const double mag = std::sqrt(wx * wx + wy * wy + wz * wz);
    n[0] = wx / mag; n[1] = wy / mag; n[2] = wz / mag;
  }

  // DScene.kops / kplanes (device_scene.h): a short, jump-free program travels in the kernel arguments.
  void fill_kernarg_program(DScene& d) const {
    d.n_kops = 0; d.n_kplanes = 0;
    if (std::getenv("RTC_NO_KOPS")) return;
    // (per-primitive gates — has_groups == 2 — do not change the op sequence: the gate is part of a primitive's own test.  Kernel
    // variant 5 serves such programs; RTC_KOPS_GROUPS=0 sends them back to the program in memory.)
    static const bool kops_groups = [] { const char* e = std::getenv("RTC_KOPS_GROUPS"); return !(e && e[0] == '0'); }();
    if (ops.size() > RTC_KOPS || d.has_csg || (d.has_groups == 2 && !kops_groups)) return;
    for (const DOp& o : ops) if (o.op == OP_GROUP || o.op == OP_CSG || o.op == OP_CSG_END) return;
    int n_aux = 0;
    bool have_qgrid = false;
    for (size_t i = 0; i < ops.size(); i++) {
      DOp o = ops[i];
      o.c = (o.op == OP_PRIM) ? -1 : o.c;
      o.pad[0] = -1;
      if ((o.op == OP_BVH || o.op == OP_MESH) && n_aux < RTC_KAUX && o.a >= 0) {
        DKAux& A = d.kaux[n_aux];
        std::memcpy(A.frame, &bvh_frame[(size_t)o.c * 4], 4 * sizeof(double));
        if (o.op == OP_MESH) std::memcpy(A.xf, &xf_inv[(size_t)o.b * 12], 12 * sizeof(double));
        else std::memset(A.xf, 0, sizeof(A.xf));
        A.root = bvh[(size_t)o.a];
        o.pad[0] = n_aux++;
      }
      if (o.op == OP_QGRID && !have_qgrid) { d.kqgrid = qgrids[(size_t)o.a]; have_qgrid = true; o.pad[0] = 0; }
      if (o.op == OP_PRIM && prims[(size_t)o.a].geom == RTC_PLANE && d.n_kplanes < RTC_KPLANES) {
        DPlaneK& k = d.kplanes[d.n_kplanes];
        std::memcpy(k.row, &xf_inv[(size_t)prims[(size_t)o.a].xform * 12 + 4], 4 * sizeof(double));
        k.prim = o.a;
        k.gcond = prims[(size_t)o.a].gcond;
        plane_world_normal(&xf_inv[(size_t)prims[(size_t)o.a].xform * 12], k.n);
        o.c = d.n_kplanes++;
      }
      d.kops[i] = o;
    }
    d.n_kops = (int32_t)ops.size();
  }

  DScene view() const {
    DScene d{};
    d.ops = ops.data(); d.group_box = group_box.data(); d.group_parent = group_parent.data(); d.bvh = bvh.data(); d.mtri = mtri.data(); d.mtri_prim = mtri_prim.data();
    d.item_prim = items.data(); d.quirk_prim = items.data(); d.qgrids = qgrids.data(); d.qcell = qcell.data(); d.bvh_frame = bvh_frame.data(); d.csg = csg.data(); d.qitem = items.data(); d.prims = prims.data(); d.pisect = pisect.data(); d.xf_inv = xf_inv.data(); d.xf_matinv = xf_matinv.data(); d.limits = limits.data();
    d.tri_geo = tri_geo.data(); d.tri_nrm = tri_nrm.data(); d.mat = mat.data(); d.mat_pattern = mat_pattern.data(); d.pats = pats.data();
    d.lights = lights.data();
    d.n_ops = (int32_t)ops.size(); d.n_prims = (int32_t)prims.size(); d.n_recs = (int32_t)pisect.size(); d.n_lights = n_lights; d.all_cast_shadow = all_cast_shadow;
    d.n_bvh = (int32_t)bvh.size(); d.n_items = (int32_t)items.size(); d.n_mtri = (int32_t)mtri_prim.size(); d.n_quirk = (int32_t)items.size();
    d.n_qitem = (int32_t)items.size(); d.n_qcell = (int32_t)qcell.size(); d.n_groups = (int32_t)(group_box.size() / 6); d.n_qgrids = (int32_t)qgrids.size();
    d.bvh_stack = bvh_stack;
    d.csg_max_hits = csg_max_hits;
    d.csg_slab = nullptr;
    d.has_mesh = 0;
    d.has_csg = 0;
    d.has_groups = 0;
    for (const DOp& o : ops) if (o.g >= 0) d.has_groups = 1;
    // a mesh triangle inherits its OP_MESH gate; only primitives reached one by one need their own
    for (const DOp& o : ops) if ((o.op == OP_PRIM) && prims[o.a].gcond >= 0) d.has_groups = 2;
    for (size_t i = 0; i < std::min(items.size(), n_plain_items); i++) if (prims[items[i]].gcond >= 0) d.has_groups = 2;  // (light-grid items name bvh_prims)
    for (int32_t pi : bvh_prims) if (prims[pi].gcond >= 0) d.has_groups = 2;
    for (const DOp& o : ops) { if (o.op == OP_MESH) d.has_mesh = 1; if (o.op == OP_CSG) d.has_csg = 1; }
    fill_kernarg_program(d);
    d.quirk_reach2 = quirk_reach2;
    std::memcpy(d.abvh_frame, abvh_frame, sizeof(d.abvh_frame));
    d.light_grid_first = 0;
    for (const DOp& o : ops) if (o.op == OP_BVH && o.b > 0) d.light_grid_first = o.b;
    d.light_grid_n = 0; d.light_grid_cell_off = 0;
    if (d.light_grid_first > 0) { d.light_grid_n = qgrids[(size_t)d.light_grid_first - 1].n; d.light_grid_cell_off = qgrids[(size_t)d.light_grid_first - 1].cell_off; }
    d.all_plain = 1;
    for (int32_t root : mat_pattern) if (root < 0 || (size_t)root >= pats.size() || pats[(size_t)root].tag != 1) d.all_plain = 0;
    d.no_glass_mirror = 1;
    for (size_t m = 0; m + 7 < mat.size(); m += 8) if (mat[m + 4] != 0.0 && mat[m + 5] != 0.0) d.no_glass_mirror = 0;
    {
      // light_is_behind() relies on a finite ray never producing a NaN t (the reference would panic on one in a list of two or more,
      // whether or not the shadow test's answer matters): true while no intermediate of the intersection formulas can overflow
      const char* env = std::getenv("RTC_BACKFACE_SKIP");
      const bool on = !(env && env[0] == '0');
      auto bounded = [](const std::vector<double>& v) {  // (9 doubles per triangle: threaded for the meshes that matter)
        std::atomic<bool> ok{true};
        parallel_for(v.size(), 1u << 20, [&](size_t b, size_t e) {
          for (size_t i = b; i < e; i++) if (!(std::fabs(v[i]) < 1e30)) { ok = false; return; }
        });
        return ok.load();
      };
      // ... nor underflow: a sphere divides by a = |object-space direction|^2 unchecked, so the world -> object matrices must not shrink
      // a unit vector below ~1e-90: sigma_min >= |det| / |M|_F^2 with |det| >= 1e-30 and entries < 1e30
      auto not_flat = [](const std::vector<double>& m) {
        for (size_t i = 0; i + 11 < m.size(); i += 12) {
          const double* r = &m[i];
          const double det = r[0] * (r[5] * r[10] - r[6] * r[9]) - r[1] * (r[4] * r[10] - r[6] * r[8]) + r[2] * (r[4] * r[9] - r[5] * r[8]);
          if (!(std::fabs(det) >= 1e-30)) return false;
        }
        return true;
      };
      if (backface_cached < 0) backface_cached = (bounded(xf_inv) && not_flat(xf_inv) && bounded(tri_geo) && bounded(lights)) ? 1 : 0;
      d.backface_skip = on ? backface_cached : 0;
    }
    d.has_recs = 0;
    for (size_t i = 0; i < ops.size(); i++) {
      const DOp& o = ops[i];
      if (o.op == OP_BVH || o.op == OP_QUIRK || o.op == OP_QGRID || o.op == OP_CSG) d.has_recs = 1;
      if (o.op == OP_PRIM && (d.n_kops == 0 || d.kops[i].c < 0)) d.has_recs = 1;  // not a plane record of the kernel arguments
    }
    return d;
  }
};

// desc -> arrays.  Returns an RTC_* status; message in *err.
inline int build_arrays(const rtc_scene_desc& D, HostArrays* H, std::string* err, bvh::DeviceBuildFn device_build = nullptr, size_t device_build_min = 4096) {
  int rc = validate(D, err);
  if (rc != RTC_OK) return rc;
  if (D.n_lights > 64) { *err = "more than 64 lights (the wavefront path keeps one shadow bit per light)"; return RTC_ERR_UNSUPPORTED; }
  const bool timing = std::getenv("RTC_TIMING") != nullptr;
  auto t_start = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!timing) return;
    auto now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[rtc-timing] %-28s %.3f s\n", what, std::chrono::duration<double>(now - t_start).count());
    t_start = now;
  };
  lap("validate");
  ProgramBuilder pb{D};
  pb.device_build = device_build;
  pb.device_build_min = device_build_min;
  if (const char* e = std::getenv("RTC_CUBE_PAD")) pb.cube_pad = std::max(0.0, std::atof(e));
  for (uint32_t i = 0; i < D.n_prims && pb.cube_pad > 0.0; i++) {  // the reach argument needs |t| EPSILON s << |t| |d| (s = the cube's scale)
    if (D.prims[i].geometry != RTC_CUBE) continue;
    rth::M4 inv = rth::M4::from(D.xforms[D.prims[i].xform].transform_inv), fwd;
    if (!inv.invert(&fwd)) { pb.cube_pad = 0.0; break; }
    for (int c = 0; c < 3; c++) {
      const double nrm = std::sqrt(fwd.a[0][c] * fwd.a[0][c] + fwd.a[1][c] * fwd.a[1][c] + fwd.a[2][c] * fwd.a[2][c]);
      if (!(nrm < 1e3)) pb.cube_pad = 0.0;
    }
  }
  if (const char* e = std::getenv("RTC_DEVICE_BVH_MIN")) pb.device_build_min = (size_t)std::strtoull(e, nullptr, 10);
  if (!pb.emit(0, D.n_nodes) || pb.status != RTC_OK) { *err = pb.error; return pb.status != RTC_OK ? pb.status : RTC_ERR_INVALID; }
  lap("program + BVH build");
  H->prims.resize(D.n_prims);
  std::atomic<int> all_cast{1};
  parallel_for(D.n_prims, 65536, [&](size_t b, size_t e) {
    int cast = 1;
    for (size_t i = b; i < e; i++) {
      H->prims[i] = {D.prims[i].geometry, D.prims[i].flags, D.prims[i].material, D.prims[i].xform, D.prims[i].data, pb.prim_gcond.empty() ? -1 : pb.prim_gcond[i], {0, 0}};
      if (!(D.prims[i].flags & RTC_FLAG_CASTS_SHADOW)) cast = 0;
    }
    if (!cast) all_cast.store(0);
  });
  H->all_cast_shadow = all_cast.load();
  H->xf_inv.resize((size_t)D.n_xforms * 12);
  H->xf_matinv.resize((size_t)D.n_xforms * 16);
  for (uint32_t i = 0; i < D.n_xforms; i++) {
    std::memcpy(&H->xf_inv[(size_t)i * 12], D.xforms[i].transform_inv, 12 * sizeof(double));
    std::memcpy(&H->xf_matinv[(size_t)i * 16], D.xforms[i].material_inv, 16 * sizeof(double));
  }
  H->limits.assign(D.limits, D.limits + (size_t)D.n_limits * 2);
  H->tri_geo.assign(D.tri_p1e1e2, D.tri_p1e1e2 + (size_t)D.n_tris * 9);
  H->tri_nrm.assign(D.tri_normals, D.tri_normals + (size_t)D.n_tris * 9);
  H->mat.assign((size_t)D.n_materials * 8, 0.0);
  H->mat_pattern.resize(D.n_materials);
  for (uint32_t i = 0; i < D.n_materials; i++) {
    const rtc_material& m = D.materials[i];
    double* q = &H->mat[(size_t)i * 8];
    q[0] = m.ambient; q[1] = m.diffuse; q[2] = m.specular; q[3] = m.shininess; q[4] = m.reflective; q[5] = m.transparency; q[6] = m.refractive_index;
    H->mat_pattern[i] = m.pattern;
  }
  H->pats.resize(D.n_pattern_nodes);
  for (uint32_t i = 0; i < D.n_pattern_nodes; i++) {
    const rtc_pattern_node& p = D.pattern_nodes[i];
    DPat& q = H->pats[i];
    std::memset(&q, 0, sizeof(q));
    q.tag = p.tag; q.kind = p.kind; q.noise_kind = p.noise_kind; q.octaves = p.octaves; q.left = p.left; q.right = p.right; q.scale = p.scale;
    std::memcpy(q.color, p.color, sizeof(q.color));
    std::memcpy(q.m, p.transform_inv, sizeof(q.m));
  }
  H->lights.resize((size_t)D.n_lights * 6);
  for (uint32_t i = 0; i < D.n_lights; i++) {
    std::memcpy(&H->lights[(size_t)i * 6], D.lights[i].intensity, 3 * sizeof(double));
    std::memcpy(&H->lights[(size_t)i * 6 + 3], D.lights[i].origin, 3 * sizeof(double));
  }
  H->n_lights = (int32_t)D.n_lights;
  if (pb.ops.empty()) {  // empty world: one group whose rejected branch ends the program
    pb.ops.push_back({OP_GROUP, 0, 1, 0, -1, {0, 0, 0}});
    pb.group_box.insert(pb.group_box.end(), {1, 1, 1, 0, 0, 0});
    pb.group_parent.push_back(-1);
  }
  lap("array copies");
  H->ops = std::move(pb.ops);
  H->group_box = std::move(pb.group_box);
  H->group_parent = std::move(pb.group_parent);
  H->bvh = std::move(pb.bvh_nodes);
  H->mtri = std::move(pb.mtri);
  H->mtri_prim = std::move(pb.mtri_prim);
  H->items = std::move(pb.items);
  H->n_plain_items = pb.n_plain_items;
  if (pb.have_abvh && pb.cube_pad > 0.0) {
    const double reach = pb.cube_pad / 0.00001;   // |t| below which a parallel-axis point stays inside the pad
    H->quirk_reach2 = reach * reach / 48.0;       // the ray-side test is 4 sqrt(3) m / |d| <= reach (rtc_device.hpp)
    std::memcpy(H->abvh_frame, pb.abvh_frame, sizeof(H->abvh_frame));
    if (timing) std::fprintf(stderr, "[rtc-timing]   cube pad %.6f: rays with |o - (%.2f, %.2f, %.2f)|_inf + %.2f <= %.1f |d| skip the cubes' quirk scan\n", pb.cube_pad,
                             pb.abvh_frame[0], pb.abvh_frame[1], pb.abvh_frame[2], pb.abvh_frame[3], std::sqrt(H->quirk_reach2));
  }
  H->bvh_prims = std::move(pb.bvh_prims);
  H->qgrids = std::move(pb.qgrids);
  H->qcell = std::move(pb.qcell);
  H->bvh_frame = std::move(pb.bvh_frame);
  H->csg = std::move(pb.csg);
  H->bvh_depth = pb.max_depth;
  H->csg_max_hits = pb.csg_max_hits;
  H->built_on_device = pb.built_on_device;
  H->bvh_stack = std::max(8, pb.max_stack + 1);
  lap("moves");
  {  // planes whose record travels in the kernel arguments: the primitive names its slot (its world normal is read there, DPrim.pad)
    const DScene dv = H->view();
    lap("view()");
    for (int i = 0; i < dv.n_kplanes; i++) H->prims[(size_t)dv.kplanes[i].prim].pad[0] = i + 1;
    // the 128-byte intersection records, one per primitive -- only if some op of the program reads them (DScene.has_recs): a mesh
    // and a few planes in the kernel arguments do not, and 10^6 triangles would carry 128 MB of them to the device for nothing
    if (dv.has_recs) {
      // ... and only up to the last primitive an op can name: the program's own OP_PRIMs (CSG sub-programs included) and the analytic
      // BVH's primitives.  Mesh triangles are reached through their BVH's packed triangle array, never through a record; an OBJ
      // group at the end of the world -- where the bins put it -- leaves the array a few entries long.
      int64_t last = -1;
      for (const DOp& o : H->ops) if (o.op == OP_PRIM) last = std::max<int64_t>(last, o.a);
      for (int32_t pi : H->bvh_prims) last = std::max<int64_t>(last, pi);
      const size_t n_rec = (size_t)(last + 1);
      H->pisect.resize(n_rec);
      parallel_for(n_rec, 65536, [&](size_t b, size_t e) {
        for (size_t i = b; i < e; i++) {
          const DPrim& P = H->prims[i];
          DPrimI& q = H->pisect[i];
          q.geom = P.geom; q.flags = P.flags; q.data = P.data; q.gcond = P.gcond;
          const bool lim = P.geom == RTC_CYLINDER || P.geom == RTC_CONE;
          q.mn = lim ? D.limits[2 * (size_t)P.data] : 0.0;
          q.mx = lim ? D.limits[2 * (size_t)P.data + 1] : 0.0;
          std::memcpy(q.m, &H->xf_inv[(size_t)P.xform * 12], 12 * sizeof(double));
        }
      });
    }
    if (timing) std::fprintf(stderr, "[rtc-timing]   %zu intersection records for %u primitives\n", H->pisect.size(), D.n_prims);
    lap("intersection records");
  }
  return RTC_OK;
}

}  // namespace rtb
