// rtc_device.hpp — device functions and the templated ray kernels of the hot path (included by rtc_feat.hip, which
// instantiates the kernels of ONE kernel variant per translation unit, and by rtc_kernels.hip).
#pragma once

// Hand-written HIP for gfx950 (MI355X): the reference's per-pixel path
//   Image::par_render -> Camera::ray_at_pixel -> World::color_at -> {intersect, sort, hit, prepare_state,
//   shade_hit -> {is_shadowed, Shape::lighting -> Pattern::color_at -> noise, reflected_color, refracted_color}}
// (src/image.rs:65-81, src/camera.rs:39-55, src/world.rs:18-149, src/intersection.rs:24-139,
//  src/shape.rs:414-462 + :592-946, src/bounding_box.rs:80-92, src/material.rs:164-302, src/noise.rs:31-237).
//
// All arithmetic is IEEE f64, compiled with -ffp-contract=off, in the reference's operation order wherever a
// value can decide a hit.  What is *not* the reference's data flow:
//   * no intersection lists, no sort: the nearest hit is the minimum of (t, primitive sequence, push index)
//     over t >= 0 — exactly what "stable sort by t, first t >= 0" selects (src/intersection.rs:123-132);
//   * n1/n2 (src/intersection.rs:70-103) come from one storage-free pass over the same ray: a shape is in the
//     container list iff it has an odd number of intersections before the hit, and lists are ordered by each
//     shape's last such intersection;
//   * recursion (src/world.rs:84-132) is a per-lane stack of pending rays with scalar path weights; the
//     reference's once-per-light re-tracing of identical subtrees (src/world.rs:58-79) becomes a factor L.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "device_scene.h"

#define EPS 0.00001
#ifndef RTC_BLOCK
#define RTC_BLOCK 64
#endif
#ifndef RTC_WAVES_PER_SIMD
#define RTC_WAVES_PER_SIMD 1
#endif
#ifdef RTC_TRAVERSE_NOINLINE
#define TRAVERSE_INLINE __noinline__
#else
#define TRAVERSE_INLINE __forceinline__
#endif
// Launch-geometry hooks.  A build may pre-define any of them (the CPU emulation of this source under tests/cpu_emu does, in its
// hip_runtime.h stand-in: one-lane or thread-per-lane "waves", static arrays for LDS); what follows is the device's.
#ifndef RTC_LANE_ID
#define RTC_LANE_ID ((int)(threadIdx.x & 63u))
#endif
// wf_shade reserves queue space once per block and iteration (same-address device atomics serialise: one per wave cost 0.4 ms per level)
#ifndef RTC_WF_SHADE_BLOCK
#define RTC_WF_SHADE_BLOCK 512  // measured: 128 -> +11 % frame time (more same-address atomics), 256 and 512 equal
#endif
#ifndef RTC_WF_SHADE_GRID_DIV
#define RTC_WF_SHADE_GRID_DIV 2u
#endif
// per-lane BVH stacks live in LDS, sized per scene at launch (DScene.bvh_stack entries per lane)
#ifndef RTC_LDS_STACK
#define RTC_LDS_STACK(name) extern __shared__ int name[]
#endif
#ifndef RTC_WF_LANES
#define RTC_WF_LANES 64u
#endif
#define DINF (__builtin_inf())
#ifndef RTC_PROBE
#define RTC_PROBE 0  // cost probes (scripts/build_variant.sh ... -DRTC_PROBE=<bits>): a section runs twice, pixels unchanged. 1 planes, 2 quirk grid, 4 node steps, 8 Phong, 16 analytic leaf tests, 32 container passes (wavefront path)
#endif
// RTC_LAUNDER(x): the compiler may not assume it knows x's value any more.  Used on a work item's index right after a traversal:
// without it every output address derived from the index before the traversal (hipcc computes them all up front) stays live
// across it, and at a 128-register budget that meant ~60 dwords spilled to scratch and reloaded per ray.
#ifndef RTC_LAUNDER
#define RTC_LAUNDER(x) asm volatile("" : "+v"(x))
#endif
// 1 / x to ~2^-24 relative (V_RCP_F64; see approx_rcp below)
#ifndef RTC_APPROX_RCP
#define RTC_APPROX_RCP(x) __builtin_amdgcn_rcp(x)
#endif
static inline unsigned rtc_stack_bytes(const DScene& S) { return (unsigned)S.bvh_stack * RTC_BLOCK * (unsigned)sizeof(int); }

namespace {

struct Ray { double ox, oy, oz, dx, dy, dz; };

enum { MODE_CLOSEST = 0, MODE_SHADOW_ANY = 1, MODE_SHADOW_CLOSEST = 2, MODE_CONTAINERS = 3 };

// Per-ray traversal state (registers).  One layout serves all four passes:
//   CLOSEST / SHADOW_CLOSEST: best_* = running minimum of (t, prim, push) over t >= 0; thi follows best_t.
//   SHADOW_ANY: thi = distance to the light; `shadowed` set by any intersection with 0 <= t < thi.
//   CONTAINERS: (thi, best_prim, best_klast) = key of the hit (left over from the closest pass);
//               c1/c2 = odd-count shape with the largest (t, prim) before / up to that key (-1 = none).
struct Trav {
  int mode;
  double tlo, thi;  // interval of t the pass cares about (also the accelerator's culling interval)
  double best_t;
  int best_prim, best_k, best_klast;
  int shadowed;
  int unordered;  // any-hit pass that skips the nearest-first ordering of a node's children
  int flags;  // bit 0 (TF_CUBES_IN_LEAF): this ray's reach is short enough for the padded cube boxes (DScene.quirk_reach2): BVH leaves
              //   test cubes for quirk rays too and the cubes' quirk scan is skipped; set by traverse()
              // bit 1 (TF_NAN): a NaN t was pushed in this pass; bits 8..: intersections pushed in this pass (nan_commit())
  int light;      // shadow passes: index of the light the ray runs towards (its light grid may stand in for the BVH walk), the ray's
                  // length in c1_t; else -1
  double c1_t, c2_t;
  int c1_prim, c2_prim;
  // per-ray cache of the reference's group box tests (groups 0..63): bit set in g_known once evaluated, in g_pass if it hit
  unsigned long long g_known, g_pass;
};

enum { TF_CUBES_IN_LEAF = 1, TF_NAN = 2, TF_PUSH_SHIFT = 8 };

struct Counters {
  unsigned int accel_nodes, group_tests, tri_tests, analytic_tests, nan_ts;
  unsigned int knodes, kplanes;  // of accel_nodes / analytic_tests: records that came from the kernel arguments (no memory traffic)
  unsigned int light_cells;      // light-grid cells looked up (each in place of a BVH walk)
  unsigned int kgroups;          // of group_tests: the gate's group came from a kernel-argument op (wave-uniform box address)
};

// ---- diagnostics (RTC_DIAG builds only; the shipped kernel compiles these to nothing) ----------------------------------
// diag[2r] += cycles a lane spent in region r (every participating lane measures the wave's wall time of the region),
// diag[2r+1] += 1 per participation; diag[16+2j] += active lanes, diag[16+2j+1] += 1 per executed iteration of loop j.
#ifdef RTC_DIAG
// accumulated per block in LDS (one wave per block: no cross-wave contention in the hot loops), flushed once at kernel end
__shared__ unsigned long long s_diag[64];
#define DIAG_T0() diag_t0_ = __builtin_amdgcn_s_memtime()
#define DIAG_REGION(r) do { unsigned long long t1_ = __builtin_amdgcn_s_memtime(); atomicAdd(&s_diag[2 * (r)], t1_ - diag_t0_); atomicAdd(&s_diag[2 * (r) + 1], 1ull); diag_t0_ = t1_; } while (0)
#define DIAG_SPAN_BEGIN() unsigned long long span_t0_ = __builtin_amdgcn_s_memtime()
#define DIAG_SPAN_END(r) do { atomicAdd(&s_diag[2 * (r)], __builtin_amdgcn_s_memtime() - span_t0_); atomicAdd(&s_diag[2 * (r) + 1], 1ull); } while (0)
#define DIAG_LOOP(j) do { unsigned long long m_ = __ballot(1); if ((int)(threadIdx.x & 63) == __ffsll((long long)m_) - 1) { atomicAdd(&s_diag[16 + 2 * (j)], (unsigned long long)__popcll(m_)); atomicAdd(&s_diag[16 + 2 * (j) + 1], 1ull); } } while (0)
#else
#define DIAG_T0() do {} while (0)
#define DIAG_REGION(r) do {} while (0)
#define DIAG_SPAN_BEGIN() do {} while (0)
#define DIAG_SPAN_END(r) do {} while (0)
#define DIAG_LOOP(j) do {} while (0)
#endif

// The specular factor of Shape::lighting (src/shape.rs:453): intensity * specular * powf(r.e, shininess).  When the material's specular is
// (+-)0 the product is the same (+-)0 — or the same NaN for a non-finite intensity — for EVERY finite factor, so the library call (~150
// instructions; every hit on a matte floor or wall makes it, per light) is skipped where the factor is certainly finite: 0 < r.e <= 1 + 1e-7
// and 0 <= shininess <= 1e6 (then r.e^shininess <= e^0.1).  Anything else takes the real powf.
__device__ double rtc_pow(double a, double b);
__device__ __forceinline__ double specular_factor(double rde, double shininess, double specular) {
  if (specular == 0.0 && shininess >= 0.0 && shininess <= 1.0e6 && rde <= 1.0000001) return 1.0;
  return rtc_pow(rde, shininess);
}

// f64::powf (src/shape.rs:453).  Out of line: inlined, the library routine's ~50 f64 polynomial constants are hoisted to the top of
// the calling kernel and spilled to scratch from there (wf_ts: 2/3 of its spill traffic), for a call that only lit highlights make.
__device__ __noinline__ double rtc_pow(double a, double b) { return pow(a, b); }

// Rust f64::max/min: a NaN operand is ignored.
__device__ __forceinline__ double rmax(double a, double b) { return (a != a) ? b : ((b != b) ? a : (a > b ? a : b)); }
__device__ __forceinline__ double rmin(double a, double b) { return (a != a) ? b : ((b != b) ? a : (a < b ? a : b)); }

// src/shape.rs:635-653
__device__ __forceinline__ void cube_axis(double origin, double direction, double mn, double mx, double& tmin, double& tmax) {
  double a_num = mn - origin, b_num = mx - origin;
  double a, b;
  if (fabs(direction) >= EPS) {
    a = a_num / direction;
    b = b_num / direction;
  } else {
    a = a_num * DINF;
    b = b_num * DINF;
  }
  if (a > b) { tmin = b; tmax = a; } else { tmin = a; tmax = b; }
}

// 1 / x to ~2^-24 relative (V_RCP_F64: one quarter-rate instruction instead of the ~15 of a correctly rounded division); only ever
// used where a decision tolerates far more than that and the exact expression is evaluated otherwise.
__device__ __forceinline__ double approx_rcp(double x) { return RTC_APPROX_RCP(x); }

// src/bounding_box.rs:80-92 on the group's f64 box.  The answer is a comparison of six quotients; most rays miss or cross a group's box
// by a wide margin, so approximate quotients (reciprocals to 2^-24, margin 2^-20 of the largest quotient) settle those rays and the
// reference's expression — correctly rounded divisions, its |d| < EPSILON and NaN rules — only runs for the ones in between and for
// rays with a direction component below 2 EPSILON or a non-finite operand.
__device__ __forceinline__ bool group_box_hit(const double* __restrict__ b, const Ray& r) {
#ifndef RTC_NO_GROUP_PRETEST
  if (fabs(r.dx) >= 2.0 * EPS && fabs(r.dy) >= 2.0 * EPS && fabs(r.dz) >= 2.0 * EPS) {
    const double rx = approx_rcp(r.dx), ry = approx_rcp(r.dy), rz = approx_rcp(r.dz);
    const double a0 = (b[0] - r.ox) * rx, b0 = (b[3] - r.ox) * rx;
    const double a1 = (b[1] - r.oy) * ry, b1 = (b[4] - r.oy) * ry;
    const double a2 = (b[2] - r.oz) * rz, b2 = (b[5] - r.oz) * rz;
    const double big = fabs(a0) + fabs(b0) + fabs(a1) + fabs(b1) + fabs(a2) + fabs(b2);  // NaN / inf anywhere -> not < 1e300
    if (big < 1e300) {
      const double t0 = fmax(fmax(fmin(a0, b0), fmin(a1, b1)), fmin(a2, b2));
      const double t1 = fmin(fmin(fmax(a0, b0), fmax(a1, b1)), fmax(a2, b2));
      const double m = big * 9.5367431640625e-07;
      if (t0 > t1 + m) return false;
      if (t0 < t1 - m) return true;
    }
  }
#endif
  double xa, xb, ya, yb, za, zb;
  cube_axis(r.ox, r.dx, b[0], b[3], xa, xb);
  cube_axis(r.oy, r.dy, b[1], b[4], ya, yb);
  cube_axis(r.oz, r.dz, b[2], b[5], za, zb);
  double t_min = rmax(rmax(xa, ya), za);
  double t_max = rmin(rmin(xb, yb), zb);
  return t_min <= t_max;
}

// Group::intersect's gate (src/shape.rs:251): a primitive (mesh, CSG node) under aggregation groups is reached only if
// BoundingBox::intersects passes for every ancestor.  g = innermost group; walks group_parent; results cached per ray.
template <bool CACHE = true>
__device__ __forceinline__ bool groups_pass(const DScene& S, int g, const Ray& r, Trav& T, Counters& C) {
  while (g >= 0) {
    bool hit;
    if (CACHE && g < 64) {
      const unsigned long long bit = 1ull << g;
      if (T.g_known & bit) hit = (T.g_pass & bit) != 0ull;
      else {
        C.group_tests++;
        hit = group_box_hit(S.group_box + 6 * g, r);
        T.g_known |= bit;
        if (hit) T.g_pass |= bit;
      }
    } else {
      C.group_tests++;
      hit = group_box_hit(S.group_box + 6 * g, r);
    }
    if (!hit) return false;
    g = S.group_parent[g];
  }
  return true;
}

// The gate of an op of the kernel-argument program: the group index is the same for every lane, so the chain's boxes are read
// with scalar loads of one address each (counted apart: they move no bytes through the vector memory system).
__device__ __forceinline__ bool kops_gate(const DScene& S, int g, const Ray& r, Trav& T, Counters& C) {
  const unsigned before = C.group_tests;
  const bool pass = groups_pass<false>(S, g, r, T, C);
  C.kgroups += C.group_tests - before;
  return pass;
}

// Ray::transform (src/ray.rs:14-19) with Matrix*Vector (src/linalg/matrix.rs:261-284); origin.w = 1, direction.w = 0.
__device__ __forceinline__ Ray to_object(const double* __restrict__ m, const Ray& r) {
  Ray o;
  o.ox = m[0] * r.ox + m[1] * r.oy + m[2] * r.oz + m[3] * 1.0;
  o.oy = m[4] * r.ox + m[5] * r.oy + m[6] * r.oz + m[7] * 1.0;
  o.oz = m[8] * r.ox + m[9] * r.oy + m[10] * r.oz + m[11] * 1.0;
  o.dx = m[0] * r.dx + m[1] * r.dy + m[2] * r.dz + m[3] * 0.0;
  o.dy = m[4] * r.dx + m[5] * r.dy + m[6] * r.dz + m[7] * 0.0;
  o.dz = m[8] * r.dx + m[9] * r.dy + m[10] * r.dz + m[11] * 0.0;
  return o;
}

// ---- feeding intersections of ONE primitive (in push order) to the current pass ------------------------
__device__ __forceinline__ bool key_before(double t, int prim, int k, double ht, int hprim, int hk) {
  return (t < ht) || (t == ht && (prim < hprim || (prim == hprim && k < hk)));
}

// The reference sorts the list of a closest-hit or shadow pass with `partial_cmp().unwrap()` (src/intersection.rs:123-125): a NaN t
// panics when the comparator sees it, which is when the list holds at least two entries (a slice of 0 or 1 is returned as it is).
// A NaN t comes from a ray with a NaN in it (e.g. the zero-length normal at a cone's apex), and such a ray takes every box of the
// accelerators, so the pushes counted here are the reference's list.  Called once after a closest / shadow pass.
__device__ __forceinline__ void nan_commit(const Trav& T, Counters& C) {
  if ((T.flags & TF_NAN) && (T.flags >> TF_PUSH_SHIFT) >= 2) C.nan_ts++;
}

// `ks` (optional) = the original push index of each entry, when a CSG filter has dropped some of a primitive's pushes.
__device__ __forceinline__ void accept(Trav& T, Counters& C, int prim, int n, const double* t, const int* ks = nullptr) {
  if (n == 0) return;
  T.flags += n << TF_PUSH_SHIFT;
  if (T.mode == MODE_CLOSEST || T.mode == MODE_SHADOW_CLOSEST) {
    for (int j = 0; j < n; j++) {
      const int k = ks ? ks[j] : j;
      double tk = t[j];
      if (tk != tk) T.flags |= TF_NAN;
      if (tk >= 0.0) {
        if (tk < T.best_t || (tk == T.best_t && prim < T.best_prim)) {
          T.best_t = tk; T.best_prim = prim; T.best_k = k; T.best_klast = k;
          T.thi = tk;
        } else if (tk == T.best_t && prim == T.best_prim) {
          T.best_klast = k;
        }
      }
    }
  } else if (T.mode == MODE_SHADOW_ANY) {
    for (int k = 0; k < n; k++) {
      double tk = t[k];
      if (tk != tk) T.flags |= TF_NAN;
      if (tk >= 0.0 && tk < T.thi) T.shadowed = 1;
    }
  } else {  // MODE_CONTAINERS
    int cnt1 = 0, cnt2 = 0;
    double m1 = 0.0, m2 = 0.0;
    for (int j = 0; j < n; j++) {
      const int k = ks ? ks[j] : j;
      double tk = t[j];
      bool b1 = key_before(tk, prim, k, T.thi, T.best_prim, T.best_klast);
      bool b2 = b1 || (tk == T.thi && prim == T.best_prim && k == T.best_klast);
      // pushes of one primitive arrive in push order, so ">=" keeps the latest push among equal t
      if (b1) { if (cnt1 == 0 || tk >= m1) m1 = tk; cnt1++; }
      if (b2) { if (cnt2 == 0 || tk >= m2) m2 = tk; cnt2++; }
    }
    if ((cnt1 & 1) && (T.c1_prim < 0 || m1 > T.c1_t || (m1 == T.c1_t && prim > T.c1_prim))) { T.c1_t = m1; T.c1_prim = prim; }
    if ((cnt2 & 1) && (T.c2_prim < 0 || m2 > T.c2_t || (m2 == T.c2_t && prim > T.c2_prim))) { T.c2_t = m2; T.c2_prim = prim; }
  }
}

// Plane (src/shape.rs:621-633), t = -oy / dy: true when the quotient is certainly a normal negative number (operands of opposite
// sign, no underflow to -0, which would pass `t >= 0`), so that no pass but the container pass (which counts intersections
// behind the origin too) can use it: about half of all plane tests end here, without the f64 division and the hit bookkeeping.
__device__ __forceinline__ bool plane_behind(const Trav& T, double oy, double dy) {
  return T.mode != MODE_CONTAINERS && ((oy > 0.0) != (dy < 0.0)) && fabs(oy) > 1e-290 && fabs(dy) < 1e17;
}

// Geometry::intersect_triangle (src/shape.rs:824-860); o = object-space ray
__device__ __forceinline__ int tri_hit(const double* __restrict__ g, const Ray& o, double& t, double& u, double& v) {
  double p1x = g[0], p1y = g[1], p1z = g[2], e1x = g[3], e1y = g[4], e1z = g[5], e2x = g[6], e2y = g[7], e2z = g[8];
  double cx = o.dy * e2z - o.dz * e2y, cy = o.dz * e2x - o.dx * e2z, cz = o.dx * e2y - o.dy * e2x;  // dir x e2
  double det = e1x * cx + e1y * cy + e1z * cz;
  if (fabs(det) < EPS) return 0;
  double f = 1.0 / det;
  double sx = o.ox - p1x, sy = o.oy - p1y, sz = o.oz - p1z;  // p1_to_origin
  u = f * (sx * cx + sy * cy + sz * cz);
  if (u < 0.0 || u > 1.0) return 0;
  double qx = sy * e1z - sz * e1y, qy = sz * e1x - sx * e1z, qz = sx * e1y - sy * e1x;  // origin x e1
  v = f * (o.dx * qx + o.dy * qy + o.dz * qz);
  if (v < 0.0 || u + v > 1.0) return 0;
  t = f * (e2x * qx + e2y * qy + e2z * qz);
  return 1;
}

// Geometry::intersect (src/shape.rs:862-885) for one primitive; returns the number of pushes, in push order.
__device__ __forceinline__ int prim_hits(const DScene& S, const DPrimI& P, const Ray& o, double* t, double& u, double& v) {
  int n = 0;
  if (P.geom == 1) {  // plane :621-633
    if (fabs(o.dy - 0.0) < EPS) return 0;
    t[0] = -o.oy / o.dy;
    return 1;
  }
  if (P.geom == 2) {  // cube :655-679
    DIAG_LOOP(14);
    double xa, xb, ya, yb, za, zb;
    cube_axis(o.ox, o.dx, -1.0, 1.0, xa, xb);
    cube_axis(o.oy, o.dy, -1.0, 1.0, ya, yb);
    cube_axis(o.oz, o.dz, -1.0, 1.0, za, zb);
    double t_min = rmax(rmax(xa, ya), za);
    double t_max = rmin(rmin(xb, yb), zb);
    if (t_min <= t_max) { t[0] = t_min; t[1] = t_max; return 2; }
    return 0;
  }
  if (P.geom >= 5) return tri_hit(S.tri_geo + 9 * P.data, o, t[0], u, v);  // triangles :824-860
  // Sphere :592-619, cylinder :724-768, cone :770-822: one quadratic a t^2 + b t + c = 0 for the three of them, so that lanes
  // holding different kinds share the square root and the two divisions (a wave's leaf tests are a mix of kinds; each kind alone
  // ran at ~20 % of the lanes).  The coefficients keep each kind's own operation order, bit for bit:
  //   sphere    a = (dx dx + dy dy) + dz dz    b = 2 ((dx ox + dy oy) + dz oz)              c = ((ox ox + oy oy) + oz oz) - 1
  //   cylinder  a =  dx dx + dz dz             b = 2 ox dx + 2 oz dz  = 2 (ox dx + oz dz)   c =  (ox ox + oz oz) - 1
  //   cone      a = (dx dx - dy dy) + dz dz    b = (2 ox dx - 2 oy dy) + 2 oz dz = 2 (...)  c =  (ox ox - oy oy) + oz oz
  // (x + (-0.0) = x and x - 0.0 = x for every x, a square is never -0.0, and scaling by 2 commutes with rounding, so the shared
  // expressions below give the same bits -- the sign of a zero b included: a ray that starts ON a cylinder with -0.0 direction
  // components has b = -0.0 and t = +0.0 in the reference; `+ 0.0` here made that b = +0.0 and t = -0.0 until the special-point
  // rays of tests/cases.py found it.)
  DIAG_LOOP(13);
  const bool sph = P.geom == 0, cone = P.geom == 4;
  const double mn = P.mn, mx = P.mx;
  const double ydd = o.dy * o.dy, ydo = o.dy * o.oy, yoo = o.oy * o.oy;
  const double a = (o.dx * o.dx + (sph ? ydd : (cone ? -ydd : 0.0))) + o.dz * o.dz;
  const double b = 2.0 * ((o.dx * o.ox + (sph ? ydo : (cone ? -ydo : -0.0))) + o.dz * o.oz);  // -0.0: x + (-0.0) = x for x = -0.0 too
  const double c = ((o.ox * o.ox + (sph ? yoo : (cone ? -yoo : 0.0))) + o.oz * o.oz) - (cone ? 0.0 : 1.0);
  const bool a0 = fabs(a - 0.0) < EPS;
  if (cone && a0 && !(fabs(b - 0.0) < EPS)) t[n++] = -c / (2.0 * b);  // single-root branch :812-818
  const double disc = b * b - 4.0 * a * c;
  // sphere: `if disc < 0 { return }` lets a NaN discriminant through; cylinder / cone: `if disc >= 0` does not
  if (sph ? !(disc < 0.0) : (!a0 && disc >= 0.0)) {
    const double sq = sqrt(disc);
    const double t0 = (-b - sq) / (2.0 * a);
    const double t1 = (-b + sq) / (2.0 * a);
    if (sph) { t[0] = t0; t[1] = t1; return 2; }
    const double y0 = o.oy + t0 * o.dy;
    if (mn < y0 && y0 < mx) t[n++] = t0;
    const double y1 = o.oy + t1 * o.dy;
    if (mn < y1 && y1 < mx) t[n++] = t1;
  }
  if (sph) return 0;
  // intersect_cap :681-722 (cylinder radii 1,1; cone radii min,max)
  if ((P.flags & 2u) && !(fabs(o.dy - 0.0) < EPS)) {
    DIAG_LOOP(15);
    double r0 = cone ? mn : 1.0, r1 = cone ? mx : 1.0;
    double tc = (mn - o.oy) / o.dy;
    double x = o.ox + tc * o.dx, z = o.oz + tc * o.dz;
    if ((x * x + z * z) <= r0 * r0) t[n++] = tc;
    tc = (mx - o.oy) / o.dy;
    x = o.ox + tc * o.dx; z = o.oz + tc * o.dz;
    if ((x * x + z * z) <= r1 * r1) t[n++] = tc;
  }
  return n;
}

// Block-local copy of the scene's hot tables (kernels instantiated with LDSC = true: scenes whose accelerator and intersection
// records fit a CU's 160 KB of LDS beside the traversal stacks; the block copies them in once and every dependent node / record
// fetch of a walk is then an LDS read of ~100 cycles instead of an L1 / L2 round trip).  Tables are stored row by row (row r of
// element n at [r * count + n], 16 B each) so that lanes reading the same row of different elements spread over the banks.
struct LdsScene {
  const float4* nodes;   // 7 rows: lox loy loz hix hiy hiz child refs
  const float4* recs;    // 8 rows: {geom, flags, data, gcond} {mn, mx} m[0..11]   (scenes whose program reads records)
  const double* tris;    // 9 rows: p1 e1 e2 xyz of the leaf-order triangles        (scenes with a mesh BVH)
  const int32_t* tri_prim;
  int n_nodes, n_recs, n_tris;
};
__device__ __forceinline__ int4 as_int4(const float4 v) { int4 r; __builtin_memcpy(&r, &v, 16); return r; }
__device__ __forceinline__ DPrimI lds_record(const LdsScene& L, int prim) {
  DPrimI P;
  float4 rows[8];
#pragma unroll
  for (int r = 0; r < 8; r++) rows[r] = L.recs[r * L.n_recs + prim];
  __builtin_memcpy(&P, rows, sizeof(P));
  return P;
}
// bytes of the tables (host + device agree on the layout: nodes, records, triangles, triangle -> primitive, 16-byte aligned pieces)
static inline unsigned long long rtc_lds_table_bytes(const DScene& S) {
  unsigned long long b = 7ull * 16 * (unsigned)S.n_bvh;
  if (S.has_recs) b += 8ull * 16 * (unsigned)S.n_recs;
  if (S.has_mesh) b += (72ull * (unsigned)S.n_mtri + 4ull * (unsigned)S.n_mtri + 15ull) & ~15ull;
  return b;
}
#ifndef RTC_EMU
// The block copies the tables from memory into dynamic LDS at `base`; returns the first byte after them (the traversal stacks).
__device__ __forceinline__ char* lds_fill(const DScene& S, char* base, LdsScene& L) {
  float4* ln = (float4*)base;
  const float4* gn = (const float4*)S.bvh;
  for (int k = (int)threadIdx.x; k < 8 * S.n_bvh; k += (int)blockDim.x) { const int n = k >> 3, r = k & 7; if (r < 7) ln[r * S.n_bvh + n] = gn[k]; }
  char* p = (char*)(ln + 7 * (size_t)S.n_bvh);
  L.nodes = ln; L.n_nodes = S.n_bvh;
  L.recs = nullptr; L.n_recs = 0; L.tris = nullptr; L.tri_prim = nullptr; L.n_tris = 0;
  if (S.has_recs) {
    float4* lr = (float4*)p;
    const float4* gr = (const float4*)S.pisect;
    for (int k = (int)threadIdx.x; k < 8 * S.n_recs; k += (int)blockDim.x) { const int n = k >> 3, r = k & 7; lr[r * S.n_recs + n] = gr[k]; }
    L.recs = lr; L.n_recs = S.n_recs;
    p = (char*)(lr + 8 * (size_t)S.n_recs);
  }
  if (S.has_mesh) {
    double* lt = (double*)p;
    for (int k = (int)threadIdx.x; k < 9 * S.n_mtri; k += (int)blockDim.x) { const int n = k / 9, c = k - 9 * n; lt[c * S.n_mtri + n] = S.mtri[k]; }
    int32_t* lp = (int32_t*)(lt + 9 * (size_t)S.n_mtri);
    for (int k = (int)threadIdx.x; k < S.n_mtri; k += (int)blockDim.x) lp[k] = S.mtri_prim[k];
    L.tris = lt; L.tri_prim = lp; L.n_tris = S.n_mtri;
    p += (72ull * (unsigned)S.n_mtri + 4ull * (unsigned)S.n_mtri + 15ull) & ~15ull;
  }
  __syncthreads();
  return p;
}
#endif

// Shape::intersect (src/shape.rs:414-417) for one primitive.
// Two reference quirks let a primitive report an intersection OUTSIDE any finite bound of its surface, so a
// bounding-volume hierarchy alone would lose them:
//   cube: an axis with |direction| < EPSILON only requires the ORIGIN to be inside the slab (src/shape.rs:641-648);
//   cone: when a ~ 0 the single root -c/(2b) is pushed without the min < y < max check (src/shape.rs:812-818).
// Rays in that state ("quirk rays" for this primitive) are tested by the linear OP_QUIRK pass and skipped in
// the BVH leaf; all other rays are tested in the leaf only.  policy: 0 always, 1 skip if quirk, 2 only if quirk.
// FEAT: feature level of the kernel instantiation — 0: no groups, no CSG in the scene (no gate code at all); 1: only whole
// meshes are gated (one uncached chain walk per OP_MESH: the teapot scenes); 2: per-primitive gates with the per-ray cache;
// 3: + CSG.  Keeps the common kernels under the register cliff.
template <int FEAT, bool LDSC = false>
__device__ __forceinline__ void visit_prim(const DScene& S, int prim, const Ray& r, Trav& T, Counters& C, int policy, const LdsScene& L = LdsScene{}) {
  const DPrimI P = LDSC ? lds_record(L, prim) : S.pisect[prim];
  if (FEAT >= 2 && P.gcond >= 0 && !groups_pass(S, P.gcond, r, T, C)) return;
  const double* __restrict__ m = P.m;
  if (policy == 0) { DIAG_LOOP(9); } else if (policy == 1) { DIAG_LOOP(10); } else { DIAG_LOOP(11); }
  if (P.geom == 1) {
    // Plane (src/shape.rs:621-633) only reads origin.y and direction.y of the object-space ray: evaluate that one row of
    // Ray::transform, in the same order, and skip the other two.
    C.analytic_tests++;
    double oy = m[4] * r.ox + m[5] * r.oy + m[6] * r.oz + m[7] * 1.0;
    double dy = m[4] * r.dx + m[5] * r.dy + m[6] * r.dz + m[7] * 0.0;
    if (fabs(dy - 0.0) < EPS || plane_behind(T, oy, dy)) return;
    double t = -oy / dy;
    accept(T, C, prim, 1, &t);
    return;
  }
  if (policy == 2) {
    // quirk scan: the condition needs the object-space DIRECTION only; most candidates are rejected here
    double dx = m[0] * r.dx + m[1] * r.dy + m[2] * r.dz + m[3] * 0.0;
    double dy = m[4] * r.dx + m[5] * r.dy + m[6] * r.dz + m[7] * 0.0;
    double dz = m[8] * r.dx + m[9] * r.dy + m[10] * r.dz + m[11] * 0.0;
    bool quirk = false;
    if (P.geom == 2) quirk = fabs(dx) < EPS || fabs(dy) < EPS || fabs(dz) < EPS;
    else if (P.geom == 4) quirk = fabs((dx * dx - dy * dy + dz * dz) - 0.0) < EPS;
    if (!quirk) return;
    DIAG_LOOP(12);
  }
  Ray o = to_object(m, r);
  if (policy == 1) {
    bool quirk = false;
    if (P.geom == 2) quirk = !(T.flags & TF_CUBES_IN_LEAF) && (fabs(o.dx) < EPS || fabs(o.dy) < EPS || fabs(o.dz) < EPS);
    else if (P.geom == 4) quirk = fabs((o.dx * o.dx - o.dy * o.dy + o.dz * o.dz) - 0.0) < EPS;
    if (quirk) return;
  }
  double t[4], u = 0.0, v = 0.0;
  if (P.geom >= 5) C.tri_tests++; else C.analytic_tests++;
  int n = prim_hits(S, P, o, t, u, v);
  if (n > 0) { DIAG_LOOP(23); }
  accept(T, C, prim, n, t);
}

// Direction-grid culled quirk scan (device_scene.h OP_QGRID).  The cell lookup must mirror build_quirk_grid().
// The cell of a direction grid (quirk grids, light grids: cube map, n x n cells per face) a ray's direction falls into -> its item
// range [b, e) of qitem; false: the direction is too short or not finite for the builders' bounds (they assume length >=
// RTC_QGRID_MIN_LEN), the caller takes its fallback.  Must mirror build_quirk_grid() / build_light_grids().
__device__ __forceinline__ bool dir_grid_cell(const DScene& S, const DQuirkGrid G, const Ray& r, unsigned& b, unsigned& e) {
  double ax = fabs(r.dx), ay = fabs(r.dy), az = fabs(r.dz);
  double len2 = r.dx * r.dx + r.dy * r.dy + r.dz * r.dz;
  if (!(len2 >= RTC_QGRID_MIN_LEN * RTC_QGRID_MIN_LEN) || !(len2 < DINF)) return false;
  int face;
  double u, v;
  if (ax >= ay && ax >= az) { face = r.dx > 0.0 ? 0 : 1; u = r.dy / ax; v = r.dz / ax; }
  else if (ay >= az) { face = r.dy > 0.0 ? 2 : 3; u = r.dx / ay; v = r.dz / ay; }
  else { face = r.dz > 0.0 ? 4 : 5; u = r.dx / az; v = r.dy / az; }
  int iu = (int)((u + 1.0) * 0.5 * (double)G.n), iv = (int)((v + 1.0) * 0.5 * (double)G.n);
  iu = iu < 0 ? 0 : (iu >= G.n ? G.n - 1 : iu);
  iv = iv < 0 ? 0 : (iv >= G.n ? G.n - 1 : iv);
  int cell = G.cell_off + (face * G.n + iv) * G.n + iu;
  b = S.qcell[cell]; e = S.qcell[cell + 1];
  return true;
}

template <int FEAT, bool LDSC = false>
__device__ __forceinline__ void quirk_grid_scan(const DScene& S, const DQuirkGrid G, const Ray& r, Trav& T, Counters& C, const LdsScene& L = LdsScene{}) {
  DIAG_LOOP(21);
  unsigned b, e;
  if (!dir_grid_cell(S, G, r, b, e)) {
    for (int i = G.lin_first; i < G.lin_first + G.lin_count; i++) visit_prim<FEAT, LDSC>(S, S.quirk_prim[i], r, T, C, 2, L);
    return;
  }
  for (unsigned i = b; i < e; i++) visit_prim<FEAT, LDSC>(S, S.qitem[i], r, T, C, 2, L);
}

// Light grids (scene_build.hpp build_light_grids): a shadow ray towards light T.light takes the candidates of its direction's cell
// instead of walking the analytic BVH.  The lookup is two dependent fetches from memory (cell range, then items), so a traversal
// starts it before its first op (light_grid_fetch) and only collects the answer at the OP_BVH (light_grid_candidates): the plane
// tests in between hide the latency.
struct LightCell {
  int state;            // 0: no grid / not a shadow ray / direction not covered -> ordinary walk; 1: range and first items fetched
  unsigned b, e;        // item range of the cell (qitem, pairs)
  DLightItem i0, i1;    // its first two items (most cells hold no more)
};
#define RTC_WALK_END ((int)0x80000000)
template <int MODE>
__device__ __forceinline__ LightCell light_grid_fetch(const DScene& S, const Ray& r, const Trav& T) {
  LightCell c;
  c.state = 0; c.b = 0; c.e = 0;
  c.i0.ref = RTC_WALK_END; c.i0.dmin = 0.0f; c.i1 = c.i0;
  if (MODE == MODE_CLOSEST || MODE == MODE_CONTAINERS) return c;
  if (S.light_grid_first <= 0 || T.light < 0) return c;
  const DQuirkGrid G = {S.light_grid_n, S.light_grid_cell_off + T.light * (6 * S.light_grid_n * S.light_grid_n + 1), 0, 0};  // (no fetch of the grid's record)
  if (!dir_grid_cell(S, G, r, c.b, c.e)) return c;
  c.state = 1;
  if (c.b < c.e) c.i0 = *(const DLightItem*)(S.qitem + c.b);  // (8-byte aligned by the builder)
  if (c.b + 2 < c.e) c.i1 = *(const DLightItem*)(S.qitem + c.b + 2);
  return c;
}
// The cell's candidates — leaf references, as the BVH's own — go where the walk would have put them: the first becomes `cur`, the
// others wait on the stack.  Items: {reference, f32 lower bound of the candidate's distance from the light}, nearest first: a candidate
// farther from the light than the ray's origin (T.c1_t = the ray's length, slightly raised in f32) lies behind the origin, it and all
// after it are skipped.  true: (cur, sp) are set (cur = RTC_WALK_END: nothing to test); false: the ordinary walk.
__device__ __forceinline__ bool light_grid_candidates(const DScene& S, const LightCell& c, const Trav& T, Counters& C, int& cur, int& sp, int* __restrict__ stack, int stride) {
  if (c.state == 0) return false;
  const float reach = (float)T.c1_t * 1.000001f + 1e-30f;
  int first = RTC_WALK_END;
  int n = 0;
  if (c.b < c.e) {
    if (c.i0.ref == RTC_LIGHT_CELL_WALK) return false;
    if (c.i0.dmin <= reach) {
      first = c.i0.ref; n = 1;
      if (c.b + 2 < c.e && c.i1.dmin <= reach) {
        stack[0] = c.i1.ref; n = 2;
        for (unsigned i = c.b + 4; i < c.e; i += 2) {
          const DLightItem it = *(const DLightItem*)(S.qitem + i);
          if (!(it.dmin <= reach)) break;
          stack[(n - 1) * stride] = it.ref;
          n++;
        }
      }
    }
  }
  sp = n > 0 ? n - 1 : 0;
  C.light_cells++;
  DIAG_LOOP(22);
  cur = first;
  return true;
}

// ---- accelerator -------------------------------------------------------------------------------------
// ---- f32 slab test in the BVH's own frame (culling only) ------------------------------------------------
// Node boxes are f32, stored relative to the BVH's centre c and already widened at build time.  Per ray and BVH:
//   of  = fl32(o - c)                     (the f64 subtraction is exact to 2^-53, the rounding to f32 costs 2^-24 |o - c|)
//   eps = 2^-20 (|of|_inf + R)            R = radius of the BVH around c: >= 16x every f32 rounding error of the test,
//                                         each of which is bounded by 2^-23 times a coordinate or a distance travelled
//   every box is inflated by eps by testing its low faces against of + eps and its high faces against of - eps,
//   and the pass's t interval is widened by 2^-18 relative.  A face's distance is ONE fused multiply-add, face * i - (of +- eps) * i
//   with i = 1 / direction: the product is rounded once per ray (2^-24 |of * i|, an eighth of what eps moves the face by), the fma
//   once per face as a multiplication would be.  |i| is capped at 1e30 (a zero direction component: +-1e30 decides inside /
//   outside the slab like +-inf would, but inf - inf would be NaN); a product that still overflows makes a NaN, which the
//   min / max ignore: the box is taken.
// So a box is rejected only if the f64 ray misses the box inflated by ~15 eps or its [tn, tf] misses [t_lo, t_hi]; NaN
// (0 * inf) operands are ignored by fminf/fmaxf exactly as in the f64 version.  A ray whose origin does not fit f32
// relative to c takes every box (id = 0 makes every slab interval [0, 0]).
struct Frame32 {
  float olx, oly, olz, ohx, ohy, ohz, ix, iy, iz;  // ol / oh: (origin + eps) * i and (origin - eps) * i, see make_frame
  float px, py, pz, pm;  // container passes of the wavefront path: the hit point in the frame and its margin (make_frame_point)
};
// Container pass: a sphere or a cube reports 0 or 2 intersections, both inside its bounds, so it has an ODD number of
// them before the hit only if the hit point X = o + t_hit d lies between the two — inside its (convex, padded) bounds.  A leaf of such a
// primitive (bit 0 of its reference, set by the builder) is therefore taken only if X is inside the leaf's box, widened by pm; every
// other leaf (cylinders and cones can report 1 or 3: open ends, rims) and every inner node is taken when the LINE meets its box, as ever.
__device__ __forceinline__ void make_frame_point(const double* __restrict__ fr, const Ray& o, double t_hit, Frame32& f) {
  const float x = (float)((o.ox + t_hit * o.dx) - fr[0]), y = (float)((o.oy + t_hit * o.dy) - fr[1]), z = (float)((o.oz + t_hit * o.dz) - fr[2]);
  const float ox = (float)(o.ox - fr[0]), oy = (float)(o.oy - fr[1]), oz = (float)(o.oz - fr[2]);
  float m = fmaxf(fmaxf(fmaxf(fabsf(x), fabsf(y)), fabsf(z)), fmaxf(fmaxf(fabsf(ox), fabsf(oy)), fabsf(oz))) + (float)fr[3];
  f.px = x; f.py = y; f.pz = z;
  f.pm = m * 9.5367431640625e-07f + 1e-30f;  // 2^-20 (|X| + |o| + R), the scale of make_frame's eps
  if (!(m < 1e30f)) f.pm = __builtin_inff();  // no usable point: every box "contains" it (a NaN point fails every comparison below, so:)
  if (!(x == x) || !(y == y) || !(z == z)) { f.px = 0.0f; f.py = 0.0f; f.pz = 0.0f; f.pm = __builtin_inff(); }
}
__device__ __forceinline__ bool point_in_box(float lx, float ly, float lz, float hx, float hy, float hz, const Frame32& f) {
  return lx - f.pm <= f.px && f.px <= hx + f.pm && ly - f.pm <= f.py && f.py <= hy + f.pm && lz - f.pm <= f.pz && f.pz <= hz + f.pm;
}
__device__ __forceinline__ void make_frame(const double* __restrict__ fr, const Ray& o, Frame32& f) {
  float ox = (float)(o.ox - fr[0]), oy = (float)(o.oy - fr[1]), oz = (float)(o.oz - fr[2]);
  float m = fmaxf(fmaxf(fabsf(ox), fabsf(oy)), fabsf(oz)) + (float)fr[3];
  if (!(m < 1e30f)) {
    f.olx = f.oly = f.olz = f.ohx = f.ohy = f.ohz = 0.0f;
    f.ix = f.iy = f.iz = 0.0f;
    return;
  }
  float eps = m * 9.5367431640625e-07f + 1e-30f;
  f.olx = ox + eps; f.oly = oy + eps; f.olz = oz + eps;
  f.ohx = ox - eps; f.ohy = oy - eps; f.ohz = oz - eps;
  f.ix = 1.0f / (float)o.dx; f.iy = 1.0f / (float)o.dy; f.iz = 1.0f / (float)o.dz;
  f.ix = f.ix > 1e30f ? 1e30f : (f.ix < -1e30f ? -1e30f : f.ix);  // (a NaN stays a NaN: every box is taken, as before)
  f.iy = f.iy > 1e30f ? 1e30f : (f.iy < -1e30f ? -1e30f : f.iy);
  f.iz = f.iz > 1e30f ? 1e30f : (f.iz < -1e30f ? -1e30f : f.iz);
  f.olx *= f.ix; f.oly *= f.iy; f.olz *= f.iz;
  f.ohx *= f.ix; f.ohy *= f.iy; f.ohz *= f.iz;
}
__device__ __forceinline__ void t_interval32(const Trav& T, float& lo, float& hi) {
  const double SL = 3.814697265625e-06;  // 2^-18
  lo = (T.tlo == -DINF) ? -__builtin_inff() : (float)(T.tlo - SL * fmax(fabs(T.tlo), 1.0));
  hi = (T.thi == DINF) ? __builtin_inff() : (float)(T.thi + SL * fmax(fabs(T.thi), 1.0));
}
__device__ __forceinline__ float4 ld4(const float* p) { float4 v; __builtin_memcpy(&v, p, 16); return v; }
__device__ __forceinline__ int4 ld4(const int32_t* p) { int4 v; __builtin_memcpy(&v, p, 16); return v; }
__device__ __forceinline__ bool slab32c(float lx, float ly, float lz, float hx, float hy, float hz, const Frame32& f, float tlo, float thi, float& tn_out) {
  float t0 = __builtin_fmaf(lx, f.ix, -f.olx), t1 = __builtin_fmaf(hx, f.ix, -f.ohx);
  float tn = fminf(t0, t1), tf = fmaxf(t0, t1);
  t0 = __builtin_fmaf(ly, f.iy, -f.oly); t1 = __builtin_fmaf(hy, f.iy, -f.ohy);
  tn = fmaxf(tn, fminf(t0, t1)); tf = fminf(tf, fmaxf(t0, t1));
  t0 = __builtin_fmaf(lz, f.iz, -f.olz); t1 = __builtin_fmaf(hz, f.iz, -f.ohz);
  tn = fmaxf(tn, fminf(t0, t1)); tf = fminf(tf, fmaxf(t0, t1));
  tn_out = tn;
  return fmaxf(tn, tlo) <= fminf(tf, thi) && lx <= hx;
}

// One inner-node step of the walk: the node's four child boxes against the ray; hits ordered nearest first, the nearest
// becomes `cur`, the others are stacked; no hit pops (or ends the walk: cur = END).
// PM: container pass with a hit point in F (see make_frame_point); `analytic`: the node belongs to the analytic BVH, whose leaf references carry the
// "sphere or cube" bit in bit 0.  (A MESH leaf's low bits are its triangle count - 1: round 2 applied the point test to every leaf with bit 0 set, so
// a container pass skipped mesh leaves of 2 or 4 triangles unless the hit point lay in their box — and missed a triangle that the line crosses once
// before the hit, which the reference counts as a container.  Found by the hit-tree digest in a 400-seed fuzz run, seed 1058; one copy of the test
// for analytic walks only since.)
template <bool PM = false>
__device__ __forceinline__ void node_step(const float4 lox, const float4 loy, const float4 loz, const float4 hix, const float4 hiy, const float4 hiz, const int4 cc,
                                          const Frame32& F, float lo, float hi, int& cur, int& sp, int* __restrict__ stack, int stride, bool any_hit, bool analytic = true) {
  const float FINF = __builtin_inff();
  float t0, t1, t2, t3;
  bool h0 = slab32c(lox.x, loy.x, loz.x, hix.x, hiy.x, hiz.x, F, lo, hi, t0);
  bool h1 = slab32c(lox.y, loy.y, loz.y, hix.y, hiy.y, hiz.y, F, lo, hi, t1);
  bool h2 = slab32c(lox.z, loy.z, loz.z, hix.z, hiy.z, hiz.z, F, lo, hi, t2);
  bool h3 = slab32c(lox.w, loy.w, loz.w, hix.w, hiy.w, hiz.w, F, lo, hi, t3);
  if (PM && analytic) {
    if (cc.x < 0 && ((~cc.x) & 1)) h0 = h0 && point_in_box(lox.x, loy.x, loz.x, hix.x, hiy.x, hiz.x, F);
    if (cc.y < 0 && ((~cc.y) & 1)) h1 = h1 && point_in_box(lox.y, loy.y, loz.y, hix.y, hiy.y, hiz.y, F);
    if (cc.z < 0 && ((~cc.z) & 1)) h2 = h2 && point_in_box(lox.z, loy.z, loz.z, hix.z, hiy.z, hiz.z, F);
    if (cc.w < 0 && ((~cc.w) & 1)) h3 = h3 && point_in_box(lox.w, loy.w, loz.w, hix.w, hiy.w, hiz.w, F);
  }
  const int nh = (int)h0 + (int)h1 + (int)h2 + (int)h3;
  if (nh == 0) {
    if (sp == 0) cur = RTC_WALK_END;
    else { sp--; cur = stack[sp * stride]; }
    return;
  }
  if (any_hit) {
    // any-hit shadow pass (wave-uniform): the visiting order cannot change the answer, so no ordering work: the first hit
    // child is walked next, the other hit children are stacked as they come
    bool have = false;
    int nxt = 0;
    if (h0) { nxt = cc.x; have = true; }
    if (h1) { if (have) { stack[sp * stride] = cc.y; sp++; } else { nxt = cc.y; have = true; } }
    if (h2) { if (have) { stack[sp * stride] = cc.z; sp++; } else { nxt = cc.z; have = true; } }
    if (h3) { if (have) { stack[sp * stride] = cc.w; sp++; } else { nxt = cc.w; have = true; } }
    cur = nxt;
    return;
  }
  // order the (entry distance, child) pairs: a hit's key is finite (min with FLT_MAX also replaces a NaN), a miss's
  // is +inf, so after the network the first nh pairs are exactly the hits, nearest first
  const float FMAXV = 3.4028234663852886e38f;
  t0 = h0 ? fminf(t0, FMAXV) : FINF; t1 = h1 ? fminf(t1, FMAXV) : FINF; t2 = h2 ? fminf(t2, FMAXV) : FINF; t3 = h3 ? fminf(t3, FMAXV) : FINF;
  int c0 = cc.x, c1 = cc.y, c2 = cc.z, c3 = cc.w;
#define RTC_CSWAP(ta, ca, tb, cb) { const bool s_ = tb < ta; const float tt_ = s_ ? tb : ta; const int ct_ = s_ ? cb : ca; tb = s_ ? ta : tb; cb = s_ ? ca : cb; ta = tt_; ca = ct_; }
  RTC_CSWAP(t0, c0, t1, c1) RTC_CSWAP(t2, c2, t3, c3) RTC_CSWAP(t0, c0, t2, c2) RTC_CSWAP(t1, c1, t3, c3) RTC_CSWAP(t1, c1, t2, c2)
#undef RTC_CSWAP
  if (nh > 3) { stack[sp * stride] = c3; sp++; }
  if (nh > 2) { stack[sp * stride] = c2; sp++; }
  if (nh > 1) { stack[sp * stride] = c1; sp++; }
  cur = c0;
}

// First step of a walk whose root node travels in the kernel arguments (DScene.kaux; scalar loads): every lane starts at the
// root, so its boxes need no vector load.
template <bool PM = false>
__device__ __forceinline__ void walk_root_k(const DBvhNode4& R, const Frame32& F, const Trav& T, Counters& C, int& cur, int& sp, int* __restrict__ stack, int stride, bool analytic) {
  C.accel_nodes++;
  C.knodes++;
  float lo, hi;
  t_interval32(T, lo, hi);
  const float4 lox = {R.lox[0], R.lox[1], R.lox[2], R.lox[3]}, loy = {R.loy[0], R.loy[1], R.loy[2], R.loy[3]};
  const float4 loz = {R.loz[0], R.loz[1], R.loz[2], R.loz[3]}, hix = {R.hix[0], R.hix[1], R.hix[2], R.hix[3]};
  const float4 hiy = {R.hiy[0], R.hiy[1], R.hiy[2], R.hiy[3]}, hiz = {R.hiz[0], R.hiz[1], R.hiz[2], R.hiz[3]};
  const int4 cc = {R.c[0], R.c[1], R.c[2], R.c[3]};
  node_step<PM>(lox, loy, loz, hix, hiy, hiz, cc, F, lo, hi, cur, sp, stack, stride, T.mode == MODE_SHADOW_ANY && T.unordered, analytic);
}

// The walk itself, from (cur, sp): ONE inlined copy per traversal (the op loop sets the walk up per lane and all walks of a
// program — mesh or analytic, kernel-argument root or not — run through this loop).  `mesh` is uniform across the wave (it comes
// from the program op): leaves hold triangles of the packed arrays (object-space ray `o`) or name one analytic primitive.
template <int FEAT, bool LDSC = false, bool PM = false>
__device__ __forceinline__ void walk_loop(const DScene& S, const bool mesh, int cur, int sp, const Frame32& F, const Ray& world, const Ray& o, Trav& T, Counters& C,
                                          int* __restrict__ stack, int stride, const LdsScene& L = LdsScene{}) {
  const int END = RTC_WALK_END;
  const bool any_hit = T.mode == MODE_SHADOW_ANY && T.unordered;  // wavefront shadow role only: 3 % there, -2 % in the one-kernel path
  float lo, hi;  // the pass's t interval in f32 (widened): only a leaf test can change it
  t_interval32(T, lo, hi);
  for (;;) {
    DIAG_LOOP(0);
    // "while-while": descend through inner nodes until this lane holds a leaf (or has drained its stack); lanes that
    // already hold a leaf wait here, so the expensive leaf tests below run with as many lanes as possible.
    while (cur >= 0) {
      DIAG_LOOP(3);
      C.accel_nodes++;
      float4 lox, loy, loz, hix, hiy, hiz;
      int4 cc;
      if (LDSC) {
        const float4* N = L.nodes + cur;
        const int n = L.n_nodes;
        lox = N[0]; loy = N[n]; loz = N[2 * n]; hix = N[3 * n]; hiy = N[4 * n]; hiz = N[5 * n];
        cc = as_int4(N[6 * n]);
      } else {
        // the node's seven 16-byte rows: one line, all loads in flight together
        const DBvhNode4* N = S.bvh + cur;
        lox = ld4(N->lox); loy = ld4(N->loy); loz = ld4(N->loz); hix = ld4(N->hix); hiy = ld4(N->hiy); hiz = ld4(N->hiz);
        cc = ld4(N->c);
      }
      if (RTC_PROBE & 4) {  // cost probe: the node step twice (same pushes, same result)
        const int cur0 = cur, sp0 = sp;
        node_step<PM>(lox, loy, loz, hix, hiy, hiz, cc, F, lo, hi, cur, sp, stack, stride, any_hit, !mesh);
        RTC_LAUNDER(lox.x);
        cur = cur0; sp = sp0;
      }
      node_step<PM>(lox, loy, loz, hix, hiy, hiz, cc, F, lo, hi, cur, sp, stack, stride, any_hit, !mesh);
    }
    if (cur == END) return;
    {
      DIAG_SPAN_BEGIN();
      int first = (~cur) >> 3, cnt = ((~cur) & 7) + 1;
      if (mesh) {
        for (int i = first; i < first + cnt; i++) {
          DIAG_LOOP(1);
          double t, u, v;
          C.tri_tests++;
          if (LDSC) {
            double g[9];
#pragma unroll
            for (int c = 0; c < 9; c++) g[c] = L.tris[c * L.n_tris + i];
            if (tri_hit(g, o, t, u, v)) accept(T, C, L.tri_prim[i], 1, &t);
          } else if (tri_hit(S.mtri + 9 * (size_t)i, o, t, u, v)) accept(T, C, S.mtri_prim[i], 1, &t);
        }
      } else {
        DIAG_LOOP(1);
#pragma unroll 1
        for (int rep_ = 0; rep_ < ((RTC_PROBE & 16) ? 2 : 1); rep_++) {
          Ray w2 = world;
          if (RTC_PROBE & 16) RTC_LAUNDER(w2.ox);
          visit_prim<FEAT, LDSC>(S, first, w2, T, C, 1, L);  // analytic leaf = one primitive, named by the ref itself
        }
      }
      DIAG_SPAN_END(6);
      if (T.mode == MODE_SHADOW_ANY && T.shadowed) return;
      if (T.mode != MODE_SHADOW_ANY && T.mode != MODE_CONTAINERS) t_interval32(T, lo, hi);  // closest passes: best_t may have come down
    }
    if (sp == 0) return;
    sp--;
    cur = stack[sp * stride];
  }
}

// ---- CSG groups (src/shape.rs:161-178 allows_intersection, :230-246 filter_by_group, :257-266 Group::intersect) ------------
// ops[pc] is an OP_CSG.  The subtree's sub-program is walked once; every primitive's pushes go to a per-lane buffer in
// insertion order; at each OP_CSG_END the node's range of the buffer is stable-sorted by t and filtered in place — post-order,
// so a nested CSG hands its parent exactly the list the reference's recursion would.  What survives the outermost filter is
// fed to the current pass with each entry's own (primitive, push index), so tie-breaks and the container pass see the same keys.
typedef DCsgHit CsgHit;
__device__ __noinline__ int csg_eval(const DScene& S, int pc, const Ray& r, Trav& T, Counters& C) {
  CsgHit local[RTC_CSG_MAX_HITS];
  // scenes whose subtrees can produce more intersections than the per-lane buffer holds: this thread's rows of the launch's slab
  CsgHit* buf = local;
  int cap = RTC_CSG_MAX_HITS;
  if (S.csg_slab) { buf = S.csg_slab + ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * (size_t)S.csg_max_hits; cap = S.csg_max_hits; }
  int n = 0;
  int frame_begin[RTC_CSG_MAX_DEPTH];
  int depth = 0;
  const int end_pc = S.ops[pc].b;  // the matching OP_CSG_END
  int p = pc;
  while (p <= end_pc) {
    DOp op = S.ops[p];
    if (op.op == OP_CSG) {
      C.group_tests++;
      if (!group_box_hit(S.group_box + 6 * op.a, r)) { p = op.b + 1; continue; }  // whole node contributes nothing
      frame_begin[depth++] = n;
      p++;
    } else if (op.op == OP_GROUP) {
      C.group_tests++;
      p = group_box_hit(S.group_box + 6 * op.a, r) ? p + 1 : op.b;
    } else if (op.op == OP_PRIM) {
      const DPrimI P = S.pisect[op.a];
      Ray o = to_object(P.m, r);
      double t[4], u = 0.0, v = 0.0;
      if (P.geom >= 5) C.tri_tests++; else C.analytic_tests++;
      int m = prim_hits(S, P, o, t, u, v);
      for (int j = 0; j < m && n < cap; j++) {
        buf[n].t = t[j]; buf[n].prim = op.a; buf[n].k = j; n++;
      }
      p++;
    } else {  // OP_CSG_END: sort [b, n) by t (stable: insertion sort), then filter_by_group
      const DCsg G = S.csg[op.c];
      int b = frame_begin[--depth];
      if (n - b >= 2)  // the reference sorts this list: a NaN t panics once the comparator runs (src/intersection.rs:124)
        for (int i = b; i < n; i++)
          if (buf[i].t != buf[i].t) { C.nan_ts++; break; }
      for (int i = b + 1; i < n; i++) {
        CsgHit h = buf[i];
        int j = i - 1;
        while (j >= b && buf[j].t > h.t) { buf[j + 1] = buf[j]; j--; }
        buf[j + 1] = h;
      }
      bool in_left = false, in_right = false;
      int w = b;
      for (int i = b; i < n; i++) {
        bool left_hit = buf[i].prim >= G.left_first && buf[i].prim < G.left_end;
        bool keep;
        if (G.kind == 0) keep = (left_hit && !in_right) || (!left_hit && !in_left);         // Union
        else if (G.kind == 1) keep = (left_hit && in_right) || (!left_hit && in_left);      // Intersection
        else keep = (left_hit && !in_right) || (!left_hit && in_left);                      // Difference
        if (left_hit) in_left = !in_left; else in_right = !in_right;
        if (keep) buf[w++] = buf[i];
      }
      n = w;
      p++;
    }
  }
  // hand the retained intersections to the pass, one primitive at a time (its entries keep their relative order)
  for (int i = 0; i < n; i++) {
    int prim = buf[i].prim;
    bool seen = false;
    for (int j = 0; j < i; j++) seen = seen || buf[j].prim == prim;
    if (seen) continue;
    double t[8];
    int ks[8], m = 0;
    for (int j = i; j < n && m < 8; j++)
      if (buf[j].prim == prim) { t[m] = buf[j].t; ks[m] = buf[j].k; m++; }
    accept(T, C, prim, m, t, ks);
  }
  return end_pc + 1;
}

// World::intersect (src/world.rs:18-24) + Group::intersect (src/shape.rs:248-269) over the flattened program.
// KOPS: the program is the short jump-free one of the kernel arguments (DScene.kops: pc is wave-uniform, the op and the plane
// records are scalar loads); else the program array in memory (any length, OP_GROUP jumps, per-primitive gates, CSG).  A kernel
// instantiation has exactly one of the two, and each has exactly ONE inlined copy of the BVH walk.
// MODE (optional): the pass kind as a compile-time constant (the caller set T.mode to it); -1 = read T.mode at run time.
template <int FEAT, bool KOPS, int MODE = -1, bool LDSC = false>
__device__ TRAVERSE_INLINE void traverse(const DScene& S, const Ray& r, Trav& T, Counters& C, int* __restrict__ stack, int stride, const LdsScene& L = LdsScene{}) {
  if (MODE >= 0) T.mode = MODE;
  T.flags = 0;
  if (S.quirk_reach2 > 0.0) {
    const double m = fmax(fmax(fabs(r.ox - S.abvh_frame[0]), fabs(r.oy - S.abvh_frame[1])), fabs(r.oz - S.abvh_frame[2])) + S.abvh_frame[3];
    const double len2 = r.dx * r.dx + r.dy * r.dy + r.dz * r.dz;
    T.flags = (len2 >= 0.0025 && m * m <= S.quirk_reach2 * len2) ? TF_CUBES_IN_LEAF : 0;  // (a NaN anywhere: 0, the scan)
  }
  const LightCell lcell = light_grid_fetch<MODE>(S, r, T);
  if (KOPS) {
    for (int pc = 0; pc < S.n_kops; pc++) {
      DIAG_LOOP(2);
      const DOp op = S.kops[pc];
      bool walk = false;
      const bool mesh = op.op == OP_MESH;
      int cur = 0, sp = 0;
      Frame32 F;
      Ray o = r;
      if (op.op == OP_PRIM) {
        if (op.c >= 0) {
          // Plane (src/shape.rs:621-633): the same row-1 evaluation as visit_prim's plane case, operands from kernargs
          const DPlaneK P = S.kplanes[op.c];
          // a plane under aggregation groups is reached through their box tests like any other child (src/shape.rs:251): a group
          // that holds a plane has an unbounded or NaN-poisoned box, which a finite ray nearly always passes and a NaN ray never
          if (FEAT >= 2 && P.gcond >= 0 && !kops_gate(S, P.gcond, r, T, C)) continue;
          C.analytic_tests++;
          C.kplanes++;
          DIAG_LOOP(8);
          for (int rep_ = 0; rep_ < ((RTC_PROBE & 1) ? 2 : 1); rep_++) {
            double rox = r.ox;
            if (RTC_PROBE & 1) RTC_LAUNDER(rox);
            double oy = P.row[0] * rox + P.row[1] * r.oy + P.row[2] * r.oz + P.row[3] * 1.0;
            double dy = P.row[0] * r.dx + P.row[1] * r.dy + P.row[2] * r.dz + P.row[3] * 0.0;
            if (RTC_PROBE & 1) RTC_LAUNDER(dy);
            if (!(fabs(dy - 0.0) < EPS) && !plane_behind(T, oy, dy)) {
              double t = -oy / dy;
              accept(T, C, P.prim, 1, &t);
            }
          }
        } else {
          visit_prim<FEAT, LDSC>(S, op.a, r, T, C, 0, L);
        }
      } else if ((op.op == OP_QUIRK || op.op == OP_QGRID) && op.c == 1 && (T.flags & TF_CUBES_IN_LEAF)) {
        // the cubes' quirk scan: this ray's leaves have tested them
      } else if (op.op == OP_QUIRK) {
        for (int i = op.a; i < op.a + op.b; i++) visit_prim<FEAT, LDSC>(S, S.quirk_prim[i], r, T, C, 2, L);
      } else if (op.op == OP_QGRID) {
        DIAG_SPAN_BEGIN();
        for (int rep_ = 0; rep_ < ((RTC_PROBE & 2) ? 2 : 1); rep_++) {
          Ray rr = r;
          if (RTC_PROBE & 2) RTC_LAUNDER(rr.dx);
          quirk_grid_scan<FEAT, LDSC>(S, op.pad[0] >= 0 ? S.kqgrid : S.qgrids[op.a], rr, T, C, L);
        }
        DIAG_SPAN_END(4);
      } else if (!mesh || FEAT == 0 || op.g < 0 || kops_gate(S, op.g, r, T, C)) {  // OP_MESH / OP_BVH
        walk = true;
        cur = op.a;
        if (!mesh && light_grid_candidates(S, lcell, T, C, cur, sp, stack, stride)) {
          walk = cur != RTC_WALK_END;
        } else if (op.pad[0] >= 0) {
          DIAG_LOOP(16);
          const DKAux& A = S.kaux[op.pad[0]];
          if (mesh) o = to_object(A.xf, r);
          make_frame(A.frame, o, F);
          if (MODE == MODE_CONTAINERS) make_frame_point(A.frame, o, T.thi, F);
#ifndef RTC_NO_KROOT
          walk_root_k<MODE == MODE_CONTAINERS>(A.root, F, T, C, cur, sp, stack, stride, !mesh);
#endif
        } else {
          if (mesh) o = to_object(S.xf_inv + 12 * op.b, r);
          make_frame(S.bvh_frame + 4 * op.c, o, F);
          if (MODE == MODE_CONTAINERS) make_frame_point(S.bvh_frame + 4 * op.c, o, T.thi, F);
        }
      }
      if (walk) walk_loop<FEAT, LDSC, MODE == MODE_CONTAINERS>(S, mesh, cur, sp, F, r, o, T, C, stack, stride, L);
      if (T.mode == MODE_SHADOW_ANY && T.shadowed) return;
    }
    return;
  }
  int pc = 0;
  const int n = S.n_ops;
  while (pc < n) {
    DIAG_LOOP(2);
    DOp op = S.ops[pc];
    bool walk = false, mesh = false;
    int cur = 0, sp = 0;
    Frame32 F;
    Ray o = r;
    if (op.op == OP_PRIM) {
      visit_prim<FEAT, LDSC>(S, op.a, r, T, C, 0, L);
      pc++;
    } else if ((op.op == OP_QUIRK || op.op == OP_QGRID) && op.c == 1 && (T.flags & TF_CUBES_IN_LEAF)) {
      pc++;  // the cubes' quirk scan: this ray's leaves have tested them
    } else if (op.op == OP_QUIRK) {
      for (int i = op.a; i < op.a + op.b; i++) visit_prim<FEAT, LDSC>(S, S.quirk_prim[i], r, T, C, 2, L);
      pc++;
    } else if (op.op == OP_QGRID) {
      DIAG_SPAN_BEGIN();
      quirk_grid_scan<FEAT, LDSC>(S, S.qgrids[op.a], r, T, C, L);
      DIAG_SPAN_END(4);
      pc++;
    } else if (op.op == OP_GROUP) {
      C.group_tests++;
      pc = group_box_hit(S.group_box + 6 * op.a, r) ? pc + 1 : op.b;
    } else if (FEAT >= 3 && op.op == OP_CSG) {
      if (op.g >= 0 && !groups_pass(S, op.g, r, T, C)) pc = op.b + 1;
      else pc = csg_eval(S, pc, r, T, C);
    } else {  // OP_MESH / OP_BVH
      mesh = op.op == OP_MESH;
      cur = op.a; sp = 0;
      if (!mesh || FEAT == 0 || op.g < 0 || groups_pass<(FEAT >= 2)>(S, op.g, r, T, C)) {
        walk = true;
        if (!mesh && light_grid_candidates(S, lcell, T, C, cur, sp, stack, stride)) {
          walk = cur != RTC_WALK_END;
        } else {
          if (mesh) o = to_object(S.xf_inv + 12 * op.b, r);
          make_frame(S.bvh_frame + 4 * op.c, o, F);
          if (MODE == MODE_CONTAINERS) make_frame_point(S.bvh_frame + 4 * op.c, o, T.thi, F);
        }
      }
      pc++;
    }
    if (walk) walk_loop<FEAT, LDSC, MODE == MODE_CONTAINERS>(S, mesh, cur, sp, F, r, o, T, C, stack, stride, L);
    if (T.mode == MODE_SHADOW_ANY && T.shadowed) return;
  }
}

// ---- noise (src/noise.rs) ------------------------------------------------------------------------------
__device__ const unsigned char PERM[256] = {
    151, 160, 137, 91,  90,  15,  131, 13,  201, 95,  96,  53,  194, 233, 7,   225, 140, 36,  103, 30,  69,  142, 8,   99,  37,  240,
    21,  10,  23,  190, 6,   148, 247, 120, 234, 75,  0,   26,  197, 62,  94,  252, 219, 203, 117, 35,  11,  32,  57,  177, 33,  88,
    237, 149, 56,  87,  174, 20,  125, 136, 171, 168, 68,  175, 74,  165, 71,  134, 139, 48,  27,  166, 77,  146, 158, 231, 83,  111,
    229, 122, 60,  211, 133, 230, 220, 105, 92,  41,  55,  46,  245, 40,  244, 102, 143, 54,  65,  25,  63,  161, 1,   216, 80,  73,
    209, 76,  132, 187, 208, 89,  18,  169, 200, 196, 135, 130, 116, 188, 159, 86,  164, 100, 109, 198, 173, 186, 3,   64,  52,  217,
    226, 250, 124, 123, 5,   202, 38,  147, 118, 126, 255, 82,  85,  212, 207, 206, 59,  227, 47,  16,  58,  17,  182, 189, 28,  42,
    223, 183, 170, 213, 119, 248, 152, 2,   44,  154, 163, 70,  221, 153, 101, 155, 167, 43,  172, 9,   129, 22,  39,  253, 19,  98,
    108, 110, 79,  113, 224, 232, 178, 185, 112, 104, 218, 246, 97,  228, 251, 34,  242, 193, 238, 210, 144, 12,  191, 179, 162, 241,
    81,  51,  145, 235, 249, 14,  239, 107, 49,  192, 214, 31,  181, 199, 106, 157, 184, 84,  204, 176, 115, 121, 50,  45,  127, 4,
    150, 254, 138, 236, 205, 93,  222, 114, 67,  29,  24,  72,  243, 141, 128, 195, 78,  66,  215, 61,  156, 180};

__device__ __forceinline__ unsigned nhash(unsigned i) { return PERM[i & 255u]; }  // :90-92, table period 256

__device__ __forceinline__ int as_i32(double x) {  // Rust `as i32`: saturating, NaN -> 0
  if (x != x) return 0;
  if (x >= 2147483647.0) return 2147483647;
  if (x <= -2147483648.0) return (int)0x80000000;
  return (int)x;
}
__device__ __forceinline__ int wadd(int a, int b) { return (int)((unsigned)a + (unsigned)b); }
__device__ __forceinline__ int fast_floor(double x) { return x > 0.0 ? as_i32(x) : (int)((unsigned)as_i32(x) - 1u); }  // :116-122
__device__ __forceinline__ unsigned modulus256(int x) { int a = x % 256; return a < 0 ? (unsigned)(a + 256) : (unsigned)a; }  // :124-131

__device__ __forceinline__ double ngrad(unsigned h, double x, double y, double z) {  // :94-114
  switch (h & 0xF) {
    case 0x0: return x + y;
    case 0x1: return -x + y;
    case 0x2: return x - y;
    case 0x3: return -x - y;
    case 0x4: return x + z;
    case 0x5: return -x + z;
    case 0x6: return x - z;
    case 0x7: return -x - z;
    case 0x8: return y + z;
    case 0x9: return -y + z;
    case 0xA: return y - z;
    case 0xB: return -y - z;
    case 0xC: return y + x;
    case 0xD: return -y + z;
    case 0xE: return y - x;
    default: return -y - z;
  }
}

__device__ __noinline__ double simplex3(double x, double y, double z) {  // :134-219
  const double F3 = 1.0 / 3.0, G3 = 1.0 / 6.0;
  double s = (x + y + z) * F3;
  int i = fast_floor(x + s), j = fast_floor(y + s), k = fast_floor(z + s);
  double t = (double)wadd(wadd(i, j), k) * G3;
  double x0 = x - ((double)i - t), y0 = y - ((double)j - t), z0 = z - ((double)k - t);
  int i1, j1, k1, i2, j2, k2;
  if (x0 >= y0) {
    if (y0 >= z0) { i1 = 1; j1 = 0; k1 = 0; i2 = 1; j2 = 1; k2 = 0; }
    else if (x0 >= z0) { i1 = 1; j1 = 0; k1 = 0; i2 = 1; j2 = 0; k2 = 1; }
    else { i1 = 0; j1 = 0; k1 = 1; i2 = 1; j2 = 0; k2 = 1; }
  } else {
    if (y0 < z0) { i1 = 0; j1 = 0; k1 = 1; i2 = 0; j2 = 1; k2 = 1; }
    else if (x0 < z0) { i1 = 0; j1 = 1; k1 = 0; i2 = 0; j2 = 1; k2 = 1; }
    else { i1 = 0; j1 = 1; k1 = 0; i2 = 1; j2 = 1; k2 = 0; }
  }
  double x1 = x0 - (double)i1 + G3, y1 = y0 - (double)j1 + G3, z1 = z0 - (double)k1 + G3;
  double x2 = x0 - (double)i2 + 2.0 * G3, y2 = y0 - (double)j2 + 2.0 * G3, z2 = z0 - (double)k2 + 2.0 * G3;
  double x3 = x0 - 1.0 + 3.0 * G3, y3 = y0 - 1.0 + 3.0 * G3, z3 = z0 - 1.0 + 3.0 * G3;
  unsigned ii = modulus256(i), jj = modulus256(j), kk = modulus256(k);
  unsigned gi0 = nhash(ii + nhash(jj + nhash(kk)));
  unsigned gi1 = nhash(ii + i1 + nhash(jj + j1 + nhash(kk + k1)));
  unsigned gi2 = nhash(ii + i2 + nhash(jj + j2 + nhash(kk + k2)));
  unsigned gi3 = nhash(ii + 1 + nhash(jj + 1 + nhash(kk + 1)));
  double n0, n1, n2, n3;
  double t0 = 0.6 - x0 * x0 - y0 * y0 - z0 * z0;
  if (t0 < 0.0) n0 = 0.0; else { t0 *= t0; n0 = t0 * t0 * ngrad(gi0, x0, y0, z0); }
  double t1 = 0.6 - x1 * x1 - y1 * y1 - z1 * z1;
  if (t1 < 0.0) n1 = 0.0; else { t1 *= t1; n1 = t1 * t1 * ngrad(gi1, x1, y1, z1); }
  double t2 = 0.6 - x2 * x2 - y2 * y2 - z2 * z2;
  if (t2 < 0.0) n2 = 0.0; else { t2 *= t2; n2 = t2 * t2 * ngrad(gi2, x2, y2, z2); }
  double t3 = 0.6 - x3 * x3 - y3 * y3 - z3 * z3;
  if (t3 < 0.0) n3 = 0.0; else { t3 *= t3; n3 = t3 * t3 * ngrad(gi3, x3, y3, z3); }
  return 32.0 * (n0 + n1 + n2 + n3);
}

__device__ __forceinline__ double fractal3(double x, double y, double z, unsigned octaves) {  // :221-237
  double output = 0.0, denom = 0.0, frequency = 1.0, amplitude = 1.0;
  for (unsigned o = 0; o < octaves; o++) {
    output += amplitude * simplex3(x * frequency, y * frequency, z * frequency);
    denom += amplitude;
    frequency *= 2.0;
    amplitude *= 0.5;
  }
  return output / denom;
}

__device__ __forceinline__ void jitter_3d(const DPat& p, double x, double y, double z, double& ox, double& oy, double& oz) {  // :31-52
  double nx, ny, nz;
  if (p.noise_kind == 0) {
    nx = simplex3(x, y, z) * p.scale;
    ny = simplex3(x, y, z + 1.0) * p.scale;
    nz = simplex3(x, y, z + 2.0) * p.scale;
  } else {
    nx = fractal3(x, y, z, p.octaves) * p.scale;
    ny = fractal3(x, y, z + 1.0, p.octaves) * p.scale;
    nz = fractal3(x, y, z + 2.0, p.octaves) * p.scale;
  }
  ox = x + nx; oy = y + ny; oz = z + nz;
}

// ---- Pattern::color_at (src/material.rs:164-302) as an explicit-stack walk of the node array ------------
struct PFrame {
  int node, stage;
  double px, py, pz, pw, frac;
  double lr, lg, lb;
};

__device__ __noinline__ void pattern_color(const DScene& S, int root, double px, double py, double pz, double pw, double& r, double& g, double& b) {
  PFrame fr[8];  // RTC_MAX_PATTERN_DEPTH, validated at scene creation
  int sp = 0;
  int node = root;
  for (;;) {
    // ---- descend until a colour is known
    for (;;) {
      const DPat& p = S.pats[node];
      if (p.tag == 0) { r = px; g = py; b = pz; break; }
      if (p.tag == 1) { r = p.color[0]; g = p.color[1]; b = p.color[2]; break; }
      if (p.tag == 2) {
        if (p.kind == 1) {  // JitterKind::Point :218-221
          double nx, ny, nz;
          jitter_3d(p, px, py, pz, nx, ny, nz);
          px = nx; py = ny; pz = nz; pw = 1.0;
          node = p.left;
        } else {  // JitterKind::Color :209-217
          fr[sp].node = node; fr[sp].stage = 2; sp++;
          node = p.left;
        }
        continue;
      }
      // Mixture: point = transform_inv * point (:181-183)
      {
        const double* m = p.m;
        double x = m[0] * px + m[1] * py + m[2] * pz + m[3] * pw;
        double y = m[4] * px + m[5] * py + m[6] * pz + m[7] * pw;
        double z = m[8] * px + m[9] * py + m[10] * pz + m[11] * pw;
        double w = m[12] * px + m[13] * py + m[14] * pz + m[15] * pw;
        px = x; py = y; pz = z; pw = w;
      }
      if (p.kind == 1) {  // Checkers :258-268
        int xi = as_i32(floor(px)), yi = as_i32(floor(py)), zi = as_i32(floor(pz));
        node = (wadd(wadd(xi, yi), zi) % 2 == 0) ? p.left : p.right;
      } else if (p.kind == 3) {  // Ring :278-284
        node = (as_i32(floor(sqrt(px * px + pz * pz))) % 2 == 0) ? p.left : p.right;
      } else if (p.kind == 5) {  // Stripes :293-299
        node = (as_i32(floor(px)) % 2 == 0) ? p.left : p.right;
      } else {  // Blend / RingGradient / Gradient: both children
        double frac = 0.0;
        if (p.kind == 2) { double d = sqrt(px * px + py * py + pz * pz); frac = d - floor(d); }  // :269-277
        else if (p.kind == 4) frac = px - floor(px);                                             // :285-292
        fr[sp].node = node; fr[sp].stage = 0; fr[sp].px = px; fr[sp].py = py; fr[sp].pz = pz; fr[sp].pw = pw; fr[sp].frac = frac;
        sp++;
        node = p.left;
      }
    }
    // ---- ascend
    bool again = false;
    while (sp > 0) {
      PFrame& f = fr[sp - 1];
      const DPat& p = S.pats[f.node];
      if (f.stage == 2) {
        double nr, ng, nb;
        jitter_3d(p, r, g, b, nr, ng, nb);
        r = nr; g = ng; b = nb;
        sp--;
      } else if (f.stage == 0) {
        f.lr = r; f.lg = g; f.lb = b; f.stage = 1;
        px = f.px; py = f.py; pz = f.pz; pw = f.pw;
        node = p.right;
        again = true;
        break;
      } else {
        if (p.kind == 0) { r = (f.lr + r) * 0.5; g = (f.lg + g) * 0.5; b = (f.lb + b) * 0.5; }  // Color::avg
        else { r = f.lr + ((r - f.lr) * f.frac); g = f.lg + ((g - f.lg) * f.frac); b = f.lb + ((b - f.lb) * f.frac); }
        sp--;
      }
    }
    if (!again) return;
  }
}

// ---- hit state (Intersection::prepare_state, src/intersection.rs:50-121) --------------------------------
struct State {
  double px, py, pz;     // over_point
  double ux, uy, uz;     // under_point
  double nx, ny, nz;     // normal (flipped when inside)
  double ex, ey, ez;     // eye
  double rx, ry, rz;     // reflect
};

// Geometry::normal (src/shape.rs:887-946) in object space
__device__ __forceinline__ void local_normal(const DScene& S, const DPrim& P, double x, double y, double z, double u, double v, double& nx, double& ny, double& nz) {
  switch (P.geom) {
    case 0: nx = x; ny = y; nz = z; return;
    case 1: nx = 0.0; ny = 1.0; nz = 0.0; return;
    case 2: {
      double xa = fabs(x), ya = fabs(y), za = fabs(z);
      double mx = rmax(rmax(xa, ya), za);
      if (mx == xa) { nx = x; ny = 0.0; nz = 0.0; }
      else if (mx == ya) { nx = 0.0; ny = y; nz = 0.0; }
      else { nx = 0.0; ny = 0.0; nz = z; }
      return;
    }
    case 3:
    case 4: {
      double mn = S.limits[2 * P.data], mxl = S.limits[2 * P.data + 1];
      double dist = x * x + z * z;
      if (dist < 1.0 && y >= mxl - EPS) { nx = 0.0; ny = 1.0; nz = 0.0; return; }
      if (dist < 1.0 && y <= mn + EPS) { nx = 0.0; ny = -1.0; nz = 0.0; return; }
      if (P.geom == 3) { nx = x; ny = 0.0; nz = z; return; }
      double yy = sqrt(dist);
      if (y > 0.0) yy = -yy;
      nx = x; ny = yy; nz = z;
      return;
    }
    case 5: {
      const double* n = S.tri_nrm + 9 * P.data;
      nx = n[0]; ny = n[1]; nz = n[2];
      return;
    }
    default: {  // *n2 * u + *n3 * v + *n1 * (1.0 - u - v)
      const double* n = S.tri_nrm + 9 * P.data;
      double w = 1.0 - u - v;
      nx = n[3] * u + n[6] * v + n[0] * w;
      ny = n[4] * u + n[7] * v + n[1] * w;
      nz = n[5] * u + n[8] * v + n[2] * w;
      return;
    }
  }
}

__device__ __forceinline__ void prepare_state(const DScene& S, const DPrim& P, const Ray& r, double t, double u, double v, State& st) {
  // point = ray.position(t); eye = -direction
  double qx = r.ox + r.dx * t, qy = r.oy + r.dy * t, qz = r.oz + r.dz * t;
  st.ex = -r.dx; st.ey = -r.dy; st.ez = -r.dz;
  double nx, ny, nz;
  if (P.geom == 1 && P.pad[0] > 0) {
    // A plane's normal does not depend on the point (src/shape.rs:896: vector(0, 1, 0)): the host evaluated this function's own
    // expressions for it once (scene_build.hpp plane_world_normal: same operations, same order, same bits) and it travels in the
    // kernel arguments — most hits of the room scenes are on planes, and this was a square root and three divisions each.
    const DPlaneK& K = S.kplanes[P.pad[0] - 1];
    nx = K.n[0]; ny = K.n[1]; nz = K.n[2];
  } else {
    // Shape::normal (src/shape.rs:419-427)
    const double* m = S.xf_inv + 12 * P.xform;
    double sx = m[0] * qx + m[1] * qy + m[2] * qz + m[3] * 1.0;
    double sy = m[4] * qx + m[5] * qy + m[6] * qz + m[7] * 1.0;
    double sz = m[8] * qx + m[9] * qy + m[10] * qz + m[11] * 1.0;
    double lx, ly, lz;
    local_normal(S, P, sx, sy, sz, u, v, lx, ly, lz);
    // transform_inv_tsp * n: row r of the transpose = column r of transform_inv; the w term is (+-0)*0
    double wx = m[0] * lx + m[4] * ly + m[8] * lz + 0.0;
    double wy = m[1] * lx + m[5] * ly + m[9] * lz + 0.0;
    double wz = m[2] * lx + m[6] * ly + m[10] * lz + 0.0;
    double mag = sqrt(wx * wx + wy * wy + wz * wz);
    nx = wx / mag; ny = wy / mag; nz = wz / mag;
  }
  if (nx * st.ex + ny * st.ey + nz * st.ez < 0.0) { nx = -nx; ny = -ny; nz = -nz; }
  st.nx = nx; st.ny = ny; st.nz = nz;
  st.px = qx + nx * EPS; st.py = qy + ny * EPS; st.pz = qz + nz * EPS;
  st.ux = qx - nx * EPS; st.uy = qy - ny * EPS; st.uz = qz - nz * EPS;
  // reflect = direction - normal * (2 * direction.dot(normal))
  double dn = 2.0 * (r.dx * nx + r.dy * ny + r.dz * nz);
  st.rx = r.dx - nx * dn; st.ry = r.dy - ny * dn; st.rz = r.dz - nz * dn;
}

// schlick (src/intersection.rs:24-39)
__device__ __forceinline__ double schlick(const State& st, double n1, double n2) {
  double c = st.ex * st.nx + st.ey * st.ny + st.ez * st.nz;
  if (n1 > n2) {
    double n = n1 / n2;
    double sin2_t = n * n * (1.0 - c * c);
    if (sin2_t > 1.0) return 1.0;
    c = sqrt(1.0 - sin2_t);
  }
  double q = (n1 - n2) / (n1 + n2);
  double r0 = q * q;
  double x = 1.0 - c;
  double x5 = x * ((x * x) * (x * x));
  return r0 + (1.0 - r0) * x5;
}

// The reference blends the reflected and the refracted colour of a reflective AND transparent surface with state.reflectance for
// every light, whatever the fuel and whatever the two colours are (src/world.rs:70-78): a NaN reflectance -- a NaN normal, e.g. the
// zero local normal at a cone's apex -- makes the pixel NaN even where both are black (0 * NaN).  Returns the reflectance; a NaN
// one also poisons the surface colour, which every light's term carries.  With fuel left n1 / n2 are known and this is schlick();
// at fuel 0 (no container pass) only the NaN-ness matters and that is eye . normal's: r0 is finite for the refractive indices
// validate() admits (scene_build.hpp).
__device__ __forceinline__ double blend_reflectance(const State& st, double n1, double n2, int fuel, double& cr, double& cg, double& cb) {
  const double R = fuel > 0 ? schlick(st, n1, n2) : st.ex * st.nx + st.ey * st.ny + st.ez * st.nz;
  if (R != R) { cr = R; cg = R; cb = R; }
  return R;
}

// A light BEHIND the surface: Shape::lighting adds diffuse and specular only if `!shadowed && light . normal >= 0`
// (src/shape.rs:448-459), so where light . normal < 0 the answer of World::is_shadowed cannot change the pixel, and the shadow ray
// need not be traced -- PROVIDED tracing it could not have panicked either: the reference sorts its intersections whatever the answer
// is worth, and a NaN t in a list of two or more panics.  A finite ray cannot make a NaN t while no intermediate of the intersection
// formulas overflows (planes, triangles and the quadrics divide by quantities they have just compared with EPSILON; a cube's
// 0 * inf is ignored by f64::max / min unless all three axes have one; infinite limits give infinite, not NaN, cap t's), which
// DScene.backface_skip vouches for together with the bound on |v| here.  The decision uses v . n with a margin of 1e-6 |v| -- far
// outside the rounding of the reference's own `(v / |v|) . n` (~1e-15), so every ray skipped here has a computed light . normal
// below zero; NaN operands fail the comparisons (then the ray is traced).  v = light - over_point, n = the unit normal.
__device__ __forceinline__ bool light_is_behind(const DScene& S, double vx, double vy, double vz, double nx, double ny, double nz) {
  const double vn = vx * nx + vy * ny + vz * nz, vv = vx * vx + vy * vy + vz * vz;
  return S.backface_skip && vn < 0.0 && vn * vn > 1e-12 * vv && vv < 1e60;
}

__device__ __forceinline__ void reset_closest(Trav& T, int mode) {
  T.mode = mode;
  T.tlo = 0.0; T.thi = DINF;
  T.best_t = DINF; T.best_prim = 0x7fffffff; T.best_k = 0; T.best_klast = 0;
  T.shadowed = 0;
  T.unordered = 0;
  T.light = -1;
  T.flags = 0;
  T.c1_t = 0.0; T.c2_t = 0.0; T.c1_prim = -1; T.c2_prim = -1;
  T.g_known = 0ull; T.g_pass = 0ull;
}

// u, v of the winning triangle: the traversal does not carry them; the same test on the same numbers gives the same bits.
__device__ __forceinline__ void hit_uv(const DScene& S, const DPrim& P, const Ray& world, double& u, double& v) {
  u = 0.0; v = 0.0;
  if (P.geom >= 5) {
    Ray o = to_object(S.xf_inv + 12 * P.xform, world);
    double t;
    tri_hit(S.tri_geo + 9 * P.data, o, t, u, v);
  }
}

struct Pending {
  double ox, oy, oz, dx, dy, dz, weight;
  int fuel, kind;
};

// Camera::ray_at_pixel (src/camera.rs:39-55)
__device__ __forceinline__ Ray camera_ray(const DCamera& cam, uint64_t i) {
  uint64_t x = i % cam.hsize, y = i / cam.hsize;
  double xoffset = ((double)x + 0.5) * cam.pixel_size;
  double yoffset = ((double)y + 0.5) * cam.pixel_size;
  double world_x = cam.half_width - xoffset;
  double world_y = cam.half_height - yoffset;
  const double* m = cam.inv;
  double px = m[0] * world_x + m[1] * world_y + m[2] * -1.0 + m[3] * 1.0;
  double py = m[4] * world_x + m[5] * world_y + m[6] * -1.0 + m[7] * 1.0;
  double pz = m[8] * world_x + m[9] * world_y + m[10] * -1.0 + m[11] * 1.0;
  Ray r;
  r.ox = m[3]; r.oy = m[7]; r.oz = m[11];
  double dx = px - r.ox, dy = py - r.oy, dz = pz - r.oz;
  double mag = sqrt(dx * dx + dy * dy + dz * dz);
  r.dx = dx / mag; r.dy = dy / mag; r.dz = dz / mag;
  return r;
}



// Pixel slots.  A launch covers `pm.n` output slots.  With tiling (full-width pixel sets: mode 0 with whole rows, mode 2) the
// work items are enumerated tile by tile (8x8 pixels, 64 consecutive work ids = one tile) so that the lanes of a wave,
// and the pixels a lane fetches later, stay spatially close; a work id that falls outside the image is skipped.
struct WorkMap {
  uint64_t n_work;      // number of work ids
  uint32_t tiled, tiles_x, width, height;
};
__device__ __forceinline__ WorkMap make_workmap(const DPixelMap& pm, const DCamera& cam) {
  WorkMap w;
  w.tiled = 0; w.tiles_x = 0; w.width = 0; w.height = 0; w.n_work = pm.n;
  if (pm.mode == 2 && cam.hsize >= 8 && pm.n % cam.hsize == 0) {
    w.tiled = 1;
    w.width = (uint32_t)cam.hsize;
    w.height = (uint32_t)(pm.n / cam.hsize);
    w.tiles_x = (w.width + 7u) / 8u;
    w.n_work = (uint64_t)w.tiles_x * ((w.height + 7u) / 8u) * 64u;
  }
  return w;
}
// work id -> output slot q (row-major within the launch's pixel set); false if the id is padding.
__device__ __forceinline__ bool work_to_slot(const WorkMap& w, uint64_t id, uint64_t& q) {
  if (!w.tiled) { q = id; return true; }
  uint64_t tile = id >> 6;
  uint32_t in = (uint32_t)(id & 63u);
  uint32_t x = (uint32_t)(tile % w.tiles_x) * 8u + (in & 7u), y = (uint32_t)(tile / w.tiles_x) * 8u + (in >> 3);
  if (x >= w.width || y >= w.height) return false;
  q = (uint64_t)y * w.width + x;
  return true;
}
__device__ __forceinline__ Ray slot_ray(const DPixelMap& pm, const DCamera& cam, uint64_t q) {
  if (pm.mode == 3) {
    const double* rr = pm.rays + 6 * q;
    Ray r;
    r.ox = rr[0]; r.oy = rr[1]; r.oz = rr[2]; r.dx = rr[3]; r.dy = rr[4]; r.dz = rr[5];
    return r;
  }
  // the host only issues modes 1 (index list), 2 (interleaved rows; a contiguous whole-row range is step 1) and 3 (rays)
  uint64_t i;
  if (pm.mode == 1) i = pm.indices[q];
  else {
    const uint64_t j = q / cam.hsize, band = pm.band ? pm.band : 1u;
    i = (((uint64_t)pm.row_first + (j / band) * pm.row_step) * band + j % band) * cam.hsize + (q % cam.hsize);
  }
  return camera_ray(cam, i);
}

}  // namespace

// One-kernel path: one lane walks one pixel's whole ray tree (closest pass, shading, shadow passes, pending children); the wave
// ends with its slowest pixel.  (A persistent variant whose lanes took the next work id from a global counter was measured at
// -4 % / +14 % and removed in round 2.)
// WAVES: waves per SIMD the kernel is compiled for; 0 = the default budget (every instantiation then lands at 2: 205-256 VGPRs).
// Scenes whose accelerator does not fit the L2s (config 5: 400 MB) are bound by node-fetch latency and run 13 % faster at 3 waves
// (168 VGPRs, more spills); cache-resident scenes are 3-9 % slower there (profiles/r2_onekernel_occupancy.txt).
// (LDS-resident scene tables, as in wf_ts, were measured on this kernel too — 256-thread blocks, two per CU — and lost 10 % on
// config 3: its 22 KB of nodes and triangles already hit in L1, and a block retires with its slowest wave.)
// LEAN: for scenes whose patterns are all Plain colours and whose materials never both reflect and refract (DScene.all_plain,
// no_glass_mirror: the teapot scenes) — no pattern-tree walk is compiled in and the pending-ray stack has one (unused) entry: 80 B of
// scratch per lane instead of 1 648 (config 3 -3 %, config 4 -4 %: profiles/r3_partition_probe.txt).
template <bool COUNT, int FEAT, bool KOPS, int WAVES = 0, bool LEAN = false>
__global__ void __launch_bounds__(RTC_BLOCK, WAVES ? WAVES : ((FEAT >= 2 && RTC_WAVES_PER_SIMD < 2) ? 2 : RTC_WAVES_PER_SIMD)) rtc_trace_kernel(DScene S, DCamera cam, DPixelMap pm, int fuel0, double* __restrict__ rgb, double* __restrict__ hit_t,
                                                        int* __restrict__ hit_prim, int* __restrict__ hit_k, DStats* __restrict__ stats) {
  RTC_LDS_STACK(lds_stack);
  int* stack = lds_stack + threadIdx.x;
  const int stride = RTC_BLOCK;
  Counters C = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned n_primary = 0, n_shadow = 0, n_reflect = 0, n_refract = 0, n_container = 0;
  const WorkMap wm = make_workmap(pm, cam);
#ifdef RTC_DIAG
  if (threadIdx.x < 64) s_diag[threadIdx.x] = 0ull;
  __syncthreads();
  unsigned long long diag_t0_ = 0;
  const unsigned long long diag_k0 = __builtin_amdgcn_s_memtime();
#endif

  const uint64_t id = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t q = 0;
  const bool have = id < wm.n_work && work_to_slot(wm, id, q);

  if (have) {
    Ray ray = slot_ray(pm, cam, q);

    Pending pend[LEAN ? 1 : RTC_MAX_FUEL];
    int np = 0;
    double acc_r = 0.0, acc_g = 0.0, acc_b = 0.0;
    double weight = 1.0;
    int fuel = fuel0;
    int kind = 0;
    bool first = true;
    unsigned long long dg = 0ull;  // hit-tree digest (counting variant with pm.digest set)
    const double L = (double)S.n_lights;

    for (;;) {
      if (kind == 0) n_primary++; else if (kind == 1) n_reflect++; else n_refract++;
      DIAG_LOOP(4);
      DIAG_T0();
      Trav T;
      reset_closest(T, MODE_CLOSEST);
      traverse<FEAT, KOPS, MODE_CLOSEST>(S, ray, T, C, stack, stride);
      nan_commit(T, C);
      DIAG_REGION(0);
      bool did_hit = T.best_prim != 0x7fffffff;
      if (COUNT && pm.digest) {
        unsigned long long tb = 0ull;
        if (did_hit) __builtin_memcpy(&tb, &T.best_t, 8);
        dg += rtc_hit_hash(rtc_hit_hash_base(tb, did_hit ? T.best_prim : -1, did_hit ? T.best_k : 0), fuel0 - fuel, kind);
      }
      if (first) {
        first = false;
        if (hit_t) {
          hit_t[q] = did_hit ? T.best_t : 0.0;
          hit_prim[q] = did_hit ? T.best_prim : -1;
          hit_k[q] = did_hit ? T.best_k : 0;
        }
      }
      if (did_hit) {
        const DPrim P = S.prims[T.best_prim];
        const double* M = S.mat + 8 * P.mat;
        const double ambient = M[0], diffuse = M[1], specular = M[2], shininess = M[3], reflective = M[4], transparency = M[5];
        State st;
        double hu, hv;
        hit_uv(S, P, ray, hu, hv);
        prepare_state(S, P, ray, T.best_t, hu, hv, st);

        // n1 / n2 / reflectance are only consumed when the surface is transparent (src/world.rs:70-78, :110)
        double n1 = 1.0, n2 = 1.0;
        if (transparency != 0.0 && fuel > 0) {
          n_container++;
          Trav K = T;  // keeps the hit key (thi = best_t, best_prim, best_klast)
          K.mode = MODE_CONTAINERS;
          K.tlo = -DINF; K.thi = T.best_t;
          K.c1_prim = -1; K.c2_prim = -1; K.c1_t = 0.0; K.c2_t = 0.0;
          traverse<FEAT, KOPS, MODE_CONTAINERS>(S, ray, K, C, stack, stride);
          if (K.c1_prim >= 0) n1 = S.mat[8 * S.prims[K.c1_prim].mat + 6];
          if (K.c2_prim >= 0) n2 = S.mat[8 * S.prims[K.c2_prim].mat + 6];
          DIAG_REGION(1);
        }

        // Pattern::color_at(material_inv * over_point) — identical for every light (src/shape.rs:437)
        double cr, cg, cb;
        {
          const double* mi = S.xf_matinv + 16 * P.xform;
          double x = mi[0] * st.px + mi[1] * st.py + mi[2] * st.pz + mi[3] * 1.0;
          double y = mi[4] * st.px + mi[5] * st.py + mi[6] * st.pz + mi[7] * 1.0;
          double z = mi[8] * st.px + mi[9] * st.py + mi[10] * st.pz + mi[11] * 1.0;
          double w = mi[12] * st.px + mi[13] * st.py + mi[14] * st.pz + mi[15] * 1.0;
          const DPat& root = S.pats[S.mat_pattern[P.mat]];
          if (LEAN || root.tag == 1) { cr = root.color[0]; cg = root.color[1]; cb = root.color[2]; }
          else pattern_color(S, S.mat_pattern[P.mat], x, y, z, w, cr, cg, cb);
        }

        const bool blend = !LEAN && reflective > 0.0 && transparency > 0.0;  // (LEAN: DScene.no_glass_mirror)
        double R = 0.0;
        if (blend) R = blend_reflectance(st, n1, n2, fuel, cr, cg, cb);

        DIAG_REGION(2);
        // World::shade_hit (src/world.rs:50-82): per light, shadow test + Phong (src/shape.rs:429-462)
        double sr = 0.0, sg = 0.0, sb = 0.0;
        for (int l = 0; l < S.n_lights; l++) {
          DIAG_LOOP(5);
          const double* LG = S.lights + 6 * l;
          double vx = LG[3] - st.px, vy = LG[4] - st.py, vz = LG[5] - st.pz;
          n_shadow++;
          if (light_is_behind(S, vx, vy, vz, st.nx, st.ny, st.nz)) {  // ambient term only, in the expression of the general case
            const double lr = (cr * LG[0]) * ambient, lg = (cg * LG[1]) * ambient, lb = (cb * LG[2]) * ambient;
            sr += (lr + 0.0) + 0.0; sg += (lg + 0.0) + 0.0; sb += (lb + 0.0) + 0.0;
            continue;
          }
          double distance = sqrt(vx * vx + vy * vy + vz * vz);
          Ray sray;
          sray.ox = st.px; sray.oy = st.py; sray.oz = st.pz;
          sray.dx = vx / distance; sray.dy = vy / distance; sray.dz = vz / distance;
          Trav Sh;
          reset_closest(Sh, S.all_cast_shadow ? MODE_SHADOW_ANY : MODE_SHADOW_CLOSEST);
          if (S.all_cast_shadow) Sh.thi = distance;
          Sh.light = l; Sh.c1_t = distance;
          DIAG_T0();
          traverse<FEAT, KOPS>(S, sray, Sh, C, stack, stride);
          nan_commit(Sh, C);
          DIAG_REGION(3);
          bool shadowed;
          if (S.all_cast_shadow) shadowed = Sh.shadowed != 0;
          else shadowed = (Sh.best_prim != 0x7fffffff) && (S.prims[Sh.best_prim].flags & 1u) && (Sh.best_t < distance);

          double er = cr * LG[0], eg = cg * LG[1], eb = cb * LG[2];  // effective_color
          double lr = er * ambient, lg = eg * ambient, lb = eb * ambient;
          // light vector: (light.origin - point).normalize() — same numbers as the shadow ray direction
          double ldn = sray.dx * st.nx + sray.dy * st.ny + sray.dz * st.nz;
          double dr = 0.0, dg = 0.0, db = 0.0, pr = 0.0, pg = 0.0, pb = 0.0;
          if (!shadowed && ldn >= 0.0) {
            dr = er * diffuse * ldn; dg = eg * diffuse * ldn; db = eb * diffuse * ldn;
            // reflect = (-light).reflect(normal)
            double mlx = -sray.dx, mly = -sray.dy, mlz = -sray.dz;
            double d2 = 2.0 * (mlx * st.nx + mly * st.ny + mlz * st.nz);
            double rfx = mlx - st.nx * d2, rfy = mly - st.ny * d2, rfz = mlz - st.nz * d2;
            double rde = rfx * st.ex + rfy * st.ey + rfz * st.ez;
            if (rde > 0.0) {
              double f = specular_factor(rde, shininess, specular);
              pr = LG[0] * specular * f; pg = LG[1] * specular * f; pb = LG[2] * specular * f;
            }
          }
          sr += (lr + dr) + pr; sg += (lg + dg) + pg; sb += (lb + db) + pb;
        }
        acc_r += weight * sr; acc_g += weight * sg; acc_b += weight * sb;
        DIAG_T0();

        // reflected_color / refracted_color (src/world.rs:84-132), once per light in the reference -> factor L
        if (fuel > 0) {
          bool do_refl = reflective != 0.0;
          bool do_refr = transparency != 0.0;
          double wr = weight * L * reflective, wt = weight * L * transparency;
          if (blend) {
            wr *= R;
            wt *= (1.0 - R);
          }
          double tdx = 0.0, tdy = 0.0, tdz = 0.0;
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
          // depth-first: the reflection ray (if any) is traced next; only a refraction ray that has to wait is stacked
          if (!LEAN && do_refr && do_refl) {
            Pending& p = pend[np++];
            p.ox = st.ux; p.oy = st.uy; p.oz = st.uz; p.dx = tdx; p.dy = tdy; p.dz = tdz;
            p.weight = wt; p.fuel = fuel - 1; p.kind = 2;
          }
          if (do_refl) {
            ray.ox = st.px; ray.oy = st.py; ray.oz = st.pz; ray.dx = st.rx; ray.dy = st.ry; ray.dz = st.rz;
            weight = wr; fuel = fuel - 1; kind = 1;
            continue;
          }
          if (do_refr) {
            ray.ox = st.ux; ray.oy = st.uy; ray.oz = st.uz; ray.dx = tdx; ray.dy = tdy; ray.dz = tdz;
            weight = wt; fuel = fuel - 1; kind = 2;
            continue;
          }
        }
      }
      DIAG_REGION(5);
      if (np == 0) {
        rgb[3 * q + 0] = acc_r;
        rgb[3 * q + 1] = acc_g;
        rgb[3 * q + 2] = acc_b;
        if (COUNT && pm.digest) pm.digest[q] = dg;
        break;
      }
      const Pending& p = pend[--np];
      ray.ox = p.ox; ray.oy = p.oy; ray.oz = p.oz; ray.dx = p.dx; ray.dy = p.dy; ray.dz = p.dz;
      weight = p.weight; fuel = p.fuel; kind = p.kind;
    }
  }

#ifdef RTC_DIAG
  atomicAdd(&s_diag[14], __builtin_amdgcn_s_memtime() - diag_k0);
  atomicAdd(&s_diag[15], 1ull);
  __syncthreads();
  if (threadIdx.x < 64 && s_diag[threadIdx.x]) atomicAdd(&stats->diag[threadIdx.x], s_diag[threadIdx.x]);
#endif
  if (COUNT || true) {
    // nan_ts must always be published (error reporting); the rest only in the counting variant
    if (C.nan_ts) atomicAdd(&stats->nan_ts, (unsigned long long)C.nan_ts);
  }
  if (COUNT) {
    atomicAdd(&stats->rays_primary, (unsigned long long)n_primary);
    atomicAdd(&stats->rays_shadow, (unsigned long long)n_shadow);
    atomicAdd(&stats->rays_reflect, (unsigned long long)n_reflect);
    atomicAdd(&stats->rays_refract, (unsigned long long)n_refract);
    atomicAdd(&stats->rays_container, (unsigned long long)n_container);
    atomicAdd(&stats->accel_nodes, (unsigned long long)C.accel_nodes);
    atomicAdd(&stats->group_tests, (unsigned long long)C.group_tests);
    atomicAdd(&stats->tri_tests, (unsigned long long)C.tri_tests);
    atomicAdd(&stats->analytic_tests, (unsigned long long)C.analytic_tests);
    atomicAdd(&stats->knodes, (unsigned long long)C.knodes);
    atomicAdd(&stats->kplanes, (unsigned long long)C.kplanes);
    atomicAdd(&stats->light_cells, (unsigned long long)C.light_cells);
    atomicAdd(&stats->kgroups, (unsigned long long)C.kgroups);
  }
}

// =================================================================================================================
// Wavefront path (DWave in device_scene.h): the same device functions as rtc_trace_kernel, cut into per-level kernels.
//   wf_ts      trace role: closest hit (+ container pass for transparent hits) of every ray of level d;
//              shadow role: per shade record of level d - 1 and light, shadow ray + Phong terms -> the ray's colour contribution
//   wf_shade   hit state, pattern colour -> shade record; reflected / refracted rays -> the other queue
//   wf_gather  pixel = its ray tree's contributions, added in the one-kernel path's order (bit-identical results)
// Every kernel loops over counts that live in device memory, so a frame is enqueued without host syncs.
// =================================================================================================================
namespace {

__device__ __forceinline__ unsigned wf_count(const DWave& W, int level, unsigned n0) {
  if (level == 0) return n0;
  unsigned c = W.counts[level];
  return c < W.cap ? c : W.cap;
}
__device__ __forceinline__ Ray wf_load_ray(const DWave& W, int level, unsigned i, double& weight) {
  const double* q = W.rq[level & 1];
  const size_t cap = W.cap;
  Ray r;
  r.ox = q[i]; r.oy = q[cap + i]; r.oz = q[2 * cap + i]; r.dx = q[3 * cap + i]; r.dy = q[4 * cap + i]; r.dz = q[5 * cap + i];
  weight = q[6 * cap + i];
  return r;
}

}  // namespace

// wf_ts work item of the trace role: ray i of `level` (closest hit, container pass for transparent hits).
template <int FEAT, bool KOPS, bool LDSC>
__device__ __forceinline__ void wf_trace_ray(const DScene& S, const DCamera& cam, const DPixelMap& pm, const DWave& W, const WorkMap& wm, int level, unsigned i,
                                             double* __restrict__ hit_t, int* __restrict__ hit_prim, int* __restrict__ hit_k, int* stack, int stride, Counters& C,
                                             unsigned& n_rays, unsigned& n_container, int fuel_left, const LdsScene& L, bool digest = false) {
  const size_t cap = W.cap;
  int32_t* ch = W.child + (size_t)level * 2 * cap;
  Ray ray;
  uint64_t q = 0;
  if (level == 0) {
    if (!work_to_slot(wm, i, q)) { W.h_prim[i] = -1; return; }  // tile padding: no pixel gathers this id
    ray = slot_ray(pm, cam, q);
  } else {
    double w_;
    ray = wf_load_ray(W, level, i, w_);
  }
  n_rays++;
  Trav T;
  reset_closest(T, MODE_CLOSEST);
  DIAG_LOOP(17);
  {
    DIAG_SPAN_BEGIN();
    traverse<FEAT, KOPS, MODE_CLOSEST, LDSC>(S, ray, T, C, stack, stride, L);
    nan_commit(T, C);
    DIAG_SPAN_END(0);
  }
  const bool did_hit = T.best_prim != 0x7fffffff;
  if (level == 0 && hit_t) {
    hit_t[q] = did_hit ? T.best_t : 0.0;
    hit_prim[q] = did_hit ? T.best_prim : -1;
    hit_k[q] = did_hit ? T.best_k : 0;
  }
  W.h_prim[i] = did_hit ? T.best_prim : -1;
  if (digest) {
    unsigned long long tb = 0ull;
    if (did_hit) __builtin_memcpy(&tb, &T.best_t, 8);
    W.dig[(size_t)level * cap + i] = rtc_hit_hash_base(tb, did_hit ? T.best_prim : -1, did_hit ? T.best_k : 0);
  }
  // child links: none yet; a miss is marked as such (the gather then knows that no contribution was written for this ray)
  ch[i] = did_hit ? -1 : RTC_WF_MISS; ch[cap + i] = -1;
  if (!did_hit) return;
  W.h_t[i] = T.best_t;
  // n1 / n2 are only consumed — and only stored — when the surface is transparent and the hit can still spawn rays
  // (src/world.rs:70-78, :110); wf_shade reads them under the same condition
  if (fuel_left > 0 && S.mat[8 * S.prims[T.best_prim].mat + 5] != 0.0) {
    double n1 = 1.0, n2 = 1.0;
    n_container++;
    Trav K = T;  // keeps the hit key (thi = best_t, best_prim, best_klast)
    K.mode = MODE_CONTAINERS;
    K.tlo = -DINF; K.thi = T.best_t;
    K.c1_prim = -1; K.c2_prim = -1; K.c1_t = 0.0; K.c2_t = 0.0;
    DIAG_LOOP(19);
    {
      DIAG_SPAN_BEGIN();
#pragma unroll 1
      for (int rep_ = 0; rep_ < ((RTC_PROBE & 32) ? 2 : 1); rep_++) {
        Ray r2 = ray;
        if (RTC_PROBE & 32) RTC_LAUNDER(r2.ox);
        traverse<FEAT, KOPS, MODE_CONTAINERS, LDSC>(S, r2, K, C, stack, stride, L);
      }
      DIAG_SPAN_END(1);
    }
    if (K.c1_prim >= 0) n1 = S.mat[8 * S.prims[K.c1_prim].mat + 6];
    if (K.c2_prim >= 0) n2 = S.mat[8 * S.prims[K.c2_prim].mat + 6];
    W.h_n12[i] = n1; W.h_n12[cap + i] = n2;
  }
}

// wf_ts work item of the shadow role: shade record s of `level` (per light: shadow ray, then the Phong terms).
template <int FEAT, bool KOPS, bool LDSC>
__device__ __forceinline__ void wf_shadow_rec(const DScene& S, const DCamera& cam, const DPixelMap& pm, const DWave& W, const WorkMap& wm, int level, unsigned s, int* stack,
                                              int stride, Counters& C, unsigned& n_shadow, const LdsScene& L) {
  const size_t cap = W.cap;
  double* cb = W.contrib + (size_t)level * 3 * cap;
  const double* r = W.sr;
  const double px = r[s], py = r[cap + s], pz = r[2 * cap + s];
  // World::shade_hit (src/world.rs:50-82): per light, shadow test + Phong (src/shape.rs:429-462).  First every shadow
  // ray (only the point is live across the traversals), then the Phong terms with the rest of the record.
  // (light_is_behind() is not used here: on the scenes this path serves few lights are behind their surfaces -- most hits are on the
  // floor -- and the mask's live range cost the traversal kernel 2.5 % at its register budget, whether or not a ray was skipped)
  unsigned long long shadow_mask = 0ull;  // <= 64 lights (checked at scene creation)
  for (int l = 0; l < S.n_lights; l++) {
    const double* LG = S.lights + 6 * l;
    double vx = LG[3] - px, vy = LG[4] - py, vz = LG[5] - pz;
    double distance = sqrt(vx * vx + vy * vy + vz * vz);
    Ray sray;
    sray.ox = px; sray.oy = py; sray.oz = pz;
    sray.dx = vx / distance; sray.dy = vy / distance; sray.dz = vz / distance;
    n_shadow++;
    Trav Sh;
    reset_closest(Sh, S.all_cast_shadow ? MODE_SHADOW_ANY : MODE_SHADOW_CLOSEST);
    if (S.all_cast_shadow) { Sh.thi = distance; Sh.unordered = 1; }
    Sh.light = l; Sh.c1_t = distance;
    DIAG_LOOP(18);
    DIAG_SPAN_BEGIN();
    if (S.all_cast_shadow) traverse<FEAT, KOPS, MODE_SHADOW_ANY, LDSC>(S, sray, Sh, C, stack, stride, L);  // mode: a compile-time constant in each
    else traverse<FEAT, KOPS, MODE_SHADOW_CLOSEST, LDSC>(S, sray, Sh, C, stack, stride, L);
    nan_commit(Sh, C);
    DIAG_SPAN_END(3);
    bool shadowed;
    if (S.all_cast_shadow) shadowed = Sh.shadowed != 0;
    else shadowed = (Sh.best_prim != 0x7fffffff) && (S.prims[Sh.best_prim].flags & 1u) && (Sh.best_t < distance);
    if (shadowed) shadow_mask |= 1ull << l;
  }
  DIAG_LOOP(20);
  DIAG_SPAN_BEGIN();
  const double nx = r[3 * cap + s], ny = r[4 * cap + s], nz = r[5 * cap + s];
  const double cr = r[6 * cap + s], cg = r[7 * cap + s], cbl = r[8 * cap + s];
  // the eye vector (= -direction, src/intersection.rs:56) and the path weight of the ray come from where the ray itself came
  // from: the level's queue (still intact: the next shading kernel is the first to overwrite it) or, at level 0, the camera
  const int node = W.sr_node[s];
  double ex, ey, ez, weight;
  if (level == 0) {
    uint64_t q = 0;
    (void)work_to_slot(wm, (unsigned)node, q);
    const Ray pr = slot_ray(pm, cam, q);
    ex = -pr.dx; ey = -pr.dy; ez = -pr.dz; weight = 1.0;
  } else {
    const double* rq = W.rq[level & 1];
    ex = -rq[3 * cap + node]; ey = -rq[4 * cap + node]; ez = -rq[5 * cap + node]; weight = rq[6 * cap + node];
  }
  const double* M = S.mat + 8 * W.sr_mat[s];
  const double ambient = M[0], diffuse = M[1], specular = M[2], shininess = M[3];
  double sr = 0.0, sg = 0.0, sb = 0.0;
  for (int l = 0; l < S.n_lights * ((RTC_PROBE & 8) ? 2 : 1); l++) {
    if ((RTC_PROBE & 8) && l == S.n_lights) { sr = sg = sb = 0.0; }
    const int li = (RTC_PROBE & 8) ? l % S.n_lights : l;
    const double* LG = S.lights + 6 * li;
    double vx = LG[3] - px, vy = LG[4] - py, vz = LG[5] - pz;
    if (RTC_PROBE & 8) RTC_LAUNDER(vx);
    double distance = sqrt(vx * vx + vy * vy + vz * vz);
    const double ldx = vx / distance, ldy = vy / distance, ldz = vz / distance;  // the shadow ray's direction again
    const bool shadowed = (shadow_mask >> li) & 1ull;
    double er = cr * LG[0], eg = cg * LG[1], eb = cbl * LG[2];  // effective_color
    double lr = er * ambient, lg = eg * ambient, lb = eb * ambient;
    // light vector: (light.origin - point).normalize() — same numbers as the shadow ray direction
    double ldn = ldx * nx + ldy * ny + ldz * nz;
    double dr = 0.0, dg = 0.0, db = 0.0, pr = 0.0, pg = 0.0, pb = 0.0;
    if (!shadowed && ldn >= 0.0) {
      dr = er * diffuse * ldn; dg = eg * diffuse * ldn; db = eb * diffuse * ldn;
      // reflect = (-light).reflect(normal)
      double mlx = -ldx, mly = -ldy, mlz = -ldz;
      double d2 = 2.0 * (mlx * nx + mly * ny + mlz * nz);
      double rfx = mlx - nx * d2, rfy = mly - ny * d2, rfz = mlz - nz * d2;
      double rde = rfx * ex + rfy * ey + rfz * ez;
      if (rde > 0.0) {
        double f = specular_factor(rde, shininess, specular);
        pr = LG[0] * specular * f; pg = LG[1] * specular * f; pb = LG[2] * specular * f;
      }
    }
    sr += (lr + dr) + pr; sg += (lg + dg) + pg; sb += (lb + db) + pb;
  }
  cb[node] = weight * sr; cb[cap + node] = weight * sg; cb[2 * cap + node] = weight * sb;
  DIAG_SPAN_END(5);
}

// Traversal kernel of the wavefront path: the closest-hit pass of level `tl` and the shadow + lighting pass of level `sl`
// (either may be -1) as ONE launch, so the two independent passes fill the chip together.  Work is handed out in chunks of
// RTC_WF_CHUNK items from per-XCD counters (block b belongs to XCD b % 8 and takes chunks b % 8, b % 8 + 8, ...): trace
// chunks first (the next shading kernel waits for them), then shadow chunks; a wave that finishes early simply takes more.
#ifndef RTC_WF_TS_WAVES
#define RTC_WF_TS_WAVES 3  // waves per SIMD the traversal kernel is compiled for: 168 VGPRs, no scratch.  At 4 (128 VGPRs) the ray and the
                           // leaf record spill around every leaf test: same frame time, +1.7 GB of HBM traffic per frame (profiles/r2_*)
#endif
#ifndef RTC_WF_TS_WAVES_MAXFEAT
#define RTC_WF_TS_WAVES_MAXFEAT 1  // feature levels up to this one are compiled for RTC_WF_TS_WAVES waves per SIMD, the others for 2
#endif
#ifndef RTC_LDS_BLOCK
#define RTC_LDS_BLOCK (256 * RTC_WF_TS_WAVES)  // LDSC kernels: one block per CU with all the waves the register budget allows
#endif
template <bool COUNT, int FEAT, bool KOPS, bool LDSC = false>
__global__ void __launch_bounds__(LDSC ? RTC_LDS_BLOCK : RTC_BLOCK, FEAT <= RTC_WF_TS_WAVES_MAXFEAT ? RTC_WF_TS_WAVES : 2) wf_ts(DScene S, DCamera cam, DPixelMap pm, DWave W, int tl, int sl, unsigned n0, int slot, int fuel_left,
                                                                     double* __restrict__ hit_t, int* __restrict__ hit_prim, int* __restrict__ hit_k, DStats* __restrict__ stats) {
  RTC_LDS_STACK(lds_stack);
  LdsScene L = {nullptr, nullptr, nullptr, nullptr, 0, 0, 0};
  int* stack = lds_stack + threadIdx.x;
  int stride = RTC_BLOCK;
#ifndef RTC_EMU
  if (LDSC) {
    // dynamic LDS: [scene tables][blockDim x bvh_stack stack words]; the block copies the tables in
    stack = (int*)lds_fill(S, (char*)lds_stack, L) + threadIdx.x;
    stride = (int)blockDim.x;
  }
#endif
  Counters C = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned n_rays = 0, n_container = 0, n_shadow = 0;
#ifdef RTC_DIAG
  if (threadIdx.x < 64) s_diag[threadIdx.x] = 0ull;
  __syncthreads();
#endif
  const WorkMap wm = make_workmap(pm, cam);
  const unsigned nt = tl >= 0 ? wf_count(W, tl, n0) : 0u;
  unsigned ns = sl >= 0 ? W.counts[RTC_WF_SHADE_COUNT + sl] : 0u;
  if (ns > W.cap) ns = W.cap;
  const unsigned ct = (nt + RTC_WF_CHUNK - 1) / RTC_WF_CHUNK, cs = (ns + RTC_WF_CHUNK - 1) / RTC_WF_CHUNK;
  const unsigned nx = gridDim.x < 8u ? gridDim.x : 8u;  // chunk residues in use (a grid smaller than 8 blocks has fewer)
  const unsigned x = blockIdx.x % nx;
  unsigned* next = &W.counts[RTC_WF_CHUNK_NEXT + 32 * (8 * slot + (int)x)];
  const int lane = RTC_LANE_ID;
  for (;;) {
    unsigned kx = 0;
    if (lane == 0) kx = atomicAdd(next, 1u);
    kx = __shfl(kx, 0);
    const unsigned long long chunk = (unsigned long long)kx * nx + x;
    if (chunk >= (unsigned long long)ct + cs) break;
    if (chunk < ct) {
      const unsigned base = (unsigned)chunk * RTC_WF_CHUNK;
      for (unsigned o = 0; o < RTC_WF_CHUNK; o += RTC_WF_LANES) {
        const unsigned i = base + o + (unsigned)lane;
        if (i < nt) {
          DIAG_SPAN_BEGIN();
          wf_trace_ray<FEAT, KOPS, LDSC>(S, cam, pm, W, wm, tl, i, hit_t, hit_prim, hit_k, stack, stride, C, n_rays, n_container, fuel_left, L, COUNT && W.dig != nullptr);
          DIAG_SPAN_END(7);
        }
      }
    } else {
      const unsigned base = (unsigned)(chunk - ct) * RTC_WF_CHUNK;
      for (unsigned o = 0; o < RTC_WF_CHUNK; o += RTC_WF_LANES) {
        const unsigned s = base + o + (unsigned)lane;
        if (s < ns) {
          DIAG_SPAN_BEGIN();
          wf_shadow_rec<FEAT, KOPS, LDSC>(S, cam, pm, W, wm, sl, s, stack, stride, C, n_shadow, L);
          DIAG_SPAN_END(7);
        }
      }
    }
  }
  if (C.nan_ts) atomicAdd(&stats->nan_ts, (unsigned long long)C.nan_ts);
#ifdef RTC_DIAG
  __syncthreads();
  if (threadIdx.x < 64 && s_diag[threadIdx.x]) atomicAdd(&stats->diag[threadIdx.x], s_diag[threadIdx.x]);
#endif
  if (COUNT) {
    if (tl == 0) atomicAdd(&stats->rays_primary, (unsigned long long)n_rays);
    atomicAdd(&stats->rays_container, (unsigned long long)n_container);
    atomicAdd(&stats->rays_shadow, (unsigned long long)n_shadow);
    atomicAdd(&stats->accel_nodes, (unsigned long long)C.accel_nodes);
    atomicAdd(&stats->group_tests, (unsigned long long)C.group_tests);
    atomicAdd(&stats->tri_tests, (unsigned long long)C.tri_tests);
    atomicAdd(&stats->analytic_tests, (unsigned long long)C.analytic_tests);
    atomicAdd(&stats->knodes, (unsigned long long)C.knodes);
    atomicAdd(&stats->kplanes, (unsigned long long)C.kplanes);
    atomicAdd(&stats->light_cells, (unsigned long long)C.light_cells);
    atomicAdd(&stats->kgroups, (unsigned long long)C.kgroups);
  }
}
