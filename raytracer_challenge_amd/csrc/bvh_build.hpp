// bvh_build.hpp — host-side binned-SAH builder for the results-neutral accelerator.
// Input: one f64 AABB per item.  Output: 64-byte DBvhNode records (children's boxes, f32 rounded outward after
// a relative pad) appended to `nodes`, and the leaf-order permutation of the items.  No reference counterpart:
// the reference tests every child of a group linearly (src/shape.rs:254-256).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <thread>
#include <vector>

#include "device_scene.h"

namespace bvh {

struct Item { double lo[3], hi[3]; };

struct Builder {
  const std::vector<Item>& items;
  std::vector<DBvhNode>& nodes;
  std::vector<uint32_t>& order;  // leaf-order item ids (local), appended; leaf refs index order positions + base
  uint32_t base;                 // global offset of this BVH's first packed item
  int max_depth = 0;
  bool median_only = false;  // fallback when SAH produced a tree deeper than the traversal stack
  double center[3] = {0, 0, 0};  // node boxes are stored relative to this point (the BVH's frame, see slab32 on the device)
  static constexpr int kBins = 16;
  int kLeaf = 4;  // max items per leaf (<= 8: 3 bits in the leaf ref)

  struct Range { double lo[3], hi[3]; };
  static void grow(Range& r, const Item& it) {
    for (int a = 0; a < 3; a++) { r.lo[a] = std::min(r.lo[a], it.lo[a]); r.hi[a] = std::max(r.hi[a], it.hi[a]); }
  }
  static Range none() {
    const double inf = std::numeric_limits<double>::infinity();
    return {{inf, inf, inf}, {-inf, -inf, -inf}};
  }
  static double area(const Range& r) {
    double dx = r.hi[0] - r.lo[0], dy = r.hi[1] - r.lo[1], dz = r.hi[2] - r.lo[2];
    if (!(dx >= 0 && dy >= 0 && dz >= 0)) return 0.0;
    return 2.0 * (dx * dy + dy * dz + dz * dx);
  }
  static float down(double v) {
    float f = (float)v;
    if ((double)f > v) f = std::nextafterf(f, -std::numeric_limits<float>::infinity());
    return f;
  }
  static float up(double v) {
    float f = (float)v;
    if ((double)f < v) f = std::nextafterf(f, std::numeric_limits<float>::infinity());
    return f;
  }
  void store(const Range& r, float* lo, float* hi) const {
    double ext = 0.0;
    for (int a = 0; a < 3; a++) ext = std::max(ext, r.hi[a] - r.lo[a]);
    for (int a = 0; a < 3; a++) {
      double pl = 1e-9 * (std::fabs(r.lo[a]) + ext) + 1e-30, ph = 1e-9 * (std::fabs(r.hi[a]) + ext) + 1e-30;
      lo[a] = down((r.lo[a] - pl) - center[a]);
      hi[a] = up((r.hi[a] + ph) - center[a]);
    }
  }
  static void store_absent(float* lo, float* hi) {
    for (int a = 0; a < 3; a++) { lo[a] = std::numeric_limits<float>::infinity(); hi[a] = -std::numeric_limits<float>::infinity(); }
  }

  int par = 1;                                 // threads this subtree may use
  static constexpr size_t kParMin = 16384;     // smaller ranges are built by the calling thread
  // Appends a subtree built on private vectors (base 0): node refs shift by the node offset, leaf refs by the item offset.
  int32_t append(const Builder& sub, int32_t ref) {
    const int32_t node_off = (int32_t)nodes.size();
    const uint32_t item_off = base + (uint32_t)order.size();
    auto fix = [&](int32_t c) -> int32_t {
      if (c >= 0) return c + node_off;
      const uint32_t u = (uint32_t)~c;
      return ~(int32_t)((((u >> 3) + item_off) << 3) | (u & 7u));
    };
    const size_t at = nodes.size();
    nodes.insert(nodes.end(), sub.nodes.begin(), sub.nodes.end());
    for (size_t i = at; i < nodes.size(); i++) {
      // an absent child (inverted box) carries a copy of its sibling's ref: shifting it too keeps it harmless
      nodes[i].c0 = fix(nodes[i].c0);
      nodes[i].c1 = fix(nodes[i].c1);
    }
    order.insert(order.end(), sub.order.begin(), sub.order.end());
    return fix(ref);
  }

  // Builds the subtree over ids[b,e); returns a child ref (>= 0 node, < 0 leaf) and its bounds.
  int32_t build(std::vector<uint32_t>& ids, size_t b, size_t e, int depth, Range* bounds) {
    max_depth = std::max(max_depth, depth);
    Range box = none(), cbox = none();
    for (size_t i = b; i < e; i++) {
      grow(box, items[ids[i]]);
      const Item& it = items[ids[i]];
      for (int a = 0; a < 3; a++) {
        double c = 0.5 * (it.lo[a] + it.hi[a]);
        cbox.lo[a] = std::min(cbox.lo[a], c);
        cbox.hi[a] = std::max(cbox.hi[a], c);
      }
    }
    *bounds = box;
    size_t n = e - b;
    if (n <= (size_t)kLeaf) {
      uint32_t first = base + (uint32_t)order.size();
      for (size_t i = b; i < e; i++) order.push_back(ids[i]);
      return ~(int32_t)((first << 3) | (uint32_t)(n - 1));
    }
    // binned SAH on the widest-centroid axes
    int best_axis = -1, best_split = -1;
    double best_cost = std::numeric_limits<double>::infinity();
    for (int a = 0; a < 3 && !median_only; a++) {
      double lo = cbox.lo[a], ext = cbox.hi[a] - cbox.lo[a];
      if (!(ext > 0.0) || !std::isfinite(ext)) continue;
      Range bb[kBins];
      size_t cnt[kBins] = {0};
      for (auto& r : bb) r = none();
      double scale = kBins / ext;
      for (size_t i = b; i < e; i++) {
        const Item& it = items[ids[i]];
        int k = (int)((0.5 * (it.lo[a] + it.hi[a]) - lo) * scale);
        k = std::min(std::max(k, 0), kBins - 1);
        cnt[k]++;
        grow(bb[k], it);
      }
      double right_area[kBins];
      size_t right_cnt[kBins];
      Range acc = none();
      size_t c = 0;
      for (int k = kBins - 1; k > 0; k--) {
        if (cnt[k]) { for (int q = 0; q < 3; q++) { acc.lo[q] = std::min(acc.lo[q], bb[k].lo[q]); acc.hi[q] = std::max(acc.hi[q], bb[k].hi[q]); } }
        c += cnt[k];
        right_area[k] = area(acc);
        right_cnt[k] = c;
      }
      acc = none();
      c = 0;
      for (int k = 0; k < kBins - 1; k++) {
        if (cnt[k]) { for (int q = 0; q < 3; q++) { acc.lo[q] = std::min(acc.lo[q], bb[k].lo[q]); acc.hi[q] = std::max(acc.hi[q], bb[k].hi[q]); } }
        c += cnt[k];
        if (c == 0 || right_cnt[k + 1] == 0) continue;
        double cost = area(acc) * (double)c + right_area[k + 1] * (double)right_cnt[k + 1];
        if (cost < best_cost) { best_cost = cost; best_axis = a; best_split = k; }
      }
    }
    size_t mid;
    if (best_axis >= 0) {
      double lo = cbox.lo[best_axis], scale = kBins / (cbox.hi[best_axis] - cbox.lo[best_axis]);
      auto it = std::partition(ids.begin() + b, ids.begin() + e, [&](uint32_t id) {
        int k = (int)((0.5 * (items[id].lo[best_axis] + items[id].hi[best_axis]) - lo) * scale);
        k = std::min(std::max(k, 0), kBins - 1);
        return k <= best_split;
      });
      mid = (size_t)(it - ids.begin());
    } else if (median_only) {  // median of the widest centroid axis
      int a = 0;
      for (int q = 1; q < 3; q++)
        if (cbox.hi[q] - cbox.lo[q] > cbox.hi[a] - cbox.lo[a]) a = q;
      mid = b + n / 2;
      std::nth_element(ids.begin() + b, ids.begin() + mid, ids.begin() + e, [&](uint32_t x, uint32_t y) {
        return items[x].lo[a] + items[x].hi[a] < items[y].lo[a] + items[y].hi[a];
      });
    } else {
      mid = b;
    }
    if (mid == b || mid == e) {  // all centroids coincide (or non-finite): split by count
      mid = b + n / 2;
    }
    int32_t self = (int32_t)nodes.size();
    nodes.emplace_back();
    Range r0, r1;
    int32_t c0, c1;
    if (par > 1 && n >= kParMin) {
      // both halves on private vectors (the left one in another thread; ids[b, mid) and ids[mid, e) are disjoint), then
      // appended left first: node numbering and leaf order are exactly those of the serial recursion
      std::vector<DBvhNode> ln, rn;
      std::vector<uint32_t> lo_, ro_;
      Builder L{items, ln, lo_, 0}, R{items, rn, ro_, 0};
      for (Builder* q : {&L, &R}) { q->median_only = median_only; q->kLeaf = kLeaf; for (int a = 0; a < 3; a++) q->center[a] = center[a]; }
      L.par = par / 2;
      R.par = par - par / 2;
      int32_t lc = 0, rc = 0;
      std::thread th([&]() { lc = L.build(ids, b, mid, depth + 1, &r0); });
      rc = R.build(ids, mid, e, depth + 1, &r1);
      th.join();
      c0 = append(L, lc);
      c1 = append(R, rc);
      max_depth = std::max(max_depth, std::max(L.max_depth, R.max_depth));
    } else {
      c0 = build(ids, b, mid, depth + 1, &r0);
      c1 = build(ids, mid, e, depth + 1, &r1);
    }
    DBvhNode& N = nodes[self];
    store(r0, N.lo0, N.hi0);
    store(r1, N.lo1, N.hi1);
    N.c0 = c0;
    N.c1 = c1;
    N.pad[0] = N.pad[1] = 0;
    return self;
  }
};

// A builder of the same binary tree on the device (bvh_device.hip: LBVH); returns the root or -1 (then the host builds).
typedef int32_t (*DeviceBuildFn)(const std::vector<Item>& items, std::vector<DBvhNode>& nodes, std::vector<uint32_t>& order, uint32_t base, int leaf_max, double* frame);

// Returns the root node index (always an inner node, so traversal can start from a node).
// frame[4] receives the BVH's centre (x, y, z) and the inf-norm radius of its bounds around that centre.
inline int32_t build(const std::vector<Item>& items, std::vector<DBvhNode>& nodes, std::vector<uint32_t>& order, uint32_t base, int* depth, bool median_only = false,
                     int leaf_max = 4, double* frame = nullptr) {
  Builder B{items, nodes, order, base};
  B.median_only = median_only;
  {
    Builder::Range all = Builder::none();
    for (const Item& it : items) Builder::grow(all, it);
    double rad = 0.0;
    for (int a = 0; a < 3; a++) {
      B.center[a] = items.empty() ? 0.0 : 0.5 * (all.lo[a] + all.hi[a]);
      if (!std::isfinite(B.center[a])) B.center[a] = 0.0;
      if (!items.empty()) rad = std::max(rad, std::max(std::fabs(all.hi[a] - B.center[a]), std::fabs(all.lo[a] - B.center[a])));
    }
    if (frame) { frame[0] = B.center[0]; frame[1] = B.center[1]; frame[2] = B.center[2]; frame[3] = rad * (1.0 + 1e-6) + 1e-30; }
  }
  B.kLeaf = leaf_max < 1 ? 1 : (leaf_max > 8 ? 8 : leaf_max);
  {
    const char* e = std::getenv("RTC_BUILD_THREADS");
    unsigned hw = std::thread::hardware_concurrency();
    int t = e ? std::atoi(e) : (int)(hw ? (hw > 8 ? 8 : hw) : 1);  // default: at most 8 (one process per GPU shares the host)
    B.par = t < 1 ? 1 : (t > 32 ? 32 : t);
  }
  std::vector<uint32_t> ids(items.size());
  for (uint32_t i = 0; i < ids.size(); i++) ids[i] = i;
  Builder::Range r;
  int32_t ref = B.build(ids, 0, ids.size(), 1, &r);
  if (ref < 0) {  // a single leaf: wrap it so the root is a node
    int32_t self = (int32_t)nodes.size();
    nodes.emplace_back();
    DBvhNode& N = nodes[self];
    B.store(r, N.lo0, N.hi0);
    Builder::store_absent(N.lo1, N.hi1);
    N.c0 = ref;
    N.c1 = ref;
    N.pad[0] = N.pad[1] = 0;
    ref = self;
  }
  if (depth) *depth = B.max_depth + 1;
  return ref;
}

// Folds the binary tree `n2` (root `root2`, indices local to n2) into 4-wide nodes appended to `out`: a node starts from
// its two children and, while it has a free slot, replaces the inner child with the largest box by that child's two
// children.  Boxes are copied, not recomputed (already rounded outward).  Returns the new root (global index in `out`).
// *depth = levels of 4-wide nodes; *stack_need = worst-case entries on the traversal stack (every visited node may
// push all its other children: need(node) = children - 1 + max need(child); a leaf needs none).
inline int32_t collapse4(const std::vector<DBvhNode>& n2, int32_t root2, std::vector<DBvhNode4>& out, int* depth, int* stack_need) {
  struct Child { float lo[3], hi[3]; int32_t ref; };
  struct Rec {
    const std::vector<DBvhNode>& n2;
    std::vector<DBvhNode4>& out;
    int max_depth = 0;
    static bool present(const float* lo, const float* hi) { return lo[0] <= hi[0] && lo[1] <= hi[1] && lo[2] <= hi[2]; }
    static double area(const Child& c) {
      double dx = (double)c.hi[0] - c.lo[0], dy = (double)c.hi[1] - c.lo[1], dz = (double)c.hi[2] - c.lo[2];
      return 2.0 * (dx * dy + dy * dz + dz * dx);
    }
    void children_of(int32_t i2, std::vector<Child>& v) const {
      const DBvhNode& N = n2[(size_t)i2];
      if (present(N.lo0, N.hi0)) { Child c; std::memcpy(c.lo, N.lo0, 12); std::memcpy(c.hi, N.hi0, 12); c.ref = N.c0; v.push_back(c); }
      if (present(N.lo1, N.hi1)) { Child c; std::memcpy(c.lo, N.lo1, 12); std::memcpy(c.hi, N.hi1, 12); c.ref = N.c1; v.push_back(c); }
    }
    int32_t fold(int32_t i2, int d, int* need) {
      max_depth = std::max(max_depth, d);
      std::vector<Child> ch;
      children_of(i2, ch);
      while (ch.size() < 4) {
        int best = -1;
        double best_area = -1.0;
        for (size_t k = 0; k < ch.size(); k++)
          if (ch[k].ref >= 0) {
            double a = area(ch[k]);
            if (!(a <= best_area)) { best_area = a; best = (int)k; }  // NaN / inf areas expand first
          }
        if (best < 0) break;
        std::vector<Child> sub;
        children_of(ch[(size_t)best].ref, sub);
        if (ch.size() - 1 + sub.size() > 4) break;
        ch.erase(ch.begin() + best);
        ch.insert(ch.end(), sub.begin(), sub.end());
      }
      int32_t self = (int32_t)out.size();
      out.emplace_back();
      int deepest = 0;
      int32_t refs[4] = {0, 0, 0, 0};
      for (size_t k = 0; k < ch.size(); k++) {
        int sub_need = 0;
        refs[k] = ch[k].ref >= 0 ? fold(ch[k].ref, d + 1, &sub_need) : ch[k].ref;
        deepest = std::max(deepest, sub_need);
      }
      DBvhNode4& N = out[(size_t)self];
      const float inf = std::numeric_limits<float>::infinity();
      for (size_t k = 0; k < 4; k++) {
        bool have = k < ch.size();
        N.lox[k] = have ? ch[k].lo[0] : inf; N.loy[k] = have ? ch[k].lo[1] : inf; N.loz[k] = have ? ch[k].lo[2] : inf;
        N.hix[k] = have ? ch[k].hi[0] : -inf; N.hiy[k] = have ? ch[k].hi[1] : -inf; N.hiz[k] = have ? ch[k].hi[2] : -inf;
        N.c[k] = have ? refs[k] : 0;
        N.pad[k] = 0;
      }
      *need = (ch.empty() ? 0 : (int)ch.size() - 1) + deepest;
      return self;
    }
  };
  Rec R{n2, out};
  int need = 0;
  int32_t r = R.fold(root2, 1, &need);
  if (depth) *depth = R.max_depth;
  if (stack_need) *stack_need = need;
  return r;
}

}  // namespace bvh
