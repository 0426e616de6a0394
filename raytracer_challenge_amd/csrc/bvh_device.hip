// bvh_device.hip — the accelerator's binary tree built ON THE DEVICE: a linear BVH (Morton order + Karras'
// radix tree) over the f64 item boxes, in the exact output format of the host's binned-SAH builder (bvh_build.hpp: 64-byte
// DBvhNode records with both children's boxes — f32, rounded outward after the same relative pad, relative to the BVH's centre —
// and the leaf-order permutation), so that everything after it (collapse to 4-wide nodes, stack bound, leaf packing) is shared
// and the host build is its checker: the accelerator is results-neutral, so a scene rendered through either tree
// gives the same hit records and pixels, bit for bit (tests/test_device_bvh.py).
//
// Steps: (1) 63-bit Morton code of every item's centroid inside the items' bounds; (2) radix sort of (code, item) pairs
// (hipCUB); (3) one thread per internal node: its key range and split (Karras 2012; ties broken by position); ranges of at most
// `leaf_max` items become leaves; (4) boxes bottom-up, the second child to arrive at a node merges (one atomic counter per
// node); (5) kept nodes are numbered by a prefix sum and written as DBvhNode records.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cmath>
#include <cstdint>
#include <limits>
#include <vector>

#include "bvh_build.hpp"

namespace {

struct DevBox { double lo[3], hi[3]; };

// Sort key of an item: its centroid's cell in a recursive bisection of the items' bounds that always halves the LONGEST side of
// the current cell (round 3).  A plain Morton code gives every axis every third bit whatever the bounds' shape; for a flat mesh
// (config 5: a heightfield 400 x 400 units wide and a few units high) a third of the radix tree's levels then split along the
// short axis, where nearly every triangle straddles the plane — children that overlap almost completely (config 5 rendered at
// 16.4 ms per frame against the host SAH tree's 10.9).  The bisection order (63 axis choices, two bits each in seq_lo / seq_hi,
// most significant key bit first) and the bits per axis come from the host (extent_bisection below).
__global__ void __launch_bounds__(256) morton_kernel(const DevBox* __restrict__ items, uint32_t n, double lox, double loy, double loz, double sx, double sy, double sz,
                                                     unsigned bx, unsigned by, unsigned bz, unsigned long long seq_lo, unsigned long long seq_hi,
                                                     unsigned long long* __restrict__ keys, uint32_t* __restrict__ vals) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const DevBox b = items[i];
  const double c[3] = {0.5 * (b.lo[0] + b.hi[0]), 0.5 * (b.lo[1] + b.hi[1]), 0.5 * (b.lo[2] + b.hi[2])};
  const double lo[3] = {lox, loy, loz}, s[3] = {sx, sy, sz};
  unsigned rem[3] = {bx, by, bz};
  unsigned long long q[3];
  for (int a = 0; a < 3; a++) {
    double t = (c[a] - lo[a]) * s[a];          // [0, 2^bits)
    const double top = (double)((1ull << rem[a]) - 1ull);
    if (!(t >= 0.0)) t = 0.0;                  // NaN / below
    if (t > top) t = top;
    q[a] = (unsigned long long)t;
  }
  unsigned long long key = 0ull;
  for (int k = 0; k < 63; k++) {
    const unsigned a = (unsigned)((k < 32 ? seq_lo >> (2 * k) : seq_hi >> (2 * (k - 32))) & 3ull);
    rem[a]--;
    key = (key << 1) | ((q[a] >> rem[a]) & 1ull);
  }
  keys[i] = key;
  vals[i] = i;
}

// length of the common prefix of the keys at sorted positions i and j (-1 outside); equal keys are told apart by position
__device__ __forceinline__ int delta(const unsigned long long* __restrict__ k, int n, int i, int j) {
  if (j < 0 || j >= n) return -1;
  const unsigned long long a = k[i], b = k[j];
  if (a != b) return __clzll((long long)(a ^ b));
  return 64 + __clz(i ^ j);
}

// Karras 2012, one thread per internal node i in [0, n - 2]: range [first, last], children, parents.
// child refs: >= 0 internal node, < 0 : ~item position.
__global__ void __launch_bounds__(256) radix_tree_kernel(const unsigned long long* __restrict__ keys, int n, int* __restrict__ left, int* __restrict__ right,
                                                         int* __restrict__ first_, int* __restrict__ last_, int* __restrict__ parent_node, int* __restrict__ parent_leaf) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (i >= n - 1) return;
  const int d = delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1) >= 0 ? 1 : -1;
  const int dmin = delta(keys, n, i, i - d);
  int lmax = 2;
  while (delta(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
  int l = 0;
  for (int t = lmax / 2; t >= 1; t /= 2)
    if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
  const int j = i + l * d;
  const int dnode = delta(keys, n, i, j);
  int s = 0;
  for (int t = (l + 1) / 2;; t = (t + 1) / 2) {
    if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
    if (t == 1) break;
  }
  const int gamma = i + s * d + (d < 0 ? -1 : 0);
  const int lo = i < j ? i : j, hi = i < j ? j : i;
  first_[i] = lo;
  last_[i] = hi;
  const int lc = lo == gamma ? ~gamma : gamma;
  const int rc = hi == gamma + 1 ? ~(gamma + 1) : gamma + 1;
  left[i] = lc;
  right[i] = rc;
  if (lc >= 0) parent_node[lc] = i; else parent_leaf[~lc] = i;
  if (rc >= 0) parent_node[rc] = i; else parent_leaf[~rc] = i;
  if (i == 0) parent_node[0] = -1;
}

__device__ __forceinline__ DevBox box_union(const DevBox& a, const DevBox& b) {
  DevBox r;
  for (int q = 0; q < 3; q++) { r.lo[q] = fmin(a.lo[q], b.lo[q]); r.hi[q] = fmax(a.hi[q], b.hi[q]); }
  return r;
}

// a box another thread has just written (behind its fence and the node's counter): read it from memory, not from a register copy
__device__ __forceinline__ DevBox load_fresh(const DevBox* p) {
  const volatile double* v = (const volatile double*)p;
  DevBox r;
  for (int q = 0; q < 3; q++) { r.lo[q] = v[q]; r.hi[q] = v[3 + q]; }
  return r;
}

// Bottom-up boxes: one thread per item walks towards the root; the first thread to reach a node leaves, the second merges.
__global__ void __launch_bounds__(256) fit_kernel(const DevBox* __restrict__ items, const uint32_t* __restrict__ vals, int n, const int* __restrict__ left, const int* __restrict__ right,
                                                  const int* __restrict__ parent_node, const int* __restrict__ parent_leaf, unsigned int* __restrict__ arrived,
                                                  DevBox* __restrict__ node_box) {
  const int k = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (k >= n) return;
  int p = parent_leaf[k];
  while (p >= 0) {
    __threadfence();
    if (atomicAdd(&arrived[p], 1u) == 0u) return;  // the sibling subtree is not finished: its thread will do this node
    __threadfence();
    const int lc = left[p], rc = right[p];
    const DevBox a = lc >= 0 ? load_fresh(node_box + lc) : items[vals[~lc]];
    const DevBox b = rc >= 0 ? load_fresh(node_box + rc) : items[vals[~rc]];
    node_box[p] = box_union(a, b);
    p = parent_node[p];
  }
}

__device__ __forceinline__ float round_down(double v) {
  float f = (float)v;
  if ((double)f > v) f = nextafterf(f, -__builtin_inff());
  return f;
}
__device__ __forceinline__ float round_up(double v) {
  float f = (float)v;
  if ((double)f < v) f = nextafterf(f, __builtin_inff());
  return f;
}
// bvh::Builder::store: relative pad, then outward rounding relative to the centre
__device__ __forceinline__ void store_box(const DevBox& r, double cx, double cy, double cz, float* lo, float* hi) {
  const double c[3] = {cx, cy, cz};
  double ext = 0.0;
  for (int a = 0; a < 3; a++) ext = fmax(ext, r.hi[a] - r.lo[a]);
  for (int a = 0; a < 3; a++) {
    const double pl = 1e-9 * (fabs(r.lo[a]) + ext) + 1e-30, ph = 1e-9 * (fabs(r.hi[a]) + ext) + 1e-30;
    lo[a] = round_down((r.lo[a] - pl) - c[a]);
    hi[a] = round_up((r.hi[a] + ph) - c[a]);
  }
}

__global__ void __launch_bounds__(256) keep_kernel(const int* __restrict__ first_, const int* __restrict__ last_, int n, int leaf_max, unsigned int* __restrict__ keep) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (i >= n - 1) return;
  keep[i] = (last_[i] - first_[i] + 1) > leaf_max ? 1u : 0u;  // smaller ranges are leaves of their parent
}

__global__ void __launch_bounds__(256) emit_kernel(const DevBox* __restrict__ items, const uint32_t* __restrict__ vals, int n, int leaf_max, uint32_t base, const int* __restrict__ left,
                                                   const int* __restrict__ right, const int* __restrict__ first_, const int* __restrict__ last_, const unsigned int* __restrict__ keep,
                                                   const unsigned int* __restrict__ newidx, const DevBox* __restrict__ node_box, double cx, double cy, double cz,
                                                   DBvhNode* __restrict__ out) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (i >= n - 1 || !keep[i]) return;
  DBvhNode N;
  const int ch[2] = {left[i], right[i]};
  int32_t refs[2];
  for (int q = 0; q < 2; q++) {
    const int c = ch[q];
    DevBox b;
    if (c >= 0) {
      b = node_box[c];
      if (keep[c]) refs[q] = (int32_t)newidx[c];
      else refs[q] = ~(int32_t)(((base + (uint32_t)first_[c]) << 3) | (uint32_t)(last_[c] - first_[c]));  // the whole range is one leaf
    } else {
      b = items[vals[~c]];
      refs[q] = ~(int32_t)(((base + (uint32_t)(~c)) << 3) | 0u);
    }
    store_box(b, cx, cy, cz, q == 0 ? N.lo0 : N.lo1, q == 0 ? N.hi0 : N.hi1);
  }
  N.c0 = refs[0];
  N.c1 = refs[1];
  N.pad[0] = N.pad[1] = 0;
  out[newidx[i]] = N;
}

// The 63 bisections of the key (see morton_kernel): always the longest side of the current cell; an axis without extent gets no bit.
struct Bisection { unsigned bits[3]; unsigned long long seq_lo, seq_hi; };
Bisection extent_bisection(const double ext[3]) {
  Bisection B = {{0, 0, 0}, 0ull, 0ull};
  double cell[3];
  for (int a = 0; a < 3; a++) cell[a] = ext[a] > 0.0 && std::isfinite(ext[a]) ? ext[a] : 0.0;
  for (int k = 0; k < 63; k++) {
    int a = 0;
    for (int c = 1; c < 3; c++) if (cell[c] > cell[a]) a = c;
    if (B.bits[a] >= 30) {  // (a side 2^30 times the others: give the bit to the next longest)
      int best = -1;
      for (int c = 0; c < 3; c++) if (B.bits[c] < 30 && (best < 0 || cell[c] > cell[best])) best = c;
      a = best;
    }
    B.bits[a]++;
    cell[a] *= 0.5;
    if (k < 32) B.seq_lo |= (unsigned long long)a << (2 * k); else B.seq_hi |= (unsigned long long)a << (2 * (k - 32));
  }
  return B;
}

struct Scratch {
  std::vector<void*> p;
  ~Scratch() { for (void* q : p) (void)hipFree(q); }
  template <class T> bool alloc(T** out, size_t n) {
    void* q = nullptr;
    if (hipMalloc(&q, (n ? n : 1) * sizeof(T)) != hipSuccess) return false;
    p.push_back(q);
    *out = (T*)q;
    return true;
  }
};

}  // namespace

// bvh::DeviceBuildFn (bvh_build.hpp).  Returns the root node (index into n2, always an inner node) or -1: the caller then
// builds on the host.  `order` receives the items in leaf order (appended), `frame` the centre and radius, as bvh::build does.
int32_t rtc_bvh_build_device(const std::vector<bvh::Item>& items, std::vector<DBvhNode>& n2, std::vector<uint32_t>& order, uint32_t base, int leaf_max, double* frame) {
  static_assert(sizeof(bvh::Item) == sizeof(DevBox), "item layout");
  const size_t n = items.size();
  if (n < 2 || (int)n <= leaf_max || n > 0x0fffffffu) return -1;
  // bounds and frame on the host (one pass over data the host already holds): same centre / radius rule as bvh::build
  bvh::Builder::Range all = bvh::Builder::none(), cb = bvh::Builder::none();
  for (const bvh::Item& it : items) {
    bvh::Builder::grow(all, it);
    for (int a = 0; a < 3; a++) { const double c = 0.5 * (it.lo[a] + it.hi[a]); cb.lo[a] = std::min(cb.lo[a], c); cb.hi[a] = std::max(cb.hi[a], c); }
  }
  double center[3], rad = 0.0, scale[3], ext3[3];
  for (int a = 0; a < 3; a++) ext3[a] = cb.hi[a] - cb.lo[a];
  const Bisection bis = extent_bisection(ext3);
  for (int a = 0; a < 3; a++) {
    center[a] = 0.5 * (all.lo[a] + all.hi[a]);
    if (!std::isfinite(center[a])) return -1;  // unbounded items: the host builder copes
    rad = std::max(rad, std::max(std::fabs(all.hi[a] - center[a]), std::fabs(all.lo[a] - center[a])));
    const double ext = cb.hi[a] - cb.lo[a];
    scale[a] = ext > 0.0 && std::isfinite(ext) ? (double)(1ull << bis.bits[a]) / ext * (1.0 - 1e-12) : 0.0;
  }
  Scratch S;
  DevBox *d_items = nullptr, *d_box = nullptr;
  unsigned long long *d_k0 = nullptr, *d_k1 = nullptr;
  uint32_t *d_v0 = nullptr, *d_v1 = nullptr;
  int *d_left = nullptr, *d_right = nullptr, *d_first = nullptr, *d_last = nullptr, *d_pn = nullptr, *d_pl = nullptr;
  unsigned int *d_arr = nullptr, *d_keep = nullptr, *d_new = nullptr;
  DBvhNode* d_out = nullptr;
  if (!S.alloc(&d_items, n) || !S.alloc(&d_box, n) || !S.alloc(&d_k0, n) || !S.alloc(&d_k1, n) || !S.alloc(&d_v0, n) || !S.alloc(&d_v1, n) || !S.alloc(&d_left, n) ||
      !S.alloc(&d_right, n) || !S.alloc(&d_first, n) || !S.alloc(&d_last, n) || !S.alloc(&d_pn, n) || !S.alloc(&d_pl, n) || !S.alloc(&d_arr, n) || !S.alloc(&d_keep, n) ||
      !S.alloc(&d_new, n) || !S.alloc(&d_out, n)) {
    (void)hipGetLastError();
    return -1;
  }
  if (hipMemcpy(d_items, items.data(), n * sizeof(DevBox), hipMemcpyHostToDevice) != hipSuccess) return -1;
  const unsigned blocks = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(morton_kernel, dim3(blocks), dim3(256), 0, 0, d_items, (uint32_t)n, cb.lo[0], cb.lo[1], cb.lo[2], scale[0], scale[1], scale[2], bis.bits[0], bis.bits[1], bis.bits[2],
                     bis.seq_lo, bis.seq_hi, d_k0, d_v0);
  size_t tmp_bytes = 0;
  if (hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, d_k0, d_k1, d_v0, d_v1, (int)n, 0, 63) != hipSuccess) return -1;
  unsigned char* d_tmp = nullptr;
  if (!S.alloc(&d_tmp, tmp_bytes)) return -1;
  if (hipcub::DeviceRadixSort::SortPairs(d_tmp, tmp_bytes, d_k0, d_k1, d_v0, d_v1, (int)n, 0, 63) != hipSuccess) return -1;
  if (hipMemset(d_arr, 0, n * sizeof(unsigned int)) != hipSuccess || hipMemset(d_keep, 0, n * sizeof(unsigned int)) != hipSuccess) return -1;
  hipLaunchKernelGGL(radix_tree_kernel, dim3(blocks), dim3(256), 0, 0, d_k1, (int)n, d_left, d_right, d_first, d_last, d_pn, d_pl);
  hipLaunchKernelGGL(fit_kernel, dim3(blocks), dim3(256), 0, 0, d_items, d_v1, (int)n, d_left, d_right, d_pn, d_pl, d_arr, d_box);
  hipLaunchKernelGGL(keep_kernel, dim3(blocks), dim3(256), 0, 0, d_first, d_last, (int)n, leaf_max, d_keep);
  size_t scan_bytes = 0;
  if (hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, d_keep, d_new, (int)n) != hipSuccess) return -1;
  unsigned char* d_scan = nullptr;
  if (!S.alloc(&d_scan, scan_bytes)) return -1;
  if (hipcub::DeviceScan::ExclusiveSum(d_scan, scan_bytes, d_keep, d_new, (int)n) != hipSuccess) return -1;
  hipLaunchKernelGGL(emit_kernel, dim3(blocks), dim3(256), 0, 0, d_items, d_v1, (int)n, leaf_max, base + (uint32_t)order.size(), d_left, d_right, d_first, d_last, d_keep, d_new,
                     d_box, center[0], center[1], center[2], d_out);
  if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) return -1;
  unsigned int last_new = 0, last_keep = 0;
  if (hipMemcpy(&last_new, d_new + (n - 2), sizeof(last_new), hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(&last_keep, d_keep + (n - 2), sizeof(last_keep), hipMemcpyDeviceToHost) != hipSuccess)
    return -1;
  const size_t n_nodes = (size_t)last_new + last_keep;
  if (n_nodes == 0) return -1;
  const size_t at = n2.size();
  n2.resize(at + n_nodes);
  if (hipMemcpy(n2.data() + at, d_out, n_nodes * sizeof(DBvhNode), hipMemcpyDeviceToHost) != hipSuccess) { n2.resize(at); return -1; }
  if (at) for (size_t i = at; i < n2.size(); i++) { if (n2[i].c0 >= 0) n2[i].c0 += (int32_t)at; if (n2[i].c1 >= 0) n2[i].c1 += (int32_t)at; }
  const size_t o0 = order.size();
  order.resize(o0 + n);
  if (hipMemcpy(order.data() + o0, d_v1, n * sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess) { n2.resize(at); order.resize(o0); return -1; }
  if (frame) { frame[0] = center[0]; frame[1] = center[1]; frame[2] = center[2]; frame[3] = rad * (1.0 + 1e-6) + 1e-30; }
  return (int32_t)at;  // node 0 of the radix tree is the root, it is kept (n > leaf_max) and the prefix sum gives it index 0
}
