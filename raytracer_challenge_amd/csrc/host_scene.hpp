// host_scene.hpp — host-side mirror of the reference's scene construction, and the flatten step.
//
// What the reference does once per scene on the CPU stays on the CPU here, bit-for-bit: Matrix::inverse
// (src/linalg/matrix.rs:162-207), Shape::shape (src/shape.rs:335-347), Element::composite +
// propagate_inverses (:47-101), the BoundingBox algebra that gives every group its world box
// (src/bounding_box.rs:19-78, src/shape.rs:948-996) and ObjParser::parse_obj (src/obj.rs:186-258).
// `flatten()` then walks the tree once (DFS = the reference's intersection insertion order) and emits the
// plain arrays of include/rtc.h.  Nothing here runs per pixel.  Build with -ffp-contract=off: these values
// decide hits.
#pragma once
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <map>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rtc.h"

namespace rth {

constexpr double kInf = std::numeric_limits<double>::infinity();

struct V4 { double x, y, z, w; };
inline V4 pt(double x, double y, double z) { return {x, y, z, 1.0}; }
inline V4 vec(double x, double y, double z) { return {x, y, z, 0.0}; }

struct M4 {
  double a[4][4];
  static M4 identity() {
    M4 r{};
    r.a[0][0] = r.a[1][1] = r.a[2][2] = r.a[3][3] = 1.0;
    return r;
  }
  static M4 from(const double* p) {
    M4 r;
    std::memcpy(r.a, p, sizeof(r.a));
    return r;
  }
  void to(double* p) const { std::memcpy(p, a, sizeof(a)); }
  M4 transposed() const {  // src/linalg/matrix.rs:126-136
    M4 r;
    for (int i = 0; i < 4; i++)
      for (int j = 0; j < 4; j++) r.a[j][i] = a[i][j];
    return r;
  }
  M4 mul(const M4& o) const {  // :239-259 — value starts at 0.0 and accumulates i = 0..3
    M4 r;
    for (int i = 0; i < 4; i++)
      for (int j = 0; j < 4; j++) {
        double v = 0.0;
        for (int k = 0; k < 4; k++) v += a[i][k] * o.a[k][j];
        r.a[i][j] = v;
      }
    return r;
  }
  V4 apply(const V4& v) const {  // :261-284
    return {a[0][0] * v.x + a[0][1] * v.y + a[0][2] * v.z + a[0][3] * v.w, a[1][0] * v.x + a[1][1] * v.y + a[1][2] * v.z + a[1][3] * v.w,
            a[2][0] * v.x + a[2][1] * v.y + a[2][2] * v.z + a[2][3] * v.w, a[3][0] * v.x + a[3][1] * v.y + a[3][2] * v.z + a[3][3] * v.w};
  }
  // :162-207 — pairs of 2x2 minors of the top two / bottom two rows; false when det == 0 (reference asserts).
  bool invert(M4* out) const {
    const double(*m)[4] = a;
    double top[6], bot[6];
    static const int P[6][2] = {{0, 1}, {0, 2}, {0, 3}, {1, 2}, {1, 3}, {2, 3}};
    for (int q = 0; q < 6; q++) {
      int i = P[q][0], j = P[q][1];
      top[q] = m[0][i] * m[1][j] - m[1][i] * m[0][j];  // s0..s5
      bot[q] = m[2][i] * m[3][j] - m[3][i] * m[2][j];  // c0..c5
    }
    const double *s = top, *c = bot;
    double det = s[0] * c[5] - s[1] * c[4] + s[2] * c[3] + s[3] * c[2] - s[4] * c[1] + s[5] * c[0];
    if (!(det != 0.0)) return false;
    M4 r;
    r.a[0][0] = (m[1][1] * c[5] - m[1][2] * c[4] + m[1][3] * c[3]) / det;
    r.a[0][1] = (-m[0][1] * c[5] + m[0][2] * c[4] - m[0][3] * c[3]) / det;
    r.a[0][2] = (m[3][1] * s[5] - m[3][2] * s[4] + m[3][3] * s[3]) / det;
    r.a[0][3] = (-m[2][1] * s[5] + m[2][2] * s[4] - m[2][3] * s[3]) / det;
    r.a[1][0] = (-m[1][0] * c[5] + m[1][2] * c[2] - m[1][3] * c[1]) / det;
    r.a[1][1] = (m[0][0] * c[5] - m[0][2] * c[2] + m[0][3] * c[1]) / det;
    r.a[1][2] = (-m[3][0] * s[5] + m[3][2] * s[2] - m[3][3] * s[1]) / det;
    r.a[1][3] = (m[2][0] * s[5] - m[2][2] * s[2] + m[2][3] * s[1]) / det;
    r.a[2][0] = (m[1][0] * c[4] - m[1][1] * c[2] + m[1][3] * c[0]) / det;
    r.a[2][1] = (-m[0][0] * c[4] + m[0][1] * c[2] - m[0][3] * c[0]) / det;
    r.a[2][2] = (m[3][0] * s[4] - m[3][1] * s[2] + m[3][3] * s[0]) / det;
    r.a[2][3] = (-m[2][0] * s[4] + m[2][1] * s[2] - m[2][3] * s[0]) / det;
    r.a[3][0] = (-m[1][0] * c[3] + m[1][1] * c[1] - m[1][2] * c[0]) / det;
    r.a[3][1] = (m[0][0] * c[3] - m[0][1] * c[1] + m[0][2] * c[0]) / det;
    r.a[3][2] = (-m[3][0] * s[3] + m[3][1] * s[1] - m[3][2] * s[0]) / det;
    r.a[3][3] = (m[2][0] * s[3] - m[2][1] * s[1] + m[2][2] * s[0]) / det;
    *out = r;
    return true;
  }
  bool same_bits(const M4& o) const { return std::memcmp(a, o.a, sizeof(a)) == 0; }
};

// Rust f64::min/max ignore a NaN operand.
inline double fmin_rs(double p, double q) { return p != p ? q : (q != q ? p : (p < q ? p : q)); }
inline double fmax_rs(double p, double q) { return p != p ? q : (q != q ? p : (p > q ? p : q)); }

struct Box {  // src/bounding_box.rs
  double lo[3], hi[3];
  static Box empty() { return {{kInf, kInf, kInf}, {-kInf, -kInf, -kInf}}; }  // :19-24
  void add(const V4& p) {                                                     // :30-43
    lo[0] = fmin_rs(lo[0], p.x); lo[1] = fmin_rs(lo[1], p.y); lo[2] = fmin_rs(lo[2], p.z);
    hi[0] = fmax_rs(hi[0], p.x); hi[1] = fmax_rs(hi[1], p.y); hi[2] = fmax_rs(hi[2], p.z);
  }
  void merge(const Box& o) {  // :45-47: insert(other.min).insert(other.max)
    add(pt(o.lo[0], o.lo[1], o.lo[2]));
    add(pt(o.hi[0], o.hi[1], o.hi[2]));
  }
  Box moved(const M4& m) const {  // :62-78 — the 8 corners in the reference's order
    Box r = empty();
    for (int k = 0; k < 8; k++) {
      V4 c = pt((k & 4) ? hi[0] : lo[0], (k & 2) ? hi[1] : lo[1], (k & 1) ? hi[2] : lo[2]);
      r.add(m.apply(c));
    }
    return r;
  }
};

struct Pat;
using PatRef = std::shared_ptr<const Pat>;
struct Pat {  // src/material.rs:60-65
  int tag = RTC_PAT_PLAIN, kind = 0, noise_kind = 0;
  uint32_t octaves = 1;
  double scale = 1.0, color[3] = {1, 1, 1};
  M4 transform_inv = M4::identity();
  PatRef left, right;
  int depth() const { return 1 + std::max(left ? left->depth() : 0, right ? right->depth() : 0); }
  // Frames the device's pattern walk keeps on a root-to-leaf path (rtc_device.hpp pattern_color): only a node that needs BOTH its
  // children's colours (Blend / RingGradient / Gradient) or post-processes its child's colour (colour jitter) keeps one; checkers,
  // rings, stripes and point jitters pass through.  The Box tree of the reference (src/material.rs:60-65) has no depth limit; the
  // device's is on this number (RTC_MAX_PATTERN_DEPTH), not on the tree's depth.
  bool keeps_frame() const {
    return (tag == RTC_PAT_JITTER && kind == RTC_JITTER_COLOR) || (tag == RTC_PAT_MIXTURE && (kind == RTC_MIX_BLEND || kind == RTC_MIX_RING_GRADIENT || kind == RTC_MIX_GRADIENT));
  }
  int frame_depth() const { return (keeps_frame() ? 1 : 0) + std::max(left ? left->frame_depth() : 0, right ? right->frame_depth() : 0); }
};

struct Mat {  // src/material.rs:19-43
  PatRef pattern;
  double ambient = 0.1, diffuse = 0.9, specular = 0.9, shininess = 200.0, reflective = 0.0, transparency = 0.0, refractive_index = 1.0;
};

struct Geo {  // src/shape.rs:466-498
  int kind = RTC_SPHERE;
  double lo = -kInf, hi = kInf;  // cylinder / cone limits
  bool closed = false;
  double p1e1e2[9] = {0}, normals[9] = {0};
  double p2[3] = {0}, p3[3] = {0};  // kept for the local bbox only

  Box local_box() const {  // :948-996
    switch (kind) {
      case RTC_SPHERE:
      case RTC_CUBE: return {{-1, -1, -1}, {1, 1, 1}};
      case RTC_PLANE: return {{-kInf, 0.0, -kInf}, {kInf, 0.0, kInf}};
      case RTC_CYLINDER:
        if (closed) return {{-1.0, lo, -1.0}, {1.0, hi, 1.0}};
        return {{-1.0, -kInf, -1.0}, {1.0, kInf, 1.0}};
      case RTC_CONE: {
        if (!closed) return {{-kInf, -kInf, -kInf}, {kInf, kInf, kInf}};
        double lim = fmax_rs(std::fabs(lo), std::fabs(hi));
        return {{-lim, lo, -lim}, {lim, hi, lim}};
      }
      default: {
        Box b = Box::empty();
        b.add(pt(p1e1e2[0], p1e1e2[1], p1e1e2[2]));
        b.add(pt(p2[0], p2[1], p2[2]));
        b.add(pt(p3[0], p3[1], p3[2]));
        return b;
      }
    }
  }
  static Geo triangle(const double* p, const double* n /*9 or null*/) {  // :369-412
    Geo g;
    g.kind = n ? RTC_SMOOTH_TRIANGLE : RTC_TRIANGLE;
    for (int c = 0; c < 3; c++) {
      g.p1e1e2[c] = p[c];
      g.p1e1e2[3 + c] = p[3 + c] - p[c];  // e1 = p2 - p1
      g.p1e1e2[6 + c] = p[6 + c] - p[c];  // e2 = p3 - p1
      g.p2[c] = p[3 + c];
      g.p3[c] = p[6 + c];
    }
    if (n) {
      std::memcpy(g.normals, n, 9 * sizeof(double));
    } else {  // n = normalize(e2 x e1)
      const double *e1 = g.p1e1e2 + 3, *e2 = g.p1e1e2 + 6;
      double cx = e2[1] * e1[2] - e2[2] * e1[1], cy = e2[2] * e1[0] - e2[0] * e1[2], cz = e2[0] * e1[1] - e2[1] * e1[0];
      double mag = std::sqrt(cx * cx + cy * cy + cz * cz);
      g.normals[0] = cx / mag; g.normals[1] = cy / mag; g.normals[2] = cz / mag;
    }
    return g;
  }
};

// One node of the Element tree.  Primitives keep the three matrices the reference keeps on Shape.
struct Elem {
  int group_kind = RTC_NODE_PRIM;  // RTC_NODE_PRIM or a group kind
  // primitive
  M4 inv, inv_tsp, mat_inv;
  Mat material;
  Geo geo;
  bool casts_shadow = true;
  // both: Shape.bbox / Group.bbox
  Box box = Box::empty();
  std::vector<std::unique_ptr<Elem>> kids;

  static std::unique_ptr<Elem> shape(const M4& transform, const Mat& m, bool casts, const Geo& g, std::string* err) {  // :335-347
    auto e = std::make_unique<Elem>();
    if (!transform.invert(&e->inv)) { if (err) *err = "singular shape transform (reference asserts det != 0, src/linalg/matrix.rs:181)"; return nullptr; }
    e->inv_tsp = e->inv.transposed();
    e->mat_inv = e->inv;
    e->box = g.local_box().moved(transform);
    e->material = m;
    e->geo = g;
    e->casts_shadow = casts;
    return e;
  }
  void push_down(const M4& transform, const M4& ginv, const M4& ginv_tsp, const Mat* m) {  // :47-72
    if (group_kind != RTC_NODE_PRIM) {
      for (auto& k : kids) k->push_down(transform, ginv, ginv_tsp, m);
      box = box.moved(transform);
    } else {
      inv = inv.mul(ginv);
      inv_tsp = ginv_tsp.mul(inv_tsp);
      if (m) { material = *m; mat_inv = ginv; }
      else mat_inv = mat_inv.mul(ginv);
    }
  }
  static std::unique_ptr<Elem> group(const M4& transform, const Mat* m, int kind, std::vector<std::unique_ptr<Elem>> kids, std::string* err) {  // :74-101
    if (kind != RTC_NODE_AGGREGATION && kids.size() != 2) { if (err) *err = "CSG group kinds take exactly two children (src/shape.rs:82)"; return nullptr; }
    M4 ginv;
    if (!transform.invert(&ginv)) { if (err) *err = "singular group transform"; return nullptr; }
    auto e = std::make_unique<Elem>();
    e->group_kind = kind;
    Box b = Box::empty();
    for (auto& k : kids) b.merge(k->box);
    e->box = b;
    e->kids = std::move(kids);
    e->push_down(transform, ginv, ginv.transposed(), m);
    return e;
  }
  uint64_t count_prims() const {
    if (group_kind == RTC_NODE_PRIM) return 1;
    uint64_t n = 0;
    for (auto& k : kids) n += k->count_prims();
    return n;
  }
};

struct Light { double intensity[3], origin[3]; };
struct WorldH {  // src/world.rs:12-15
  std::vector<Light> lights;
  std::vector<std::unique_ptr<Elem>> elements;
};

// ---------------------------------------------------------------------------------------------- OBJ
// ObjParser (src/obj.rs): line grammar :54-149, assembly :186-258.  Single pass over a memory-mapped/slurped
// buffer; per-line work is a handful of strtod calls, so a 10^6-triangle file parses in about a second.
struct ObjOut {
  std::unique_ptr<Elem> root;
  uint64_t ignored = 0, triangles = 0;
  std::string error;
};

namespace objp {
inline bool blanks(const char*& p) {
  const char* q = p;
  while (*q == ' ' || *q == '\t') q++;
  bool any = q != p;
  p = q;
  return any;
}
inline bool number(const char*& p, double* out) {  // nom `double`: sign, digits[.digits] | .digits, optional exponent
  const char* q = p;
  if (*q == '+' || *q == '-') q++;
  const char* d0 = q;
  while (*q >= '0' && *q <= '9') q++;
  bool ip = q > d0, fp = false;
  if (*q == '.') {
    const char* f = q + 1;
    while (*f >= '0' && *f <= '9') f++;
    fp = f > q + 1;
    if (ip || fp) q = f;
  }
  if (!ip && !fp) return false;
  if (*q == 'e' || *q == 'E') {
    const char* e = q + 1;
    if (*e == '+' || *e == '-') e++;
    const char* e0 = e;
    while (*e >= '0' && *e <= '9') e++;
    if (e > e0) q = e;
  }
  char buf[64];
  size_t len = (size_t)(q - p);
  if (len < sizeof(buf)) {
    std::memcpy(buf, p, len);
    buf[len] = 0;
    *out = std::strtod(buf, nullptr);
  } else {
    *out = std::strtod(std::string(p, q).c_str(), nullptr);
  }
  p = q;
  return true;
}
inline bool index(const char*& p, size_t* out) {
  const char* q = p;
  size_t v = 0;
  while (*q >= '0' && *q <= '9') v = v * 10 + (size_t)(*q++ - '0');
  if (q == p) return false;
  *out = v;
  p = q;
  return true;
}
struct Corner { size_t v, n; bool has_n; };
inline bool corner(const char*& p, Corner* c) {  // `v/anything/n` else plain `v` (:92-112)
  const char* q = p;
  if (!index(q, &c->v)) return false;
  c->has_n = false;
  const char* plain_end = q;
  if (*q == '/') {
    const char* r = q + 1;
    while (*r && *r != '/') r++;
    if (*r == '/') {
      r++;
      if (index(r, &c->n)) { c->has_n = true; p = r; return true; }
    }
  }
  p = plain_end;
  return true;
}
}  // namespace objp

inline ObjOut parse_obj_text(const std::string& text, const M4& transform, const Mat& material) {
  using namespace objp;
  ObjOut out;
  std::vector<double> vs, ns;  // xyz triples
  std::vector<std::pair<std::string, std::vector<std::unique_ptr<Elem>>>> groups;  // first-seen order (reference: HashMap)
  groups.emplace_back(std::string("Default"), std::vector<std::unique_ptr<Elem>>());
  size_t cur = 0;
  const M4 I = M4::identity();
  const Mat default_mat;
  size_t pos = 0, lineno = 1;
  std::string line;
  std::vector<Corner> cs;
  while (pos < text.size()) {
    size_t eol = text.find('\n', pos);
    if (eol == std::string::npos) eol = text.size();
    size_t end = eol;
    if (end > pos && text[end - 1] == '\r') end--;
    line.assign(text, pos, end - pos);
    pos = eol + 1;
    const char* p = line.c_str();
    bool ok = false;
    double x, y, z;
    if (p[0] == 'v') {
      const char* q = p + 1;
      if (blanks(q) && number(q, &x) && blanks(q) && number(q, &y) && blanks(q) && number(q, &z)) { vs.insert(vs.end(), {x, y, z}); ok = true; }
      if (!ok && p[1] == 'n') {
        q = p + 2;
        if (blanks(q) && number(q, &x) && blanks(q) && number(q, &y) && blanks(q) && number(q, &z)) { ns.insert(ns.end(), {x, y, z}); ok = true; }
      }
    } else if (p[0] == 'f') {
      const char* q = p + 1;
      if (blanks(q)) {
        cs.clear();
        Corner c;
        if (corner(q, &c)) {
          cs.push_back(c);
          for (;;) {
            const char* r = q;
            if (!blanks(r) || !corner(r, &c)) break;
            cs.push_back(c);
            q = r;
          }
        }
        if (cs.empty()) { out.error = "line " + std::to_string(lineno) + ": face without indices (reference underflows in triangulate, src/obj.rs:118)"; return out; }
        for (size_t i = 1; i + 1 < cs.size(); i++) {  // fan (:115-123)
          const Corner* t[3] = {&cs[0], &cs[i], &cs[i + 1]};
          double P[9], N[9];
          bool smooth = true;
          for (int k = 0; k < 3; k++) {
            if (t[k]->v < 1 || t[k]->v * 3 > vs.size()) { out.error = "line " + std::to_string(lineno) + ": vertex index out of range (src/obj.rs:209-214)"; return out; }
            std::memcpy(P + 3 * k, &vs[(t[k]->v - 1) * 3], 3 * sizeof(double));
            smooth = smooth && t[k]->has_n;
          }
          if (smooth)
            for (int k = 0; k < 3; k++) {
              if (t[k]->n < 1 || t[k]->n * 3 > ns.size()) { out.error = "line " + std::to_string(lineno) + ": normal index out of range (src/obj.rs:212-214)"; return out; }
              std::memcpy(N + 3 * k, &ns[(t[k]->n - 1) * 3], 3 * sizeof(double));
            }
          groups[cur].second.push_back(Elem::shape(I, default_mat, true, Geo::triangle(P, smooth ? N : nullptr), nullptr));
          out.triangles++;
        }
        ok = true;
      }
    } else if (p[0] == 'g') {
      const char* q = p + 1;
      if (blanks(q)) {
        const char* s = q;
        while ((*s >= '0' && *s <= '9') || (*s >= 'a' && *s <= 'z') || (*s >= 'A' && *s <= 'Z')) s++;
        if (s > q) {
          std::string name(q, s);
          size_t k = 0;
          while (k < groups.size() && groups[k].first != name) k++;
          if (k == groups.size()) groups.emplace_back(name, std::vector<std::unique_ptr<Elem>>());
          cur = k;
          ok = true;
        }
      }
    }
    if (!ok) out.ignored++;
    lineno++;
  }
  std::vector<std::unique_ptr<Elem>> tops;
  for (auto& g : groups)
    if (!g.second.empty()) {
      auto e = Elem::group(transform, &material, RTC_NODE_AGGREGATION, std::move(g.second), &out.error);
      if (!e) return out;
      tops.push_back(std::move(e));
    }
  if (tops.size() == 1) out.root = std::move(tops[0]);
  else out.root = Elem::group(I, nullptr, RTC_NODE_AGGREGATION, std::move(tops), &out.error);
  return out;
}

// ---------------------------------------------------------------------------------------------- flatten
struct Flat {
  std::vector<rtc_node> nodes;
  std::vector<rtc_prim> prims;
  std::vector<rtc_xform> xforms;
  std::vector<double> limits, tri_geo, tri_nrm;
  std::vector<rtc_material> materials;
  std::vector<rtc_pattern_node> pats;
  std::vector<rtc_light> lights;
  std::string error;

  rtc_scene_desc desc() const {
    rtc_scene_desc d{};
    d.n_nodes = (uint32_t)nodes.size(); d.nodes = nodes.data();
    d.n_prims = (uint32_t)prims.size(); d.prims = prims.data();
    d.n_xforms = (uint32_t)xforms.size(); d.xforms = xforms.data();
    d.n_limits = (uint32_t)(limits.size() / 2); d.limits = limits.data();
    d.n_tris = (uint32_t)(tri_geo.size() / 9); d.tri_p1e1e2 = tri_geo.data(); d.tri_normals = tri_nrm.data();
    d.n_materials = (uint32_t)materials.size(); d.materials = materials.data();
    d.n_pattern_nodes = (uint32_t)pats.size(); d.pattern_nodes = pats.data();
    d.n_lights = (uint32_t)lights.size(); d.lights = lights.data();
    return d;
  }
};

// [0, n) cut into one range per thread (at most 8, RTC_BUILD_THREADS overrides), for the flatten's fills
template <class F>
inline void par_ranges(size_t n, F fn) {
  unsigned t = std::thread::hardware_concurrency();
  if (const char* e = std::getenv("RTC_BUILD_THREADS")) t = (unsigned)std::max(1, std::atoi(e));
  t = std::min<unsigned>(std::max(1u, t), 8u);
  if (t < 2 || n < 65536) { fn((size_t)0, n); return; }
  std::vector<std::thread> th;
  const size_t per = (n + t - 1) / t;
  for (unsigned k = 0; k < t; k++) {
    const size_t b = std::min(n, per * k), e = std::min(n, per * (k + 1));
    if (e > b) th.emplace_back([=] { fn(b, e); });
  }
  for (auto& x : th) x.join();
}

class Flattener {
 public:
  explicit Flattener(Flat& f) : f_(f) {}
  bool run(const WorldH& w) {
    for (auto& l : w.lights) {
      rtc_light r;
      std::memcpy(r.intensity, l.intensity, sizeof(r.intensity));
      std::memcpy(r.origin, l.origin, sizeof(r.origin));
      f_.lights.push_back(r);
    }
    // one pass to size the arrays (10^6 triangles: the doubling reallocations of 230 MB of vectors were a quarter of the flatten time)
    uint64_t n_prims = 0;
    for (auto& e : w.elements) n_prims += e->count_prims();
    f_.prims.reserve(n_prims);
    f_.nodes.reserve(n_prims + n_prims / 8 + 16);
    if (n_prims > 4096) { f_.tri_geo.reserve(9 * n_prims); f_.tri_nrm.reserve(9 * n_prims); }
    for (auto& e : w.elements)
      if (!walk(*e)) return false;
    return true;
  }

 private:
  Flat& f_;
  std::map<const Pat*, int32_t> pat_ids_;
  // materials are de-duplicated on (pattern node, 7 scalars): a whole OBJ group shares one
  std::map<std::vector<uint64_t>, int32_t> mat_ids_;

  int32_t pattern(const PatRef& p) {
    auto it = pat_ids_.find(p.get());
    if (it != pat_ids_.end()) return it->second;
    int32_t l = p->left ? pattern(p->left) : -1, r = p->right ? pattern(p->right) : -1;
    rtc_pattern_node n{};
    n.tag = p->tag; n.kind = p->kind; n.noise_kind = p->noise_kind; n.octaves = p->octaves;
    n.left = l; n.right = r; n.scale = p->scale;
    std::memcpy(n.color, p->color, sizeof(n.color));
    p->transform_inv.to(n.transform_inv);
    f_.pats.push_back(n);
    return pat_ids_[p.get()] = (int32_t)f_.pats.size() - 1;
  }
  // (the 10^6 triangles of an OBJ group name the same material one after the other: the previous answer is checked first, without
  // the map's key allocation — 0.10 of config 5's 0.14 s of flattening)
  const Pat* last_mat_pat_ = nullptr;
  double last_mat_s_[7] = {0, 0, 0, 0, 0, 0, 0};
  int32_t last_mat_id_ = -1;
  static const PatRef& white_pat() {
    static const PatRef white = [] { auto p = std::make_shared<Pat>(); return PatRef(p); }();
    return white;
  }
  int32_t material(const Mat& m) {
    const PatRef& white = white_pat();
    const double s[7] = {m.ambient, m.diffuse, m.specular, m.shininess, m.reflective, m.transparency, m.refractive_index};
    const Pat* pp = (m.pattern ? m.pattern : white).get();
    if (last_mat_id_ >= 0 && pp == last_mat_pat_ && std::memcmp(s, last_mat_s_, sizeof(s)) == 0) return last_mat_id_;
    const int32_t id = material_slow(m, s);
    last_mat_pat_ = pp; std::memcpy(last_mat_s_, s, sizeof(s)); last_mat_id_ = id;
    return id;
  }
  int32_t material_slow(const Mat& m, const double s[7]) {
    const PatRef& white = white_pat();
    int32_t pid = pattern(m.pattern ? m.pattern : white);
    std::vector<uint64_t> key(8);
    key[0] = (uint64_t)pid;
    std::memcpy(&key[1], s, 7 * sizeof(double));
    auto it = mat_ids_.find(key);
    if (it != mat_ids_.end()) return it->second;
    rtc_material r{s[0], s[1], s[2], s[3], s[4], s[5], s[6], pid, 0};
    f_.materials.push_back(r);
    return mat_ids_[key] = (int32_t)f_.materials.size() - 1;
  }
  int32_t xform(const Elem& e) {
    // consecutive primitives of one OBJ group carry bit-identical matrices: share the record
    if (!f_.xforms.empty()) {
      const rtc_xform& last = f_.xforms.back();
      if (std::memcmp(last.transform_inv, e.inv.a, sizeof(e.inv.a)) == 0 && std::memcmp(last.material_inv, e.mat_inv.a, sizeof(e.mat_inv.a)) == 0)
        return (int32_t)f_.xforms.size() - 1;
    }
    // the device rebuilds transform_inv_tsp as transpose(transform_inv); the two are bitwise equal by
    // construction (same products, same summation order) — refuse a tree where that does not hold.
    if (!e.inv_tsp.same_bits(e.inv.transposed())) { f_.error = "transform_inv_tsp != transpose(transform_inv)"; return -1; }
    rtc_xform x;
    e.inv.to(x.transform_inv);
    e.mat_inv.to(x.material_inv);
    f_.xforms.push_back(x);
    return (int32_t)f_.xforms.size() - 1;
  }
  bool walk(const Elem& e) {
    rtc_node n{};
    if (e.group_kind == RTC_NODE_PRIM) {
      rtc_prim p{};
      p.geometry = e.geo.kind;
      p.flags = (e.casts_shadow ? RTC_FLAG_CASTS_SHADOW : 0u) | (e.geo.closed ? RTC_FLAG_CLOSED : 0u);
      p.material = material(e.material);
      p.xform = xform(e);
      if (p.xform < 0) return false;
      p.data = -1;
      if (e.geo.kind == RTC_CYLINDER || e.geo.kind == RTC_CONE) {
        p.data = (int32_t)(f_.limits.size() / 2);
        f_.limits.push_back(e.geo.lo);
        f_.limits.push_back(e.geo.hi);
      } else if (e.geo.kind >= RTC_TRIANGLE) {
        p.data = (int32_t)(f_.tri_geo.size() / 9);
        f_.tri_geo.insert(f_.tri_geo.end(), e.geo.p1e1e2, e.geo.p1e1e2 + 9);
        f_.tri_nrm.insert(f_.tri_nrm.end(), e.geo.normals, e.geo.normals + 9);
      }
      f_.prims.push_back(p);
      n.kind = RTC_NODE_PRIM;
      n.ref = (int32_t)f_.prims.size() - 1;
      n.skip = (int32_t)f_.nodes.size() + 1;
      f_.nodes.push_back(n);
      return true;
    }
    n.kind = e.group_kind;
    n.ref = -1;
    std::memcpy(n.bbox_min, e.box.lo, sizeof(n.bbox_min));
    std::memcpy(n.bbox_max, e.box.hi, sizeof(n.bbox_max));
    size_t self = f_.nodes.size();
    f_.nodes.push_back(n);
    if (e.kids.size() >= 32768 && !std::getenv("RTC_FLATTEN_SERIAL") && uniform_triangles(e)) {  // (the variable: tests compare the two)
      if (!fill_uniform_triangles(e)) return false;
    } else {
      for (auto& k : e.kids)
        if (!walk(*k)) return false;
    }
    f_.nodes[self].skip = (int32_t)f_.nodes.size();
    return true;
  }
  // A large group whose children are all triangles with one material and one set of matrices (what ObjParser makes of an OBJ group,
  // src/obj.rs:186-258): the records of its children are the ones walk() would emit, written by several threads.
  static bool same_prim_frame(const Elem& a, const Elem& b) {
    const double sa[7] = {a.material.ambient, a.material.diffuse, a.material.specular, a.material.shininess, a.material.reflective, a.material.transparency, a.material.refractive_index};
    const double sb[7] = {b.material.ambient, b.material.diffuse, b.material.specular, b.material.shininess, b.material.reflective, b.material.transparency, b.material.refractive_index};
    return a.material.pattern.get() == b.material.pattern.get() && std::memcmp(sa, sb, sizeof(sa)) == 0 && std::memcmp(a.inv.a, b.inv.a, sizeof(a.inv.a)) == 0 &&
           std::memcmp(a.mat_inv.a, b.mat_inv.a, sizeof(a.mat_inv.a)) == 0 && std::memcmp(a.inv_tsp.a, b.inv_tsp.a, sizeof(a.inv_tsp.a)) == 0;
  }
  static bool uniform_triangles(const Elem& g) {
    const Elem& k0 = *g.kids[0];
    std::atomic<bool> ok{true};
    par_ranges(g.kids.size(), [&](size_t b, size_t e) {
      for (size_t i = b; i < e; i++) {
        const Elem& k = *g.kids[i];
        if (k.group_kind != RTC_NODE_PRIM || k.geo.kind < RTC_TRIANGLE || !same_prim_frame(k, k0)) { ok = false; return; }
      }
    });
    return ok.load();
  }
  bool fill_uniform_triangles(const Elem& g) {
    const Elem& k0 = *g.kids[0];
    const int32_t mid = material(k0.material), xid = xform(k0);
    if (xid < 0) return false;
    const size_t n = g.kids.size(), prim0 = f_.prims.size(), node0 = f_.nodes.size(), tri0 = f_.tri_geo.size() / 9;
    f_.prims.resize(prim0 + n);
    f_.nodes.resize(node0 + n);
    f_.tri_geo.resize(9 * (tri0 + n));
    f_.tri_nrm.resize(9 * (tri0 + n));
    par_ranges(n, [&](size_t b, size_t e) {
      for (size_t i = b; i < e; i++) {
        const Elem& k = *g.kids[i];
        rtc_prim p{};
        p.geometry = k.geo.kind;
        p.flags = (k.casts_shadow ? RTC_FLAG_CASTS_SHADOW : 0u) | (k.geo.closed ? RTC_FLAG_CLOSED : 0u);
        p.material = mid;
        p.xform = xid;
        p.data = (int32_t)(tri0 + i);
        f_.prims[prim0 + i] = p;
        rtc_node nd{};
        nd.kind = RTC_NODE_PRIM;
        nd.ref = (int32_t)(prim0 + i);
        nd.skip = (int32_t)(node0 + i + 1);
        f_.nodes[node0 + i] = nd;
        std::memcpy(&f_.tri_geo[9 * (tri0 + i)], k.geo.p1e1e2, 9 * sizeof(double));
        std::memcpy(&f_.tri_nrm[9 * (tri0 + i)], k.geo.normals, 9 * sizeof(double));
      }
    });
    return true;
  }
};

// Camera::new (src/camera.rs:16-37).
inline bool make_camera(uint64_t hsize, uint64_t vsize, double fov, const M4& transform, rtc_camera* out) {
  double half_view = std::tan(fov / 2.0);
  double aspect = (double)hsize / (double)vsize;
  if (aspect >= 1.0) { out->half_width = half_view; out->half_height = half_view / aspect; }
  else { out->half_width = half_view * aspect; out->half_height = half_view; }
  out->pixel_size = (out->half_width * 2.0) / (double)hsize;
  out->hsize = hsize;
  out->vsize = vsize;
  M4 inv;
  if (!transform.invert(&inv)) return false;
  inv.to(out->transform_inv);
  return true;
}

}  // namespace rth
