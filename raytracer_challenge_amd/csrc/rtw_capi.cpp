// rtw_capi.cpp — include/rtw.h for the product: C handles over the host mirror (host_scene.hpp).  A world is
// flattened once (first render after the last edit), uploaded through rtc_scene_create, and every render goes
// to the HIP kernels through the rtc.h entry points.  Nothing here can compute a pixel on the CPU.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <memory>
#include <sstream>
#include <string>

#include "../../include/rtc.h"
#include "../../include/rtw.h"
#include "host_scene.hpp"

using namespace rth;

struct rtw_pattern { PatRef p; };
struct rtw_element { std::unique_ptr<Elem> e; };
struct rtw_world {
  WorldH w;
  rtc_scene* scene = nullptr;  // cached flatten+upload; dropped on edit
  int device = 0;
  std::unique_ptr<Flat> flat;  // rtw_world_flatten_desc: the arrays behind the descriptor it handed out
  ~rtw_world() { if (scene) rtc_scene_destroy(scene); }
};

static thread_local std::string g_err;
static int fail(const std::string& m) { g_err = m; return 1; }

static Mat to_mat(const rtw_material* m) {
  Mat r;
  if (!m) return r;
  r.ambient = m->ambient; r.diffuse = m->diffuse; r.specular = m->specular; r.shininess = m->shininess;
  r.reflective = m->reflective; r.transparency = m->transparency; r.refractive_index = m->refractive_index;
  if (m->pattern) r.pattern = m->pattern->p;
  return r;
}

static int ensure_scene(rtw_world* w) {
  if (w->scene) return 0;
  const bool timing = std::getenv("RTC_TIMING") != nullptr;
  const auto t0 = std::chrono::steady_clock::now();
  auto since = [](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count(); };
  int rc;
  {
    Flat f;
    Flattener fl(f);
    if (!fl.run(w->w)) return fail("flatten: " + f.error);
    rtc_scene_desc d = f.desc();
    if (timing) std::fprintf(stderr, "[rtc-timing] %-28s %.3f s\n", "flatten (host mirror -> desc)", since(t0));
    const auto t1 = std::chrono::steady_clock::now();
    rc = rtc_scene_create(&d, w->device, &w->scene);
    if (timing) std::fprintf(stderr, "[rtc-timing] %-28s %.3f s\n", "rtc_scene_create (all of it)", since(t1));
  }
  if (timing) std::fprintf(stderr, "[rtc-timing] %-28s %.3f s\n", "flatten + create + frees", since(t0));
  if (rc != RTC_OK) return fail(std::string("rtc_scene_create: ") + rtc_last_error());
  return 0;
}

extern "C" {

const char* rtw_last_error(void) { return g_err.c_str(); }
#ifndef RTW_BACKEND_NAME
#define RTW_BACKEND_NAME "hip"
#endif
const char* rtw_backend(void) { return RTW_BACKEND_NAME; }

rtw_pattern* rtw_pattern_debug(void) {
  auto p = std::make_shared<Pat>();
  p->tag = RTC_PAT_DEBUG;
  return new rtw_pattern{p};
}
rtw_pattern* rtw_pattern_plain(double r, double g, double b) {
  auto p = std::make_shared<Pat>();
  p->tag = RTC_PAT_PLAIN;
  p->color[0] = r; p->color[1] = g; p->color[2] = b;
  return new rtw_pattern{p};
}
rtw_pattern* rtw_pattern_jitter(int jk, int nk, double scale, uint64_t octaves, const rtw_pattern* child) {
  if (!child) { fail("jitter: child is NULL"); return nullptr; }
  auto p = std::make_shared<Pat>();
  p->tag = RTC_PAT_JITTER; p->kind = jk; p->noise_kind = nk; p->scale = scale; p->octaves = (uint32_t)octaves; p->left = child->p;
  if (p->frame_depth() > RTC_MAX_PATTERN_DEPTH) { fail("pattern keeps more than RTC_MAX_PATTERN_DEPTH colour frames on one path (blends / gradients / colour jitters nested deeper than 8)"); return nullptr; }
  return new rtw_pattern{p};
}
rtw_pattern* rtw_pattern_mixture(int mk, const double t[16], const rtw_pattern* l, const rtw_pattern* r) {
  if (!l || !r) { fail("mixture: child is NULL"); return nullptr; }
  auto p = std::make_shared<Pat>();
  p->tag = RTC_PAT_MIXTURE; p->kind = mk; p->left = l->p; p->right = r->p;
  if (!M4::from(t).invert(&p->transform_inv)) { fail("mixture: singular transform (src/linalg/matrix.rs:181)"); return nullptr; }
  if (p->frame_depth() > RTC_MAX_PATTERN_DEPTH) { fail("pattern keeps more than RTC_MAX_PATTERN_DEPTH colour frames on one path (blends / gradients / colour jitters nested deeper than 8)"); return nullptr; }
  return new rtw_pattern{p};
}
void rtw_pattern_release(rtw_pattern* p) { delete p; }

rtw_element* rtw_shape(int geometry, const double t[16], const rtw_material* material, int casts_shadow, const double* p, size_t np) {
  Geo g;
  switch (geometry) {
    case RTW_SPHERE: g.kind = RTC_SPHERE; break;
    case RTW_PLANE: g.kind = RTC_PLANE; break;
    case RTW_CUBE: g.kind = RTC_CUBE; break;
    case RTW_CYLINDER:
    case RTW_CONE:
      if (np != 3) { fail("cylinder/cone: params = {min,max,closed}"); return nullptr; }
      g.kind = geometry == RTW_CYLINDER ? RTC_CYLINDER : RTC_CONE;
      g.lo = p[0]; g.hi = p[1]; g.closed = p[2] != 0.0;
      break;
    case RTW_TRIANGLE:
      if (np != 9) { fail("triangle: 9 params"); return nullptr; }
      g = Geo::triangle(p, nullptr);
      break;
    case RTW_SMOOTH_TRIANGLE:
      if (np != 18) { fail("smooth triangle: 18 params"); return nullptr; }
      g = Geo::triangle(p, p + 9);
      break;
    default: fail("unknown geometry"); return nullptr;
  }
  std::string err;
  auto e = Elem::shape(M4::from(t), to_mat(material), casts_shadow != 0, g, &err);
  if (!e) { fail(err); return nullptr; }
  return new rtw_element{std::move(e)};
}

rtw_element* rtw_composite(const double t[16], const rtw_material* material, int kind, rtw_element** children, size_t n) {
  std::vector<std::unique_ptr<Elem>> kids;
  for (size_t i = 0; i < n; i++) {
    if (!children[i] || !children[i]->e) { fail("composite: NULL/consumed child"); return nullptr; }
  }
  for (size_t i = 0; i < n; i++) {
    kids.push_back(std::move(children[i]->e));
    delete children[i];
  }
  Mat m = to_mat(material);
  std::string err;
  auto e = Elem::group(M4::from(t), material ? &m : nullptr, kind, std::move(kids), &err);
  if (!e) { fail(err); return nullptr; }
  return new rtw_element{std::move(e)};
}

rtw_element* rtw_parse_obj(const char* path, const double t[16], const rtw_material* material, uint64_t* n_ignored, uint64_t* n_triangles) {
  std::ifstream f(path, std::ios::binary);
  if (!f) { fail(std::string("cannot open ") + path); return nullptr; }
  std::stringstream ss;
  ss << f.rdbuf();
  M4 tr = M4::from(t), tmp;
  if (!tr.invert(&tmp)) { fail("parse_obj: singular transform"); return nullptr; }
  ObjOut o = parse_obj_text(ss.str(), tr, to_mat(material));
  if (!o.error.empty() || !o.root) { fail(o.error.empty() ? "parse_obj: no geometry" : o.error); return nullptr; }
  if (n_ignored) *n_ignored = o.ignored;
  if (n_triangles) *n_triangles = o.triangles;
  return new rtw_element{std::move(o.root)};
}
void rtw_element_release(rtw_element* e) { delete e; }

rtw_world* rtw_world_create(void) { return new rtw_world(); }
int rtw_world_add_light(rtw_world* w, const double i[3], const double o[3]) {
  Light l;
  std::memcpy(l.intensity, i, sizeof(l.intensity));
  std::memcpy(l.origin, o, sizeof(l.origin));
  w->w.lights.push_back(l);
  if (w->scene) { rtc_scene_destroy(w->scene); w->scene = nullptr; }
  return 0;
}
int rtw_world_add_element(rtw_world* w, rtw_element* e) {
  if (!e || !e->e) return fail("add_element: NULL/consumed element");
  w->w.elements.push_back(std::move(e->e));
  delete e;
  if (w->scene) { rtc_scene_destroy(w->scene); w->scene = nullptr; }
  return 0;
}
uint64_t rtw_world_primitive_count(const rtw_world* w) {
  uint64_t n = 0;
  for (auto& e : w->w.elements) n += e->count_prims();
  return n;
}
void rtw_world_release(rtw_world* w) { delete w; }

int rtw_render(rtw_world* w, const rtw_camera* cam, int fuel, const uint64_t* idx, uint64_t n, double* rgb, rtw_hit* hits) {
  if (ensure_scene(w)) return 1;
  rtc_camera c;
  if (!make_camera(cam->hsize, cam->vsize, cam->field_of_view, M4::from(cam->transform), &c)) return fail("camera: singular transform");
  static_assert(sizeof(rtw_hit) == sizeof(rtc_hit), "hit layout");
  int rc = rtc_render(w->scene, &c, fuel, idx, 0, n, rgb, (rtc_hit*)hits, nullptr);
  if (rc != RTC_OK) return fail(std::string("rtc_render: ") + rtc_last_error());
  return 0;
}

int rtw_color_at(rtw_world* w, const double* rays, uint64_t n, int fuel, double* rgb, rtw_hit* hits) {
  if (ensure_scene(w)) return 1;
  int rc = rtc_trace_rays(w->scene, rays, n, fuel, rgb, (rtc_hit*)hits, nullptr);
  if (rc != RTC_OK) return fail(std::string("rtc_trace_rays: ") + rtc_last_error());
  return 0;
}

// ---- product-only helpers for the Python harness / bench (not in rtw.h) ------------------------------------
// The flattened + uploaded scene behind a world (created on first use).
rtc_scene* rtw_world_scene(rtw_world* w, int device) {
  if (w->scene && w->device != device) { rtc_scene_destroy(w->scene); w->scene = nullptr; }
  w->device = device;
  if (ensure_scene(w)) return nullptr;
  return w->scene;
}
// Camera::new -> rtc_camera.
int rtw_make_camera(const rtw_camera* cam, rtc_camera* out) {
  if (!make_camera(cam->hsize, cam->vsize, cam->field_of_view, M4::from(cam->transform), out)) return fail("camera: singular transform");
  return 0;
}
// Flatten only (no device): sizes of the arrays a Rust shim would hand to rtc_scene_create.  Works without a GPU.
int rtw_world_flatten_counts(rtw_world* w, uint32_t counts[8]) {
  Flat f;
  Flattener fl(f);
  if (!fl.run(w->w)) return fail("flatten: " + f.error);
  rtc_scene_desc d = f.desc();
  counts[0] = d.n_nodes; counts[1] = d.n_prims; counts[2] = d.n_xforms; counts[3] = d.n_limits;
  counts[4] = d.n_tris; counts[5] = d.n_materials; counts[6] = d.n_pattern_nodes; counts[7] = d.n_lights;
  return 0;
}

// Flatten only (no device): the descriptor a Rust shim would hand to rtc_scene_create; its arrays live in the world handle until
// the next call / the world's release.  Works without a GPU (tests compare it with a foreign flattener's output).
int rtw_world_flatten_desc(rtw_world* w, rtc_scene_desc* out) {
  w->flat.reset(new Flat());
  Flattener fl(*w->flat);
  if (!fl.run(w->w)) return fail("flatten: " + w->flat->error);
  *out = w->flat->desc();
  return 0;
}

}  // extern "C"
