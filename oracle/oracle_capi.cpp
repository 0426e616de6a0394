// oracle/oracle_capi.cpp — TEST INFRASTRUCTURE.  Implements include/rtw.h on top of the literal CPU
// restatement in rt_oracle.hpp (-> liboracle.so), plus a few oracle-only entry points (orc_*) used by
// tests and by bench.py's cpu_baseline leg.  Never loaded by the product package.
#include "rt_oracle.hpp"

#include "../include/rtw.h"

using namespace orc;

struct rtw_pattern { PatternPtr p; };
struct rtw_element { ElementPtr e; };
struct rtw_world { World w; };

static thread_local std::string g_err;
static int fail(const std::string& m) { g_err = m; return 1; }

static Material to_material(const rtw_material* m) {
  Material r;
  if (!m) return r;
  r.ambient = m->ambient; r.diffuse = m->diffuse; r.specular = m->specular; r.shininess = m->shininess;
  r.reflective = m->reflective; r.transparency = m->transparency; r.refractive_index = m->refractive_index;
  if (m->pattern) r.pattern = m->pattern->p;
  return r;
}

extern "C" {

const char* rtw_last_error(void) { return g_err.c_str(); }
const char* rtw_backend(void) { return "oracle-cpu"; }

rtw_pattern* rtw_pattern_debug(void) { return new rtw_pattern{Pattern::debug()}; }
rtw_pattern* rtw_pattern_plain(double r, double g, double b) { return new rtw_pattern{Pattern::plain({r, g, b})}; }
rtw_pattern* rtw_pattern_jitter(int jk, int nk, double scale, uint64_t octaves, const rtw_pattern* child) {
  if (!child) { fail("jitter: child is NULL"); return nullptr; }
  Noise n;
  n.kind = nk == RTW_NOISE_FRACTAL ? Noise::Fractal : Noise::Simplex;
  n.scale = scale;
  n.octaves = (size_t)octaves;
  return new rtw_pattern{Pattern::jitter(jk == RTW_JITTER_COLOR ? JitterColor : JitterPoint, n, child->p)};
}
rtw_pattern* rtw_pattern_mixture(int mk, const double t[16], const rtw_pattern* l, const rtw_pattern* r) {
  if (!l || !r) { fail("mixture: child is NULL"); return nullptr; }
  Matrix m = Matrix::from16(t), inv;
  if (!m.inverse(&inv)) { fail("mixture: singular transform (src/linalg/matrix.rs:181)"); return nullptr; }
  return new rtw_pattern{Pattern::mixture((MixtureKind)mk, m, l->p, r->p)};
}
void rtw_pattern_release(rtw_pattern* p) { delete p; }

rtw_element* rtw_shape(int geometry, const double t[16], const rtw_material* material, int casts_shadow,
                       const double* p, size_t np) {
  Matrix m = Matrix::from16(t), inv;
  if (!m.inverse(&inv)) { fail("shape: singular transform (src/linalg/matrix.rs:181)"); return nullptr; }
  Geometry g;
  auto P = [&](int k) { return Vector::point(p[k], p[k + 1], p[k + 2]); };
  auto V = [&](int k) { return Vector::vector(p[k], p[k + 1], p[k + 2]); };
  switch (geometry) {
    case RTW_SPHERE: g.kind = Sphere; break;
    case RTW_PLANE: g.kind = Plane; break;
    case RTW_CUBE: g.kind = Cube; break;
    case RTW_CYLINDER:
    case RTW_CONE:
      if (np != 3) { fail("cylinder/cone: params = {min,max,closed}"); return nullptr; }
      g.kind = geometry == RTW_CYLINDER ? Cylinder : Cone;
      g.min = p[0]; g.max = p[1]; g.closed = p[2] != 0.0;
      break;
    case RTW_TRIANGLE:
      if (np != 9) { fail("triangle: 9 params"); return nullptr; }
      g = Shape::triangle_geometry(P(0), P(3), P(6));
      break;
    case RTW_SMOOTH_TRIANGLE:
      if (np != 18) { fail("smooth triangle: 18 params"); return nullptr; }
      g = Shape::smooth_triangle_geometry(P(0), P(3), P(6), V(9), V(12), V(15));
      break;
    default: fail("unknown geometry"); return nullptr;
  }
  return new rtw_element{Element::primitive(Shape::make(m, to_material(material), casts_shadow != 0, g))};
}

rtw_element* rtw_composite(const double t[16], const rtw_material* material, int kind, rtw_element** children, size_t n) {
  Matrix m = Matrix::from16(t), inv;
  if (!m.inverse(&inv)) { fail("composite: singular transform"); return nullptr; }
  if (kind != RTW_AGGREGATION && n != 2) { fail("composite: CSG kinds take exactly 2 children (src/shape.rs:82)"); return nullptr; }
  std::vector<ElementPtr> ch;
  for (size_t i = 0; i < n; i++) {
    ch.push_back(std::move(children[i]->e));
    delete children[i];
  }
  Material mat = to_material(material);
  return new rtw_element{Element::composite(m, material ? &mat : nullptr, (GroupKind)kind, std::move(ch))};
}

rtw_element* rtw_parse_obj(const char* path, const double t[16], const rtw_material* material, uint64_t* n_ignored,
                           uint64_t* n_triangles) {
  Matrix m = Matrix::from16(t), inv;
  if (!m.inverse(&inv)) { fail("parse_obj: singular transform"); return nullptr; }
  ObjResult r = parse_obj_file(path, m, to_material(material));
  if (!r.error.empty()) { fail(r.error); return nullptr; }
  if (n_ignored) *n_ignored = r.ignored.size();
  if (n_triangles) *n_triangles = r.triangles;
  return new rtw_element{std::move(r.element)};
}
void rtw_element_release(rtw_element* e) { delete e; }

rtw_world* rtw_world_create(void) { return new rtw_world(); }
int rtw_world_add_light(rtw_world* w, const double i[3], const double o[3]) {
  w->w.lights.push_back({{i[0], i[1], i[2]}, Vector::point(o[0], o[1], o[2])});
  return 0;
}
int rtw_world_add_element(rtw_world* w, rtw_element* e) {
  if (!e || !e->e) return fail("add_element: NULL/consumed element");
  w->w.elements.push_back(std::move(e->e));
  delete e;
  return 0;
}
uint64_t rtw_world_primitive_count(const rtw_world* w) {
  std::vector<const Shape*> o;
  w->w.number_shapes(o);
  return o.size();
}
void rtw_world_release(rtw_world* w) { delete w; }

// ---- oracle-only -----------------------------------------------------------------------------
struct orc_stats {
  double seconds;
  uint64_t threads;
  double unique_rays;    // sum_d calls[d] / L^d  (SURVEY.md §8d "unique rays")
  double traced_rays;    // as the reference traces them (with its per-light re-tracing)
  uint64_t nan_seen;     // 1 if a NaN t reached a sort (the reference would panic)
};

int orc_render_ex(rtw_world* w, const rtw_camera* cam, int fuel, const uint64_t* idx, uint64_t n, double* rgb, rtw_hit* hits,
                  uint32_t threads, orc_stats* st, uint64_t* digest);
int orc_render(rtw_world* w, const rtw_camera* cam, int fuel, const uint64_t* idx, uint64_t n, double* rgb, rtw_hit* hits,
               uint32_t threads, orc_stats* st) {
  return orc_render_ex(w, cam, fuel, idx, n, rgb, hits, threads, st, nullptr);
}
// The same pass with the hit-tree digests of the pixels (rt_oracle.hpp World::hit_hash; digest may be NULL).
int orc_render_ex(rtw_world* w, const rtw_camera* cam, int fuel, const uint64_t* idx, uint64_t n, double* rgb, rtw_hit* hits,
                  uint32_t threads, orc_stats* st, uint64_t* digest) {
  Matrix m = Matrix::from16(cam->transform), inv;
  if (!m.inverse(&inv)) return fail("camera: singular transform");
  Camera c = Camera::make((size_t)cam->hsize, (size_t)cam->vsize, cam->field_of_view, m);
  static_assert(sizeof(HitRecord) == sizeof(rtw_hit), "hit record layout");
  RenderResult rr = render_pixels(c, w->w, fuel, idx, (size_t)n, rgb, (HitRecord*)hits, threads, digest);
  if (st) {
    double L = (double)w->w.lights.size(), unique = 0, traced = 0, div = 1.0;
    for (int d = 0; d < Counters::MAXD; d++) {
      double calls = (double)rr.counters.color_at[d] + (double)rr.counters.shadow[d];
      traced += calls;
      unique += div > 0 ? calls / div : 0;
      div *= L;
    }
    st->seconds = rr.seconds;
    st->threads = threads ? threads : std::max(1u, std::thread::hardware_concurrency());
    st->unique_rays = unique;
    st->traced_rays = traced;
    st->nan_seen = rr.nan_seen ? 1 : 0;
  }
  if (rr.nan_seen) return fail("NaN intersection t reached Intersection::sort (reference panics, src/intersection.rs:124)");
  return 0;
}

int rtw_render(rtw_world* w, const rtw_camera* cam, int fuel, const uint64_t* idx, uint64_t n, double* rgb, rtw_hit* hits) {
  return orc_render(w, cam, fuel, idx, n, rgb, hits, 0, nullptr);
}

int rtw_color_at(rtw_world* w, const double* rays, uint64_t n, int fuel, double* rgb, rtw_hit* hits) {
  std::vector<const Shape*> order;
  w->w.number_shapes(order);
  bool nan_any = false;
  for (uint64_t q = 0; q < n; q++) {
    const double* r = rays + 6 * q;
    Ray ray{Vector::point(r[0], r[1], r[2]), Vector::vector(r[3], r[4], r[5])};
    World::Ctx c;
    c.fuel0 = fuel;
    Intersection first{};
    bool did = false;
    Color col = w->w.color_at(ray, fuel, c, &first, &did);
    nan_any |= c.nan_seen;
    rgb[3 * q] = col.r; rgb[3 * q + 1] = col.g; rgb[3 * q + 2] = col.b;
    if (hits) {
      if (did) {
        int32_t seq = -2;
        for (size_t k = 0; k < order.size(); k++)
          if (order[k]->id == first.shape->id) { seq = (int32_t)k; break; }
        hits[q] = {first.t, seq, first.push_idx};
      } else hits[q] = {0.0, -1, 0};
    }
  }
  if (nan_any) return fail("NaN intersection t reached Intersection::sort (reference panics, src/intersection.rs:124)");
  return 0;
}

// Image::ppm (src/image.rs:93-112).  Returns bytes written (excluding NUL) or the size needed if cap is too small.
uint64_t orc_ppm(uint64_t hsize, uint64_t vsize, const double* rgb, char* out, uint64_t cap) {
  std::string s = ppm((size_t)hsize, (size_t)vsize, rgb);
  if (s.size() + 1 <= cap) std::memcpy(out, s.c_str(), s.size() + 1);
  return s.size();
}
// Color::clamp (src/color.rs:42-46) over n*3 channels.
void orc_quantize(const double* rgb, uint64_t n3, uint8_t* out) {
  for (uint64_t i = 0; i < n3; i++) out[i] = Color::clamp1(rgb[i]);
}
// Noise probes (unpinned by reference tests; used to cross-check the device noise).
double orc_simplex(double x, double y, double z) { return simplex(x, y, z); }
double orc_fractal(double x, double y, double z, uint64_t octaves) { return fractal(x, y, z, (size_t)octaves); }

}  // extern "C"
