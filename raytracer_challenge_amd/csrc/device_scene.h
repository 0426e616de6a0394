// device_scene.h — layout of a flattened scene in HBM (shared by rtc_scene.cpp, which fills it, and
// rtc_kernels.hip, which reads it).  Everything is SoA / fixed-size records, read-only during tracing.
//
// Traversal program ("ops").  The reference gates every primitive by the box tests of its ancestor groups
// (Group::intersect, src/shape.rs:248-256) and otherwise tests all of them; the order of tests is irrelevant here because
// ties are broken by key.  So aggregation groups are dissolved: every primitive carries `gcond` (its innermost group), the
// gate is evaluated per primitive along group_parent with a per-ray cache, and ALL bounded primitives of the scene share
// one accelerator per kind.  A ray walks the array once:
//   OP_GROUP  a = group box index, b = pc to jump to when BoundingBox::intersects rejects the ray
//   OP_PRIM   a = primitive index (exact test, own world->object matrix)
//   (OP_MESH / OP_BVH: c = index of the BVH's frame in bvh_frame)
//   OP_MESH   a = BVH root, b = xform index: triangles sharing one matrix; ray transformed once, BVH in
//             object space, leaves index the packed triangle arrays (mtri / mtri_prim)
//   OP_BVH    a = BVH root: world-space BVH over analytic primitives; leaves index item_prim; b = 0 or 1 + the index in qgrids of
//             the first of n_lights light grids, which hand a shadow ray its candidates without a walk (RTC_LIGHT_CELL_WALK below)
//   OP_QUIRK  a = first, b = count in quirk_prim: the cubes and cones of the preceding OP_BVH, scanned linearly for
//             rays in the state where the reference reports intersections outside the primitive's bounds
//   OP_QGRID  a = index into qgrids: the same scan, culled by ray DIRECTION: whether a ray is a quirk ray for a cube /
//             cone depends only on its direction (thin bands on the direction sphere), so a cube-map grid of
//             directions lists, per cell, the few primitives whose band can touch the cell
//   OP_CSG    a = group box index, b = pc of the matching OP_CSG_END, c = index into csg[]: a Union / Intersection / Difference
//             group (src/shape.rs:161-178, :257-266).  Its subtree follows as a LINEAR sub-program (OP_PRIM, OP_GROUP, nested
//             OP_CSG ... OP_CSG_END; no accelerator inside).  OP_CSG_END c = the same csg index.
// The accelerator may only skip primitives whose exact test would not produce an intersection with t inside
// the interval the current pass cares about, so hit records are independent of it.
#pragma once
#include <stddef.h>
#include <stdint.h>

enum { OP_PRIM = 0, OP_GROUP = 1, OP_MESH = 2, OP_BVH = 3, OP_QUIRK = 4, OP_QGRID = 5, OP_CSG = 6, OP_CSG_END = 7 };

struct DOp {  // 32 bytes
  int32_t op, a, b, c;
  int32_t g;       // OP_MESH / OP_CSG: innermost enclosing aggregation group (-1 = none) whose box chain gates the op
  int32_t pad[3];
};

#define RTC_KOPS 12
#define RTC_KPLANES 6
struct DPlaneK {  // 72 bytes: what Plane::intersect reads (row 1 of transform_inv), the primitive's index, and the plane's world normal
  double row[4];
  int32_t prim;
  int32_t gcond;   // innermost aggregation group above the plane (its Group::intersect gate, rtc_device.hpp groups_pass), -1: none
  int32_t pad[2];
  double n[3];     // Shape::normal of the plane before the eye-side flip: the same for every point (scene_build.hpp plane_world_normal)
};

// 64-byte BVH2 node: both children's boxes (f32, rounded outward) + child refs.
// ref >= 0: inner node index.  ref < 0: leaf, items [first, first+count) with first = (~ref) >> 3,
// count = ((~ref) & 7) + 1.  An absent child has an inverted box (lo = +inf, hi = -inf).
struct DBvhNode {
  float lo0[3], hi0[3];
  float lo1[3], hi1[3];
  int32_t c0, c1;
  int32_t pad[2];
};
static_assert(sizeof(DBvhNode) == 64, "BVH node must be one 64-byte line");

// 128-byte BVH4 node, what the device walks: the builder's binary tree with every other level folded away
// (bvh::collapse4), so a ray's chain of dependent node fetches is about half as long.  Boxes as above, one
// float[4] per plane (SoA over the children).  An absent child has an inverted box and ref 0.
struct alignas(16) DBvhNode4 {
  float lox[4], loy[4], loz[4], hix[4], hiy[4], hiz[4];
  int32_t c[4];
  int32_t pad[4];
};
static_assert(sizeof(DBvhNode4) == 128, "BVH4 node must be one 128-byte line");

// Direction grid of one OP_QGRID: cube map, n x n cells per face; cell c owns qitem[qcell[cell_off + c] .. qcell[cell_off + c + 1]).
// Rays shorter than RTC_QGRID_MIN_LEN (or non-finite) scan quirk_prim[lin_first .. +lin_count) instead.
struct DQuirkGrid {
  int32_t n, cell_off, lin_first, lin_count;
};
#define RTC_QGRID_MIN_LEN 0.05
// Light grids (scene_build.hpp build_light_grids; OP_BVH b = 1 + index of light 0's grid in qgrids, 0 = none): the same cell function;
// a cell's items are the BVH leaf references of its candidates, or this one value: too many candidates, walk the BVH.
#define RTC_LIGHT_CELL_WALK 0x7fffffff
struct alignas(8) DLightItem { int32_t ref; float dmin; };  // dmin: no point of the candidate's bounds is nearer to the light (nearest first)

// Kernel-argument copy of what every ray reads first of a BVH: its frame, (meshes) the world -> object matrix, the root node.
#define RTC_KAUX 3
struct DKAux {
  double frame[4];
  double xf[12];
  DBvhNode4 root;
};

// One CSG group: kind (RTC_NODE_UNION / _INTERSECTION / _DIFFERENCE) and the primitive range of children[0] (DFS order makes
// "children[0].includes(shape)" a range test on the sequence number).
struct DCsg {
  int32_t kind, left_first, left_end, pad;
};
#define RTC_CSG_MAX_HITS 32   // intersections of one top-level CSG subtree that fit the per-lane buffer; a scene whose subtrees can
                              // produce more (DScene.csg_max_hits, computed at scene creation) gets a slab in device memory with
                              // csg_max_hits entries per lane of the launch (the reference's lists are unbounded, src/shape.rs:248-269)
#define RTC_CSG_MAX_DEPTH 32  // nested CSG groups a sub-program may hold (one int of the CSG kernels' per-lane frame per level; round 2: 8)
struct DCsgHit { double t; int32_t prim, k; };
// A pixel's primary-hit record as the C ABI hands it out (include/rtc.h rtc_hit): packed on the device from the kernels' SoA rows.
struct DHit { double t; int32_t prim, k; };

struct DPrim {  // 32 bytes
  int32_t geom;
  uint32_t flags;
  int32_t mat;
  int32_t xform;
  int32_t data;
  int32_t gcond;   // innermost enclosing aggregation group outside any CSG (-1 = none): the primitive is tested only if the
                   // reference's box test passes for that group and all its ancestors (group_parent chain)
  int32_t pad[2];  // pad[0]: planes whose record travels in the kernel arguments: 1 + slot in DScene.kplanes (its world normal is there), else 0
};

// Everything ONE intersection test reads, in one 128-byte line (a divergent wave fetches one line per lane instead of
// chasing item -> DPrim -> matrix -> limits).  Built by the host from prims / xf_inv / limits; index = primitive index.
struct DPrimI {
  int32_t geom;
  uint32_t flags;
  int32_t data;    // triangle index (geom >= 5)
  int32_t gcond;
  double mn, mx;   // cylinder / cone limits
  double m[12];    // rows 0..2 of transform_inv
};

struct DPat {  // pattern node, 192 bytes
  int32_t tag, kind, noise_kind;
  uint32_t octaves;
  int32_t left, right;
  int32_t pad[2];
  double scale;
  double color[3];
  double m[16];
};

struct DScene {
  const DOp* ops;
  const int32_t* group_parent;  // per group box index: enclosing aggregation group or -1 (boxes used by OP_GROUP/OP_CSG have -1)
  const double* group_box;   // n_groups x {lo[3], hi[3]} f64 exactly as the reference computed them
  const DBvhNode4* bvh;
  const double* mtri;        // packed leaf-order triangles x {p1, e1, e2}
  const int32_t* mtri_prim;  // -> primitive sequence number
  const DPrimI* pisect;      // per primitive: the intersection record
  const int32_t* item_prim;  // unused by OP_BVH since its leaf refs carry the primitive index itself (one primitive per leaf)
  const int32_t* quirk_prim; // OP_QUIRK items -> primitive index (cubes, cones)
  const DQuirkGrid* qgrids;
  const uint32_t* qcell;     // per-cell offsets into qitem
  const DCsg* csg;
  DCsgHit* csg_slab;         // csg_max_hits > RTC_CSG_MAX_HITS: csg_max_hits entries per thread of the launch (else NULL)
  const double* bvh_frame;   // per BVH (DOp.c of OP_MESH / OP_BVH): centre xyz + inf-norm radius; node boxes are relative to the centre
  const int32_t* qitem;      // primitive indices
  const DPrim* prims;
  const double* xf_inv;      // n_xforms x 12 (rows 0..2 of Shape.transform_inv)
  const double* xf_matinv;   // n_xforms x 16 (Shape.material_inv)
  const double* limits;      // x {min, max}
  const double* tri_geo;     // x {p1, e1, e2}
  const double* tri_nrm;     // x {n1, n2, n3}
  const double* mat;         // n_materials x 8 {ambient, diffuse, specular, shininess, reflective, transparency, refractive_index, -}
  const int32_t* mat_pattern;
  const DPat* pats;
  const double* lights;      // n_lights x {intensity rgb, origin xyz}
  int32_t n_ops, n_prims, n_lights;
  int32_t all_cast_shadow;   // 1: every primitive casts a shadow -> shadow rays may stop at any hit
  int32_t has_mesh;          // 1: the program contains an OP_MESH
  int32_t has_csg;           // 1: the program contains an OP_CSG
  int32_t has_groups;        // 0: no gates; 1: only OP_MESH / OP_CSG ops are gated; 2: individual primitives are gated
  int32_t csg_max_hits;      // most intersections one top-level CSG subtree can produce
  // Cube quirk rays without a scan (scene_build.hpp, cube_pad): a ray with 48 m^2 <= reach^2 |d|^2 — m = |origin - centre|_inf + radius of
  // the analytic BVH's frame — cannot report a cube intersection outside the padded leaf boxes, so the leaves test it like any other
  // ray and the cubes' OP_QGRID / OP_QUIRK (c = 1) is skipped.  quirk_reach2 = reach^2 / 48, 0: feature off.
  double quirk_reach2;
  double abvh_frame[4];
  int32_t light_grid_n, light_grid_cell_off;  // the light grids' common n and light 0's cell_off (light l: + l * (6 n^2 + 1))
  int32_t light_grid_first;  // 0 or 1 + index in qgrids of light 0's light grid (= the b of the program's OP_BVH)
  int32_t all_plain;         // 1: every material's pattern is a Plain colour (Pattern::color_at never walks a tree: kernels without the pattern stack)
  int32_t no_glass_mirror;   // 1: no material both reflects and refracts (a hit never has two children: the one-kernel path stacks no pending ray)
  int32_t backface_skip;     // 1: a shadow ray towards a light BEHIND the surface is answered without a traversal (rtc_device.hpp
                             // light_is_behind; one-kernel path): matrices, triangle corners and lights are all below 1e30 in magnitude and no
                             // matrix flattens space, so a finite shadow ray cannot produce a NaN t
  int32_t has_recs;          // 1: some op reads intersection records (pisect): analytic BVH, quirk scans, primitives outside the kernel arguments
  int32_t n_recs;            // entries of pisect: 0, or 1 + the last primitive an op can name (scene_build.hpp build_arrays) -- NOT n_prims: the
                             // triangles of a mesh at the end of the world have no record
  // Kernel-argument copy of a short traversal program (kernargs are read with scalar loads: the op fetch and the plane
  // records stop being per-lane vector loads on every ray's dependency chain).  Used when n_kops > 0: the whole program
  // has <= RTC_KOPS ops, no OP_GROUP / OP_CSG, no per-primitive gates (so every lane runs the same op sequence); an
  // OP_PRIM whose primitive is a plane carries its slot in kplanes in `c` (else -1).
  int32_t n_kops, n_kplanes;
  DOp kops[RTC_KOPS];        // pad[0]: OP_BVH / OP_MESH -> slot in kaux, OP_QGRID -> 0 if kqgrid is its grid; -1 = read from memory
  DPlaneK kplanes[RTC_KPLANES];
  DKAux kaux[RTC_KAUX];
  DQuirkGrid kqgrid;
  // array lengths, for the traversal guards (a bad index retires the lane and raises DStats.guard instead of faulting)
  int32_t bvh_stack;  // entries each lane's traversal stack needs for this scene's trees (LDS is sized from it at launch)
  int32_t n_bvh, n_items, n_mtri, n_quirk, n_qitem, n_qcell, n_groups, n_qgrids;
};

// Which pixels a launch covers.
struct DPixelMap {
  uint64_t n;                 // pixels in this launch
  uint64_t first;             // unused on device (a contiguous range is issued as mode 2, step 1, or as mode 1)
  const uint64_t* indices;    // mode 1: i = indices[q]
  uint32_t mode;              // 1 list, 2 interleaved rows, 3 explicit rays
  uint32_t row_first, row_step;  // mode 2: dense row j = q / hsize of the launch is image row (row_first + (j / band) * row_step) * band + j % band,
  const double* rays;         // mode 3: n x {o, d}
  uint32_t band;              // mode 2: rows per band (0 = 1): row_first / row_step count bands.  Bands of 8 rows keep a wave's 8x8 pixel tile
  uint32_t pad;               //   contiguous in the image when the rows of a frame are dealt out to several devices
  unsigned long long* digest; // parity channel (counting variants only; NULL = off): per output slot, the hit-tree digest (rtc_hit_hash below)
};

struct DCamera {
  uint64_t hsize, vsize;
  double half_width, half_height, pixel_size;
  double inv[16];
};

struct DStats {  // device-side counters (atomically accumulated per wave)
  // per-launch counters: zeroed on the stream in front of every launch (the first RTC_STATS_LAUNCH_BYTES bytes)
  unsigned long long rays_primary, rays_shadow, rays_reflect, rays_refract, rays_container;
  unsigned long long accel_nodes, group_tests, tri_tests, analytic_tests;
  unsigned long long knodes;    // of accel_nodes: BVH root nodes read from the kernel arguments (scalar loads, no memory traffic)
  unsigned long long kplanes;   // of analytic_tests: plane records read from the kernel arguments
  unsigned long long light_cells;  // light-grid cells looked up by shadow rays (each in place of a BVH walk)
  unsigned long long kgroups;   // of group_tests: gates named by a kernel-argument program (one box for the whole wave: scalar loads)
  unsigned long long diag[64];  // RTC_DIAG builds only: region cycles / lane-utilisation sums (scripts/diag_report.py)
  // sticky error state: accumulated over every launch since the last rtc_scene_check() / synchronous read-back, which clear it
  unsigned long long nan_ts;       // NaN intersection t's seen (-> RTC_ERR_NAN)
  unsigned long long guard;        // bit mask of tripped traversal guards (0 = none)
  unsigned long long guard_claim;  // first tripping lane claims the info slots
  long long guard_info[8];         // code, it_kind, it, it_end, value, cur, pc, mode of the first trip
  unsigned long long wf_overflow;  // a wavefront ray queue overflowed in some launch since the last check
};
#define RTC_STATS_LAUNCH_BYTES offsetof(DStats, nan_ts)

#define RTC_MAX_FUEL 16

// Hit-tree digest (parity channel, include/rtc.h rtc_render_hit_digest): a pixel's digest is the wrapping 64-bit sum, over every
// ray of its de-duplicated ray tree (the primary ray, every reflected / refracted ray; shadow rays are not hits of the tree), of
// rtc_hit_hash(closest hit of the ray, depth, kind) — a miss hashes as (t bits 0, primitive -1, push 0).  Device and oracle share
// this definition only; each computes its own hits.  kind: 0 primary, 1 reflected, 2 refracted.
#if defined(__HIPCC__) || defined(RTC_EMU)
#define RTC_HD __host__ __device__
#else
#define RTC_HD
#endif
static inline RTC_HD unsigned long long rtc_mix64(unsigned long long x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
  return x;
}
static inline RTC_HD unsigned long long rtc_hit_hash_base(unsigned long long t_bits, int32_t prim, int32_t push) {
  return rtc_mix64(t_bits) ^ ((((unsigned long long)(uint32_t)prim << 32) | (unsigned long long)(uint32_t)push) * 0x9E3779B97F4A7C15ull);
}
static inline RTC_HD unsigned long long rtc_hit_hash(unsigned long long base, int depth, int kind) {
  return rtc_mix64(base ^ ((unsigned long long)(((unsigned)depth << 8) | (unsigned)kind) + 1ull) * 0xD6E8FEB86659FD93ull);
}

// Wavefront path (rtc_device.hpp / rtc_kernels.hip, wf_* kernels): rays of one bounce level live in a queue; per level a traversal launch
// (closest hits of the level + shadow rays and lighting of the previous level) and a shading launch (hit state, pattern
// colour, child rays into the other queue), so the traversal kernel carries no shading state (half the registers of the
// one-kernel path -> twice the resident waves).  All arrays are SoA rows of `cap` elements.  A ray's colour contribution is
// stored per level; wf_gather adds a pixel's contributions in the one-kernel path's depth-first order, so both paths give the
// same bits and a pixel's value does not depend on what else was rendered with it.
#define RTC_WF_SHADE_COUNT 32  // counts[RTC_WF_SHADE_COUNT + level] = shade records of the level
#define RTC_WF_OVERFLOW 63     // counts[RTC_WF_OVERFLOW] != 0: a queue overflowed, the frame must be rendered by the one-kernel path
#define RTC_WF_CHUNK_NEXT 64   // counts[RTC_WF_CHUNK_NEXT + 32 * (8 * launch + xcd)]: next chunk of traversal launch `launch` for blocks of
                               // that XCD; one 128-byte line per cursor (same-line device atomics serialise)
#define RTC_WF_COUNTS (64 + 32 * 8 * (RTC_MAX_FUEL + 2))
#ifndef RTC_WF_CHUNK
#define RTC_WF_CHUNK 64u       // work items per chunk: one wave pass (larger chunks leave waves idle at the small, deep levels: 256 -> +25 %)
#endif
#define RTC_WF_MISS (-2)        // child row 0 of a ray that hit nothing: no contribution was written for it, it has no children
struct DWave {
  double* rq[2];        // ray queues (level parity): 7 rows ox oy oz dx dy dz weight
  double* h_t;          // per ray of the level: closest hit t
  int32_t* h_prim;      //   primitive (-1 miss / padding)
  double* h_n12;        //   2 rows: n1, n2 — written and read only for hits on transparent surfaces that can still spawn rays
  double* sr;           // shade records (compact): 9 rows over-point(3) normal(3) colour(3); the eye vector and the path weight
                        //   are read back from the ray queue (eye = -direction), which still holds the level while its shadow pass runs
  int32_t* sr_mat;      //   material index
  int32_t* sr_node;     //   ray index within the level
  double* contrib;      // (levels) x 3 rows: colour contribution of each ray that hit something (written by the shadow pass)
  int32_t* child;       // (levels) x 2 rows: index of the reflected / refracted child ray in the next level (-1 none); row 0 = RTC_WF_MISS: no hit
  uint32_t* counts;     // RTC_WF_COUNTS counters: [level] rays of the level, [32 + level] shade records, [63] overflow flag, [64..] chunk cursors
  uint32_t cap;
  uint32_t pad;
  unsigned long long* dig;  // hit-tree digest launches only (else NULL; its own allocation): (levels) x cap rtc_hit_hash_base of each ray's closest hit
};
// Bytes of the arrays above for `cap` elements per row and `levels` levels, and the carving of one allocation into them (shared
// by rtc_scene.cpp and the CPU emulator of the kernels).  Per element at fuel 5: 50 doubles + 17 ints = 468 B.
static inline uint64_t dwave_bytes(uint64_t cap, int levels) {
  return cap * (uint64_t)(7 + 7 + 1 + 2 + 9 + 3 * levels) * sizeof(double) + (cap * (uint64_t)(1 + 1 + 1 + 2 * levels) + RTC_WF_COUNTS) * sizeof(int32_t);
}
static inline void dwave_carve(DWave* W, void* mem, uint64_t cap, int levels) {
  double* d = (double*)mem;
  W->rq[0] = d; d += 7 * cap;
  W->rq[1] = d; d += 7 * cap;
  W->h_t = d; d += cap;
  W->h_n12 = d; d += 2 * cap;
  W->sr = d; d += 9 * cap;
  W->contrib = d; d += 3 * (uint64_t)levels * cap;
  int32_t* q = (int32_t*)d;
  W->h_prim = q; q += cap;
  W->sr_mat = q; q += cap;
  W->sr_node = q; q += cap;
  W->child = q; q += 2 * (uint64_t)levels * cap;
  W->counts = (uint32_t*)q;
  W->cap = (uint32_t)cap;
  W->pad = 0;
  W->dig = nullptr;
}
#ifndef RTC_BVH_STACK
#define RTC_BVH_STACK 64
#endif
