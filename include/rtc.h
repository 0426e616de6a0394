/* rtc.h — the drop-in boundary of the hot path: `Image::par_render(&Camera, &World) -> Image`
 * (reference src/image.rs:65-81) and everything it calls (src/world.rs:18-149, src/intersection.rs:24-139,
 * src/shape.rs:139-158/225-269/414-462/592-946, src/bounding_box.rs:80-92, src/material.rs:164-302,
 * src/noise.rs:31-237, src/camera.rs:39-55), executed by hand-written HIP kernels on gfx950.
 *
 * The reference has no FFI of its own: the path sits behind one safe-Rust function.  A Rust
 * maintainer replaces the body of `par_render` with: flatten `&World` into an `rtc_scene_desc` (plain arrays,
 * field-wise copies — the Rust types are not #[repr(C)]), `rtc_scene_create`, `rtc_render`, copy the returned
 * doubles into `Vec<Color>`.  INTEGRATION.md shows that shim.  Every record below names the reference item it
 * carries.  All floating point is IEEE f64; matrices are row-major 4x4 exactly as `Matrix.data`.
 *
 * Ownership: `rtc_scene_create` copies everything it needs (the caller may free the arrays on return) and owns
 * all device memory; the caller owns output buffers.  Errors: status code + `rtc_last_error()`; nothing unwinds.
 * Threading: calls are blocking; distinct scenes may be used from distinct threads; one HIP stream per scene.
 * There is no CPU fallback: without a HIP device every entry point that computes returns RTC_ERR_DEVICE.
 */
#ifndef RTC_H
#define RTC_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct rtc_scene rtc_scene;

enum {
  RTC_OK = 0,
  RTC_ERR_INVALID = 1,     /* malformed description (index out of range, cyclic pattern nodes, ...)    */
  RTC_ERR_UNSUPPORTED = 2, /* valid in the reference, beyond a device limit: CSG groups nested deeper than 32; a
                              pattern with more than RTC_MAX_PATTERN_DEPTH colour frames on one path (below); a launch
                              whose CSG intersection slab would exceed RTC_CSG_MAX_BYTES (16 GiB: a subtree with more
                              than 32 possible intersections gets that many rows per thread); a material whose
                              refractive_index is not in (1e-70, 1e70); more than 64 lights; fuel above 16 */
  RTC_ERR_DEVICE = 3,      /* HIP failure / no device                                                   */
  RTC_ERR_NAN = 4          /* a NaN intersection t reached a sort the reference's comparator would run on: a list
                              of two or more entries of one World::intersect or CSG child list; the reference
                              panics there (src/intersection.rs:123-125).  A single NaN entry is legal: a slice
                              of one is never compared, the ray then has no hit                          */
};

/* Geometry (src/shape.rs:466-498). */
enum { RTC_SPHERE = 0, RTC_PLANE = 1, RTC_CUBE = 2, RTC_CYLINDER = 3, RTC_CONE = 4, RTC_TRIANGLE = 5, RTC_SMOOTH_TRIANGLE = 6 };
enum { RTC_FLAG_CASTS_SHADOW = 1u, RTC_FLAG_CLOSED = 2u };

/* Shape (src/shape.rs:297-306), one per primitive, in DFS order over world.elements (children in order).
 * The index of a record is the primitive's sequence number: intersection ties resolve by it exactly as the
 * reference's stable sort resolves them by insertion order (src/intersection.rs:123-125). */
typedef struct rtc_prim {
  int32_t geometry; /* RTC_SPHERE.. */
  uint32_t flags;   /* RTC_FLAG_* : Shape.casts_shadow, Cylinder/Cone.closed */
  int32_t material; /* index into materials (Shape.material) */
  int32_t xform;    /* index into xforms (Shape.transform_inv / material_inv; shared by a whole OBJ group) */
  int32_t data;     /* cylinder/cone: index into limits; triangles: index into tri_*; else -1 */
} rtc_prim;

/* Shape.transform_inv and Shape.material_inv after Element::propagate_inverses (src/shape.rs:47-72).
 * Shape.transform_inv_tsp is not passed: it is bitwise transpose(transform_inv). */
typedef struct rtc_xform {
  double transform_inv[16];
  double material_inv[16];
} rtc_xform;

/* Material (src/material.rs:19-28); `pattern` = index of the root rtc_pattern_node. */
typedef struct rtc_material {
  double ambient, diffuse, specular, shininess, reflective, transparency, refractive_index;
  int32_t pattern;
  int32_t _pad;
} rtc_material;

/* Pattern (src/material.rs:60-65) as a node array. */
enum { RTC_PAT_DEBUG = 0, RTC_PAT_PLAIN = 1, RTC_PAT_JITTER = 2, RTC_PAT_MIXTURE = 3 };
enum { RTC_JITTER_COLOR = 0, RTC_JITTER_POINT = 1 };
enum { RTC_MIX_BLEND = 0, RTC_MIX_CHECKERS = 1, RTC_MIX_RING_GRADIENT = 2, RTC_MIX_RING = 3, RTC_MIX_GRADIENT = 4, RTC_MIX_STRIPES = 5 };
enum { RTC_NOISE_SIMPLEX = 0, RTC_NOISE_FRACTAL = 1 };
/* Pattern trees may be of any depth (the reference's Box tree is unbounded); the device's walk keeps a frame only at nodes that
 * need both children's colours (Blend, RingGradient, Gradient) or post-process a child's colour (colour jitter): at most this many
 * of THOSE on one root-to-leaf path, else RTC_ERR_UNSUPPORTED.  Checkers, rings, stripes and point jitters nest freely. */
#define RTC_MAX_PATTERN_DEPTH 8
typedef struct rtc_pattern_node {
  int32_t tag;        /* RTC_PAT_* */
  int32_t kind;       /* RTC_JITTER_* or RTC_MIX_* */
  int32_t noise_kind; /* RTC_NOISE_* (jitter) */
  uint32_t octaves;   /* Noise::Fractal.octaves */
  int32_t left;       /* child node (jitter: the wrapped pattern), -1 if none */
  int32_t right;
  double scale;             /* Noise scale */
  double color[3];          /* Plain */
  double transform_inv[16]; /* Mixture.transform_inv */
} rtc_pattern_node;

/* PointLight (src/light.rs:5-8). */
typedef struct rtc_light {
  double intensity[3];
  double origin[3];
} rtc_light;

/* The Element tree (src/shape.rs:31-34, :181-185) in DFS pre-order.  A group node carries the world-space
 * bounding box the reference computed for it (Element::composite + propagate_inverses; NaN/inf included,
 * SURVEY Q9) and `skip` = index of the first node after its subtree.  The device evaluates
 * BoundingBox::intersects (src/bounding_box.rs:80-92) on exactly these numbers. */
enum { RTC_NODE_PRIM = -1, RTC_NODE_UNION = 0, RTC_NODE_INTERSECTION = 1, RTC_NODE_DIFFERENCE = 2, RTC_NODE_AGGREGATION = 3 };
typedef struct rtc_node {
  int32_t kind; /* RTC_NODE_* */
  int32_t ref;  /* RTC_NODE_PRIM: primitive index; groups: unused */
  int32_t skip; /* groups: one past the subtree; primitives: own index + 1 */
  int32_t _pad;
  double bbox_min[3]; /* groups only */
  double bbox_max[3];
} rtc_node;

typedef struct rtc_scene_desc {
  uint32_t n_nodes;  const rtc_node* nodes;
  uint32_t n_prims;  const rtc_prim* prims;
  uint32_t n_xforms; const rtc_xform* xforms;
  uint32_t n_limits; const double* limits;          /* n x {min, max}              (Geometry::Cylinder/Cone) */
  uint32_t n_tris;   const double* tri_p1e1e2;      /* n x {p1, e1, e2} xyz        (src/shape.rs:369-412)    */
                     const double* tri_normals;     /* n x {n1, n2, n3} xyz; flat triangles: {n, -, -}       */
  uint32_t n_materials;     const rtc_material* materials;
  uint32_t n_pattern_nodes; const rtc_pattern_node* pattern_nodes;
  uint32_t n_lights;        const rtc_light* lights;
} rtc_scene_desc;

/* Camera (src/camera.rs:5-13) with its derived fields as Camera::new computes them (:16-37). */
typedef struct rtc_camera {
  uint64_t hsize, vsize;
  double half_width, half_height, pixel_size;
  double transform_inv[16];
} rtc_camera;

/* Parity channel: the nearest hit of a primary ray (t bit-exact; prim = sequence number or -1). */
typedef struct rtc_hit {
  double t;
  int32_t prim;
  int32_t push_idx;
} rtc_hit;

/* Work counters of one render call (deterministic for a given scene + accelerator).  The *_kernarg counters say how many of
 * the nodes / tests read their record from the kernel arguments (scalar loads): those move no bytes through the memory
 * system and are excluded from bench.py's algorithmic-byte figure. */
typedef struct rtc_stats {
  uint64_t pixels;
  uint64_t rays_primary, rays_shadow, rays_reflect, rays_refract; /* unique rays, SURVEY.md §8d */
  uint64_t rays_container;   /* extra n1/n2 passes (device-internal, not counted as rays)                 */
  uint64_t accel_nodes;      /* accelerator BVH nodes visited (4-wide nodes, 128 B each)                  */
  uint64_t group_tests;      /* reference group boxes tested (BoundingBox::intersects; 48 B each)         */
  uint64_t tri_tests;        /* triangles tested (72 B each)                                              */
  uint64_t analytic_tests;   /* analytic primitives tested (one 128-B intersection record each)           */
  uint64_t nan_ts;           /* passes with a NaN t in a list of >= 2 since the last check (-> RTC_ERR_NAN) */
  double kernel_ms;          /* device time of the trace kernel(s), HIP events on the scene's stream */
  uint32_t n_launches;
  uint32_t _pad;
  uint64_t accel_nodes_kernarg;    /* of accel_nodes: root nodes read from the kernel arguments           */
  uint64_t analytic_tests_kernarg; /* of analytic_tests: plane records read from the kernel arguments     */
  uint64_t light_grid_cells;       /* light-grid cells looked up by shadow rays, each instead of a BVH walk (8 B of
                                      offsets + 4 B per candidate)                                         */
  uint64_t group_tests_uniform;    /* of group_tests: gates of whole meshes named by a kernel-argument program: every lane of
                                      the wave reads the SAME box (scalar loads of one address), so like the other
                                      *_kernarg records they move no bytes through the vector memory system */
} rtc_stats;

const char* rtc_last_error(void);
/* Number of HIP devices visible (0 = none; never initialises a context). */
int rtc_device_count(void);

/* Flatten-once upload: validates, builds the results-neutral accelerator (BVH over each group's bounded
 * primitive children) and copies SoA buffers to HBM on `device`. */
int rtc_scene_create(const rtc_scene_desc* desc, int device, rtc_scene** out);
void rtc_scene_destroy(rtc_scene*);
/* Size in bytes of the scene's device buffers (accelerator included). */
uint64_t rtc_scene_device_bytes(const rtc_scene*);

/* Image::par_render for pixels i = first .. first+n-1 (row-major: x = i % hsize, y = i / hsize) or, when
 * pixel_indices != NULL, for the n listed indices.  rgb: n*3 doubles (host).  hits, stats: optional. */
int rtc_render(rtc_scene*, const rtc_camera*, int32_t fuel, const uint64_t* pixel_indices, uint64_t first, uint64_t n,
               double* rgb, rtc_hit* hits, rtc_stats* stats);

/* Parity channel beyond the primary hit ("hit indices bit-exact" for the whole ray tree): per pixel, the wrapping 64-bit sum over
 * every ray of its de-duplicated ray tree — the primary ray and every reflected / refracted ray World::color_at spawns, each
 * once (the reference traces them once per light, src/world.rs:58-79) — of hash(t bits, primitive sequence number, push index of
 * the ray's nearest hit; ray depth; ray kind), a miss hashing as (0, -1, 0).  The hash is defined in csrc/device_scene.h
 * (rtc_hit_hash_base / rtc_hit_hash) and restated by the oracle; equal digests mean every closest hit of the pixel's ray tree
 * agrees bit for bit.  Pixels as in rtc_render; digest: n values (host).  Runs the counting kernel variants. */
int rtc_render_hit_digest(rtc_scene*, const rtc_camera*, int32_t fuel, const uint64_t* pixel_indices, uint64_t first, uint64_t n, uint64_t* digest);

/* The whole frame, quantised on the device (Color::clamp, src/color.rs:42-46 — what Image::ppm writes): hsize*vsize*3 bytes
 * (host), row-major; 3 bytes per pixel cross PCIe instead of 24. */
int rtc_render_rgb8(rtc_scene*, const rtc_camera*, int32_t fuel, uint8_t* rgb8, rtc_stats* stats);

/* Host buffers handed to rtc_render / rtc_render_rgb8 / rtc_render_multi* / rtc_trace_rays are written by the device copy
 * directly (one copy per array, queued behind the kernels; hit records are packed on the device); on an error return their
 * contents are unspecified.  RTC_PRETOUCH_THREADS=n (default 0: measured slower than the copy's own page handling) lets n host
 * threads write one byte to each destination page while the device renders. */

/* Same, output left in device memory (rgb_dev: n*3 doubles on the scene's device), for the rows
 * row_first, row_first+row_step, ... (n_rows of them) — the tile-interleaved multi-GPU partition.
 * Asynchronous on the scene's stream unless `sync` != 0.  count_stats != 0 uses the counting kernel variant. */
int rtc_render_rows_device(rtc_scene*, const rtc_camera*, int32_t fuel, uint32_t row_first, uint32_t row_step, uint32_t n_rows,
                           double* rgb_dev, rtc_stats* stats, int count_stats, int sync);

/* The partition the multi-device entries use ("8-row strips"): the image is cut into bands of band_rows rows
 * (the last one may be short) and part band_first of band_step owns bands band_first, band_first + band_step, ...; its dense
 * tile holds them in order.  A wave of the trace kernels is an 8x8 pixel tile of the DENSE tile, so band_rows = 8 keeps it an 8x8
 * tile of the image too (band_rows = 1 is rtc_render_rows_device: at 8 parts one wave's pixels then span 64 image rows).
 * Renders the first n_rows rows of that dense tile (rtc_band_rows_owned() = all of them). */
int rtc_render_bands_device(rtc_scene*, const rtc_camera*, int32_t fuel, uint32_t band_rows, uint32_t band_first, uint32_t band_step,
                            uint32_t n_rows, double* rgb_dev, rtc_stats* stats, int count_stats, int sync);
uint64_t rtc_band_rows_owned(uint64_t vsize, uint32_t band_rows, uint32_t band_first, uint32_t band_step);

/* ---- N GPUs of one process ----------------------------------------------------------------------------------------
 * For the caller of Image::par_render (src/image.rs:65-81) that owns several devices.  An rtc_multi holds one replica of the
 * scene per listed device (a device may be listed more than once — two replicas on one GPU — which is how a one-GPU box
 * exercises the whole path).  rtc_render_multi: replica k traces the bands k, k + n, ... of the image (8 rows each unless
 * rtc_multi_set_band_rows says otherwise; see rtc_render_bands_device) on its own device and stream (no
 * exchange while tracing: pixels are independent, src/image.rs:68-73); the dense tiles are pulled to the FIRST listed device over
 * xGMI (peer copies behind per-replica events), de-interleaved there, and the whole image (hsize*vsize*3 doubles, row-major) is
 * copied to `rgb` (host).  Same pixels, bit for bit, as rtc_render on one device.  stats (optional): counters summed over the
 * replicas (counting kernel variants), kernel_ms = the slowest replica.
 * One process per GPU + RCCL (bench.py, raytracer_challenge_amd/parallel.py) is the other way to use N GPUs; both partition alike. */
typedef struct rtc_multi rtc_multi;
int rtc_multi_create(const rtc_scene_desc* desc, const int* devices, int n_devices, rtc_multi** out);
void rtc_multi_destroy(rtc_multi*);
int rtc_multi_device_count(const rtc_multi*);
/* Rows per band of the partition (default 8; 1 = single rows interleaved).  Waits for queued frames. */
int rtc_multi_set_band_rows(rtc_multi*, uint32_t band_rows);
int rtc_render_multi(rtc_multi*, const rtc_camera*, int32_t fuel, double* rgb, rtc_stats* stats);
/* Same, quantised (Color::clamp, src/color.rs:42-46 — what Image::ppm writes): every replica quantises its own rows on its own
 * device, so 3 bytes per pixel cross xGMI instead of 24.  rgb8: hsize*vsize*3 bytes (host), row-major. */
int rtc_render_multi_rgb8(rtc_multi*, const rtc_camera*, int32_t fuel, uint8_t* rgb8, rtc_stats* stats);
/* Same, image left on the first listed device (rgb_dev: hsize*vsize*3 doubles there).  Asynchronous unless sync != 0: queue
 * several frames, then rtc_multi_sync() waits for all replicas and returns (and clears) their error state. */
int rtc_render_multi_device(rtc_multi*, const rtc_camera*, int32_t fuel, double* rgb_dev, int sync);
int rtc_multi_sync(rtc_multi*);

/* World::color_at(ray, fuel) for n rays {ox,oy,oz,dx,dy,dz} (host arrays). */
int rtc_trace_rays(rtc_scene*, const double* rays, uint64_t n, int32_t fuel, double* rgb, rtc_hit* hits, rtc_stats* stats);

/* ---- the step after the path: Color::clamp and Image::ppm ---------------------------------
 * Color::clamp (src/color.rs:42-46): u8 = round(min(max(c, 0), 1) * 255), round half away from zero, NaN -> 255 (Rust's
 * f64::min returns the non-NaN operand).  n_values = 3 * pixels.  Device pointers, on the scene's stream. */
int rtc_quantize_device(rtc_scene*, const double* rgb_dev, uint64_t n_values, uint8_t* out_dev, int sync);
/* Same through host buffers (upload, quantise on device, download). */
int rtc_quantize(rtc_scene*, const double* rgb, uint64_t n_values, uint8_t* out);
/* Image::ppm (src/image.rs:93-112) from quantised pixels: "P3\n{w} {h}\n255", <= 5 pixels per line, a new line at each
 * row start, trailing newline.  Host-only formatting (no device needed).  Returns the byte count (excluding the NUL);
 * writes only if cap is large enough. */
uint64_t rtc_ppm(uint64_t hsize, uint64_t vsize, const uint8_t* rgb8, char* out, uint64_t cap);

/* Waits for the stream and returns (and clears) the error state accumulated by EVERY launch since the last check or the last
 * synchronous render: RTC_ERR_NAN / RTC_ERR_DEVICE / RTC_ERR_UNSUPPORTED (a wavefront queue overflowed in an unsynchronised
 * launch).  Asynchronous launches (sync == 0, stats == NULL) do not report it themselves. */
int rtc_scene_check(rtc_scene*);
/* Stream markers for pipelined hosts: record marker `slot` (0..7) behind everything queued so far on the scene's stream;
 * wait for it on the host; device time between two recorded markers in ms.  RTC_ERR_INVALID for a slot outside 0..7 and for
 * waiting on / measuring a slot that was never recorded. */
int rtc_scene_record(rtc_scene*, int slot);
int rtc_scene_wait(rtc_scene*, int slot);
int rtc_scene_elapsed_ms(rtc_scene*, int slot_from, int slot_to, double* ms);

/* Blocks until the scene's stream is idle. */
int rtc_scene_sync(rtc_scene*);

/* Accelerator facts for reports: traversal-program length, BVH node count (4-wide nodes, 128 B each), triangles packed
 * into mesh BVH leaves, deepest BVH (levels of 4-wide nodes).  Any pointer may be NULL. */
void rtc_scene_accel_info(const rtc_scene*, uint32_t* n_ops, uint32_t* n_bvh_nodes, uint32_t* n_mesh_tris, uint32_t* bvh_depth);

/* Number of this scene's mesh accelerators whose binary tree was built on the device: a linear BVH — items sorted by the cell
 * of a longest-side-first bisection of their bounds, Karras' radix tree, csrc/bvh_device.hip — instead of the host's binned-SAH
 * build.  Default: meshes of at least 100 000 triangles (a quarter of the build time, frames within 8 % of the SAH tree's);
 * RTC_DEVICE_BVH=0: never; =1: from 4 096 triangles (RTC_DEVICE_BVH_MIN overrides the threshold).  The accelerator is
 * results-neutral: pixels and hit records do not depend on it. */
int rtc_scene_bvh_built_on_device(const rtc_scene*);

/* Dynamic LDS (bytes per block) the wavefront traversal kernel uses for this scene: > 0 = the scene's accelerator nodes, intersection
 * records and mesh triangles are copied into every CU's LDS and walks read them there (small scenes: the tables and the traversal
 * stacks fit 160 KB); 0 = they are read from memory.  bench.py's byte accounting counts LDS-resident records as 0 bytes. */
uint32_t rtc_scene_wavefront_lds_bytes(const rtc_scene*);

/* Which device path renders whole-row launches of this scene (both give bit-identical pixels and hits):
 *   1  one kernel: a lane walks its pixel's whole ray tree (rtc_trace_kernel);
 *   4  wavefront: per bounce level a closest-hit + shadow kernel and a shading kernel over ray queues (wf_* kernels).
 * RTC_KERNEL=1|4 pins a path for every launch of scenes created afterwards (pixel lists and explicit rays included: the parity
 * tests run both).  With the environment variable RTC_KERNEL unset the library measures: for one launch shape (camera, rows, fuel) the first
 * four SYNCHRONOUS launches alternate between the paths (the smaller of a path's two device times counts: a first launch
 * pays for code loading and scratch), every later launch of that shape takes the faster.  Until a shape is measured — a caller
 * that renders one frame per scene, asynchronous launches — a guess from the scene decides: wavefront iff it has >= 32 bounded
 * analytic primitives, >= 10 % of its primitives reflect or refract, fuel >= 2 and the launch has >= 256 K pixels (the one-kernel
 * path needs no ray queues).  Reports the state for
 * the most recent launch shape: *choice = 0 while undecided, else 1 or 4; the measured device times in ms (< 0 = not yet
 * measured).  Any pointer may be NULL. */
void rtc_scene_path_info(const rtc_scene*, int32_t* choice, double* one_kernel_ms, double* wavefront_ms);

#ifdef __cplusplus
}
#endif
#endif
