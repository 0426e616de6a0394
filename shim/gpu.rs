//! gpu.rs — the reference crate's side of the drop-in boundary (include/rtc.h): add this file as `src/gpu.rs`, declare it
//! with `pub mod gpu;` in `src/lib.rs`, apply the two small edits of shim/README.md (a `pub(crate)` accessor in camera.rs, the
//! new body of `Image::par_render` in image.rs) and link `librtc_amd.so` (shim/build.rs).
//!
//! NOT COMPILED in the build image of this repository (no cargo / rustc there).  What IS tested: tests/foreign_flattener.py is
//! this file's walk restated in Python/ctypes against the same C structs, and tests/test_cabi_desc.py proves that a descriptor
//! produced that way renders bit-identically to the library's own host mirror.
//!
//! Every record is copied field by field (`Matrix`, `Vector`, `Color` are not `#[repr(C)]`); the index at which a `Shape` is
//! emitted (depth first over `world.elements`, children in order) is its sequence number: the device resolves intersection
//! ties by it exactly as the reference's stable sort resolves them by insertion order (src/intersection.rs:123-125).
use crate::camera::Camera;
use crate::color::Color;
use crate::linalg::{Matrix, Vector};
use crate::material::{JitterKind, Material, MixtureKind, Pattern};
use crate::noise::Noise;
use crate::shape::{Element, Geometry, GroupKind, Shape};
use crate::world::World;

use std::os::raw::{c_char, c_int};

// ---- include/rtc.h ----------------------------------------------------------------------------------------------------------
#[repr(C)]
#[derive(Clone, Copy)]
pub struct RtcPrim {
    pub geometry: i32, // RTC_SPHERE = 0, PLANE, CUBE, CYLINDER, CONE, TRIANGLE, SMOOTH_TRIANGLE = 6
    pub flags: u32,    // 1 = casts_shadow, 2 = closed
    pub material: i32,
    pub xform: i32,
    pub data: i32, // cylinder / cone: index into limits; triangles: index into tri_*; else -1
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct RtcXform {
    pub transform_inv: [f64; 16],
    pub material_inv: [f64; 16],
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct RtcMaterial {
    pub ambient: f64,
    pub diffuse: f64,
    pub specular: f64,
    pub shininess: f64,
    pub reflective: f64,
    pub transparency: f64,
    pub refractive_index: f64,
    pub pattern: i32,
    pub _pad: i32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct RtcPatternNode {
    pub tag: i32,        // 0 Debug, 1 Plain, 2 Jitter, 3 Mixture
    pub kind: i32,       // JitterKind (Color 0, Point 1) or MixtureKind (Blend 0 .. Stripes 5)
    pub noise_kind: i32, // 0 Simplex, 1 Fractal
    pub octaves: u32,
    pub left: i32, // child node (jitter: the wrapped pattern), -1 if none
    pub right: i32,
    pub scale: f64,
    pub color: [f64; 3],
    pub transform_inv: [f64; 16],
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct RtcLight {
    pub intensity: [f64; 3],
    pub origin: [f64; 3],
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct RtcNode {
    pub kind: i32, // -1 primitive; 0 Union, 1 Intersection, 2 Difference, 3 Aggregation
    pub prim: i32, // (`ref` in rtc.h) primitive index, groups: -1
    pub skip: i32, // groups: one past the subtree; primitives: own index + 1
    pub _pad: i32,
    pub bbox_min: [f64; 3], // groups only: Group.bbox exactly as the reference computed it (NaN / inf included)
    pub bbox_max: [f64; 3],
}

#[repr(C)]
pub struct RtcSceneDesc {
    pub n_nodes: u32,
    pub nodes: *const RtcNode,
    pub n_prims: u32,
    pub prims: *const RtcPrim,
    pub n_xforms: u32,
    pub xforms: *const RtcXform,
    pub n_limits: u32,
    pub limits: *const f64, // n x {min, max}
    pub n_tris: u32,
    pub tri_p1e1e2: *const f64,  // n x {p1, e1, e2} xyz
    pub tri_normals: *const f64, // n x {n1, n2, n3} xyz; flat triangles: {n, -, -}
    pub n_materials: u32,
    pub materials: *const RtcMaterial,
    pub n_pattern_nodes: u32,
    pub pattern_nodes: *const RtcPatternNode,
    pub n_lights: u32,
    pub lights: *const RtcLight,
}

#[repr(C)]
pub struct RtcCamera {
    pub hsize: u64,
    pub vsize: u64,
    pub half_width: f64,
    pub half_height: f64,
    pub pixel_size: f64,
    pub transform_inv: [f64; 16],
}

#[repr(C)]
pub struct RtcHit {
    pub t: f64,
    pub prim: i32,
    pub push_idx: i32,
}

#[repr(C)]
pub struct RtcStats {
    pub pixels: u64,
    pub rays_primary: u64,
    pub rays_shadow: u64,
    pub rays_reflect: u64,
    pub rays_refract: u64,
    pub rays_container: u64,
    pub accel_nodes: u64,
    pub group_tests: u64,
    pub tri_tests: u64,
    pub analytic_tests: u64,
    pub nan_ts: u64,
    pub kernel_ms: f64,
    pub n_launches: u32,
    pub _pad: u32,
    pub accel_nodes_kernarg: u64,
    pub analytic_tests_kernarg: u64,
    pub light_grid_cells: u64,
    pub group_tests_uniform: u64,
}

pub enum RtcScene {}
pub enum RtcMulti {}

pub const RTC_OK: c_int = 0;
pub const RTC_ERR_UNSUPPORTED: c_int = 2;

extern "C" {
    fn rtc_last_error() -> *const c_char;
    fn rtc_device_count() -> c_int;
    fn rtc_scene_create(desc: *const RtcSceneDesc, device: c_int, out: *mut *mut RtcScene) -> c_int;
    fn rtc_scene_destroy(scene: *mut RtcScene);
    fn rtc_render(scene: *mut RtcScene, camera: *const RtcCamera, fuel: i32, pixel_indices: *const u64, first: u64, n: u64,
                  rgb: *mut f64, hits: *mut RtcHit, stats: *mut RtcStats) -> c_int;
    fn rtc_multi_create(desc: *const RtcSceneDesc, devices: *const c_int, n_devices: c_int, out: *mut *mut RtcMulti) -> c_int;
    fn rtc_multi_destroy(multi: *mut RtcMulti);
    fn rtc_render_multi(multi: *mut RtcMulti, camera: *const RtcCamera, fuel: i32, rgb: *mut f64, stats: *mut RtcStats) -> c_int;
    #[allow(dead_code)] // Color::clamp'ed pixels (what Image::ppm writes): 3 bytes per pixel cross xGMI instead of 24
    fn rtc_render_multi_rgb8(multi: *mut RtcMulti, camera: *const RtcCamera, fuel: i32, rgb8: *mut u8, stats: *mut RtcStats) -> c_int;
    #[allow(dead_code)] // the same on one device: quantised on the GPU, 3 bytes per pixel cross PCIe
    fn rtc_render_rgb8(scene: *mut RtcScene, camera: *const RtcCamera, fuel: i32, rgb8: *mut u8, stats: *mut RtcStats) -> c_int;
}

#[derive(Debug)]
pub struct GpuError {
    pub code: i32,
    pub message: String,
}

fn last_error(code: c_int) -> GpuError {
    let message = unsafe {
        let p = rtc_last_error();
        if p.is_null() { String::new() } else { std::ffi::CStr::from_ptr(p).to_string_lossy().into_owned() }
    };
    GpuError { code, message }
}

// ---- the flattener -----------------------------------------------------------------------------------------
fn flat16(m: &Matrix) -> [f64; 16] {
    let mut a = [0.0; 16];
    for r in 0..4 {
        for c in 0..4 {
            a[4 * r + c] = m[r][c];
        }
    }
    a
}

fn xyz(v: Vector) -> [f64; 3] {
    [v.x, v.y, v.z]
}

fn same_bits(a: &[f64; 16], b: &[f64; 16]) -> bool {
    a.iter().zip(b.iter()).all(|(x, y)| x.to_bits() == y.to_bits())
}

fn same_f64(a: f64, b: f64) -> bool {
    a.to_bits() == b.to_bits()
}

/// Bitwise structural equality (the reference's `Approx` is epsilon-based: not what de-duplication needs).
fn same_pattern(a: &Pattern, b: &Pattern) -> bool {
    match (a, b) {
        (Pattern::Debug, Pattern::Debug) => true,
        (Pattern::Plain { color: x }, Pattern::Plain { color: y }) => same_f64(x.r, y.r) && same_f64(x.g, y.g) && same_f64(x.b, y.b),
        (Pattern::Jitter { kind: ka, noise: na, pattern: pa }, Pattern::Jitter { kind: kb, noise: nb, pattern: pb }) => {
            let kinds = matches!((ka, kb), (JitterKind::Color, JitterKind::Color) | (JitterKind::Point, JitterKind::Point));
            let noises = match (na, nb) {
                (Noise::Simplex { scale: x }, Noise::Simplex { scale: y }) => same_f64(*x, *y),
                (Noise::Fractal { scale: x, octaves: ox }, Noise::Fractal { scale: y, octaves: oy }) => same_f64(*x, *y) && ox == oy,
                _ => false,
            };
            kinds && noises && same_pattern(pa, pb)
        }
        (Pattern::Mixture { kind: ka, transform_inv: ta, left: la, right: ra }, Pattern::Mixture { kind: kb, transform_inv: tb, left: lb, right: rb }) => {
            (*ka as i32) == (*kb as i32) && same_bits(&flat16(ta), &flat16(tb)) && same_pattern(la, lb) && same_pattern(ra, rb)
        }
        _ => false,
    }
}

fn same_material(a: &Material, b: &Material) -> bool {
    same_f64(a.ambient, b.ambient) && same_f64(a.diffuse, b.diffuse) && same_f64(a.specular, b.specular) && same_f64(a.shininess, b.shininess)
        && same_f64(a.reflective, b.reflective) && same_f64(a.transparency, b.transparency) && same_f64(a.refractive_index, b.refractive_index)
        && same_pattern(&a.pattern, &b.pattern)
}

#[derive(Default)]
pub struct Flat<'a> {
    last_material: Option<(&'a Material, i32)>,
    nodes: Vec<RtcNode>,
    prims: Vec<RtcPrim>,
    xforms: Vec<RtcXform>,
    limits: Vec<f64>,
    tri_geo: Vec<f64>,
    tri_nrm: Vec<f64>,
    materials: Vec<RtcMaterial>,
    pats: Vec<RtcPatternNode>,
    lights: Vec<RtcLight>,
}

impl<'a> Flat<'a> {
    pub fn from_world(world: &'a World) -> Flat<'a> {
        let mut f = Flat::default();
        for l in &world.lights {
            f.lights.push(RtcLight { intensity: [l.intensity.r, l.intensity.g, l.intensity.b], origin: xyz(l.origin) });
        }
        for e in &world.elements {
            f.walk(e);
        }
        f
    }

    /// Pattern tree -> node array, children before their parent (src/material.rs:60-65).
    fn pattern(&mut self, p: &Pattern) -> i32 {
        let id = flat16(&Matrix::id());
        let node = match p {
            Pattern::Debug => RtcPatternNode { tag: 0, kind: 0, noise_kind: 0, octaves: 1, left: -1, right: -1, scale: 1.0, color: [0.0; 3], transform_inv: id },
            Pattern::Plain { color } => {
                RtcPatternNode { tag: 1, kind: 0, noise_kind: 0, octaves: 1, left: -1, right: -1, scale: 1.0, color: [color.r, color.g, color.b], transform_inv: id }
            }
            Pattern::Jitter { kind, noise, pattern } => {
                let child = self.pattern(pattern);
                let (noise_kind, scale, octaves) = match noise {
                    Noise::Simplex { scale } => (0, *scale, 1u32),
                    Noise::Fractal { scale, octaves } => (1, *scale, *octaves as u32),
                };
                let kind = match kind { JitterKind::Color => 0, JitterKind::Point => 1 };
                RtcPatternNode { tag: 2, kind, noise_kind, octaves, left: child, right: -1, scale, color: [0.0; 3], transform_inv: id }
            }
            Pattern::Mixture { kind, transform_inv, left, right } => {
                let l = self.pattern(left);
                let r = self.pattern(right);
                let kind = match kind {
                    MixtureKind::Blend => 0, MixtureKind::Checkers => 1, MixtureKind::RingGradient => 2,
                    MixtureKind::Ring => 3, MixtureKind::Gradient => 4, MixtureKind::Stripes => 5,
                };
                RtcPatternNode { tag: 3, kind, noise_kind: 0, octaves: 1, left: l, right: r, scale: 1.0, color: [0.0; 3], transform_inv: flat16(transform_inv) }
            }
        };
        self.pats.push(node);
        self.pats.len() as i32 - 1
    }

    /// Every `Shape` owns a clone of its material; consecutive shapes of one OBJ group (10^6 of them in config 5) hold equal
    /// ones: the previous record is reused when it is bitwise the same.
    fn material(&mut self, m: &'a Material) -> i32 {
        if let Some((prev, index)) = self.last_material {
            if same_material(prev, m) {
                return index;
            }
        }
        let pattern = self.pattern(&m.pattern);
        self.materials.push(RtcMaterial {
            ambient: m.ambient, diffuse: m.diffuse, specular: m.specular, shininess: m.shininess, reflective: m.reflective,
            transparency: m.transparency, refractive_index: m.refractive_index, pattern, _pad: 0,
        });
        let index = self.materials.len() as i32 - 1;
        self.last_material = Some((m, index));
        index
    }

    /// Shape.transform_inv / material_inv after propagate_inverses; consecutive primitives of one OBJ group carry bit-identical
    /// matrices and share the record (the device then transforms a ray once per mesh instead of once per triangle).
    fn xform(&mut self, s: &Shape) -> i32 {
        let (ti, mi) = (flat16(&s.transform_inv), flat16(&s.material_inv));
        if let Some(last) = self.xforms.last() {
            if same_bits(&last.transform_inv, &ti) && same_bits(&last.material_inv, &mi) {
                return self.xforms.len() as i32 - 1;
            }
        }
        // rtc.h does not take transform_inv_tsp: it is bitwise transpose(transform_inv) by construction (src/shape.rs:57-58, :338-339)
        debug_assert!(same_bits(&flat16(&s.transform_inv_tsp), &flat16(&s.transform_inv.transpose())));
        self.xforms.push(RtcXform { transform_inv: ti, material_inv: mi });
        self.xforms.len() as i32 - 1
    }

    fn tri(&mut self, p1: Vector, e1: Vector, e2: Vector, n: [Vector; 3]) -> i32 {
        for v in [p1, e1, e2].iter() {
            self.tri_geo.extend_from_slice(&xyz(*v));
        }
        for v in n.iter() {
            self.tri_nrm.extend_from_slice(&xyz(*v));
        }
        (self.tri_geo.len() / 9) as i32 - 1
    }

    fn walk(&mut self, e: &'a Element) {
        match e {
            Element::Primitive(s) => {
                let material = self.material(&s.material);
                let xform = self.xform(s);
                let (geometry, data, closed) = match &s.geometry {
                    Geometry::Sphere => (0, -1, false),
                    Geometry::Plane => (1, -1, false),
                    Geometry::Cube => (2, -1, false),
                    Geometry::Cylinder { min, max, closed } => {
                        self.limits.extend_from_slice(&[*min, *max]);
                        (3, (self.limits.len() / 2) as i32 - 1, *closed)
                    }
                    Geometry::Cone { min, max, closed } => {
                        self.limits.extend_from_slice(&[*min, *max]);
                        (4, (self.limits.len() / 2) as i32 - 1, *closed)
                    }
                    Geometry::Triangle { p1, e1, e2, n, .. } => (5, self.tri(*p1, *e1, *e2, [*n, *n, *n]), false),
                    Geometry::SmoothTriangle { p1, e1, e2, n1, n2, n3, .. } => (6, self.tri(*p1, *e1, *e2, [*n1, *n2, *n3]), false),
                };
                let flags = (s.casts_shadow as u32) | ((closed as u32) << 1);
                self.prims.push(RtcPrim { geometry, flags, material, xform, data });
                let at = self.nodes.len() as i32;
                self.nodes.push(RtcNode { kind: -1, prim: self.prims.len() as i32 - 1, skip: at + 1, _pad: 0, bbox_min: [0.0; 3], bbox_max: [0.0; 3] });
            }
            Element::Composite(g) => {
                let kind = match g.kind { GroupKind::Union => 0, GroupKind::Intersection => 1, GroupKind::Difference => 2, GroupKind::Aggregation => 3 };
                let at = self.nodes.len();
                self.nodes.push(RtcNode { kind, prim: -1, skip: 0, _pad: 0, bbox_min: xyz(g.bbox.min), bbox_max: xyz(g.bbox.max) });
                for c in &g.children {
                    self.walk(c);
                }
                self.nodes[at].skip = self.nodes.len() as i32;
            }
        }
    }

    /// Borrowed view for rtc_scene_create (which copies everything it needs before it returns).
    pub fn desc(&self) -> RtcSceneDesc {
        RtcSceneDesc {
            n_nodes: self.nodes.len() as u32, nodes: self.nodes.as_ptr(),
            n_prims: self.prims.len() as u32, prims: self.prims.as_ptr(),
            n_xforms: self.xforms.len() as u32, xforms: self.xforms.as_ptr(),
            n_limits: (self.limits.len() / 2) as u32, limits: self.limits.as_ptr(),
            n_tris: (self.tri_geo.len() / 9) as u32, tri_p1e1e2: self.tri_geo.as_ptr(), tri_normals: self.tri_nrm.as_ptr(),
            n_materials: self.materials.len() as u32, materials: self.materials.as_ptr(),
            n_pattern_nodes: self.pats.len() as u32, pattern_nodes: self.pats.as_ptr(),
            n_lights: self.lights.len() as u32, lights: self.lights.as_ptr(),
        }
    }
}

fn rtc_camera(camera: &Camera) -> RtcCamera {
    // Camera's derived fields are private (src/camera.rs:9-12): shim/README.md adds `pub(crate) fn raw(&self)` next to them
    let (transform_inv, pixel_size, half_width, half_height) = camera.raw();
    RtcCamera { hsize: camera.hsize as u64, vsize: camera.vsize as u64, half_width, half_height, pixel_size, transform_inv: flat16(&transform_inv) }
}

/// `Image::par_render`'s pixels on the MI355X: row-major `Vec<Color>` of `hsize * vsize`, as the reference collects them
/// (src/image.rs:66-74).  `fuel` is the reference's compile-time `FUEL` (src/config.rs:2).  All visible devices are used
/// (8-row bands dealt round-robin over the devices, gathered to the first); `Err(code == RTC_ERR_UNSUPPORTED)` = a scene beyond a
/// device limit (include/rtc.h: CSG groups nested deeper than 32, more than 8 colour frames on one pattern path, more than 64 lights,
/// fuel above 16, a refractive index outside (1e-70, 1e70), an intersection slab beyond the memory budget): keep the CPU body for
/// that scene.  `RTC_ERR_NAN` = the reference would have panicked in `Intersection::sort`: panic here too.
pub fn render(camera: &Camera, world: &World, fuel: i32) -> Result<Vec<Color>, GpuError> {
    let flat = Flat::from_world(world);
    let desc = flat.desc();
    let cam = rtc_camera(camera);
    let n = camera.hsize * camera.vsize;
    let mut rgb = vec![0.0f64; 3 * n];
    let n_dev = unsafe { rtc_device_count() };
    if n_dev > 1 {
        let devices: Vec<c_int> = (0..n_dev).collect();
        let mut multi: *mut RtcMulti = std::ptr::null_mut();
        let rc = unsafe { rtc_multi_create(&desc, devices.as_ptr(), n_dev, &mut multi) };
        if rc != RTC_OK {
            return Err(last_error(rc));
        }
        let rc = unsafe { rtc_render_multi(multi, &cam, fuel, rgb.as_mut_ptr(), std::ptr::null_mut()) };
        let err = if rc != RTC_OK { Some(last_error(rc)) } else { None };
        unsafe { rtc_multi_destroy(multi) };
        if let Some(e) = err {
            return Err(e);
        }
    } else {
        let mut scene: *mut RtcScene = std::ptr::null_mut();
        let rc = unsafe { rtc_scene_create(&desc, 0, &mut scene) };
        if rc != RTC_OK {
            return Err(last_error(rc));
        }
        let rc = unsafe { rtc_render(scene, &cam, fuel, std::ptr::null(), 0, n as u64, rgb.as_mut_ptr(), std::ptr::null_mut(), std::ptr::null_mut()) };
        let err = if rc != RTC_OK { Some(last_error(rc)) } else { None };
        unsafe { rtc_scene_destroy(scene) };
        if let Some(e) = err {
            return Err(e);
        }
    }
    Ok(rgb.chunks_exact(3).map(|c| Color::new(c[0], c[1], c[2])).collect())
}
