"""Scene cases shared by the CPU (emulated-kernel) and GPU parity tests.  Each returns (camera, world)."""
import math
import os

import numpy as np

from raytracer_challenge_amd import scenes
from raytracer_challenge_amd.scene import (Camera, Color, Element, GroupKind, Material, Matrix, Noise, Pattern, PointLight, ShapeArgs, Vector, World)

PI = math.pi


def _cam(h=48, v=32, fov=1.0, frm=(0, 1.5, -6), to=(0, 0.5, 0)):
    return Camera.new(h, v, fov, Camera.transform(Vector.point(*frm), Vector.point(*to), Vector.vector(0, 1, 0)))


def nested_glass():
    """Three nested/overlapping glass spheres + tangent cases: exercises the n1/n2 container logic (src/intersection.rs:70-103)."""
    g = lambda ri, **kw: Material(pattern=Pattern.plain(Color.new(0.1, 0.1, 0.1)), diffuse=0.2, transparency=0.9, reflective=0.6, refractive_index=ri, **kw)
    els = [
        Element.plane(ShapeArgs(transform=Matrix.translation(0, -2, 0), material=Material(pattern=Pattern.checkers(Matrix.id(), Pattern.plain(Color.white()), Pattern.plain(Color.new(0.2, 0.2, 0.2)))))),
        Element.sphere(ShapeArgs(transform=Matrix.scaling(2, 2, 2), material=g(1.5))),
        Element.sphere(ShapeArgs(transform=Matrix.translation(0, 0, -0.25), material=g(2.0))),
        Element.sphere(ShapeArgs(transform=Matrix.translation(0, 0, 0.25), material=g(2.5))),
        Element.sphere(ShapeArgs(transform=Matrix.translation(3, 0, 0), material=g(1.0))),  # tangent to the big one
        Element.cube(ShapeArgs(transform=Matrix.translation(-3, 0, 1) * Matrix.rotation_y(0.4), material=g(1.3))),
        Element.cylinder(ShapeArgs(transform=Matrix.translation(0, -1, -3) * Matrix.scaling(0.5, 1, 0.5), material=g(1.4)), 0.0, 1.5, True),
    ]
    return _cam(64, 40, 1.1, (0.5, 1.0, -7)), World([PointLight(Color.white(), Vector.point(-10, 10, -10)), PointLight(Color.new(0.3, 0.3, 0.5), Vector.point(8, 6, -4))], els)


def glass_cluster(n=40, seed=21):
    """Forty overlapping glass spheres, cubes, closed cylinders and cones in a 4-unit box: hit points inside several shapes at once
    (container lists of depth > 2), enough bounded primitives for the analytic BVH's light grids, and every kind of BVH leaf on the
    container passes (spheres and cubes are skipped there unless the hit point is inside their box)."""
    rng = np.random.default_rng(seed)
    els = [Element.plane(ShapeArgs(transform=Matrix.translation(0, -3, 0), material=Material(pattern=Pattern.checkers(Matrix.id(), Pattern.plain(Color.white()), Pattern.plain(Color.new(0.2, 0.2, 0.2))))))]
    for i in range(n):
        t = Matrix.translation(*rng.uniform(-2, 2, 3)) * Matrix.rotation_y(rng.uniform(0, 6.28)) * Matrix.rotation_x(rng.uniform(0, 6.28)) * Matrix.scaling(*([rng.uniform(0.5, 1.5)] * 3))
        mat = Material(pattern=Pattern.plain(Color.new(*rng.uniform(0.05, 0.3, 3))), diffuse=0.3, transparency=0.85, reflective=0.3, refractive_index=float(rng.choice([1.0, 1.2, 1.5, 2.0])))
        args = ShapeArgs(transform=t, material=mat)
        k = i % 4
        els.append(Element.sphere(args) if k == 0 else Element.cube(args) if k == 1 else Element.cylinder(args, -0.7, 0.8, True) if k == 2 else Element.cone(args, -1.0, 0.5, True))
    return _cam(48, 32, 1.0, (0.5, 1.0, -9)), World([PointLight(Color.white(), Vector.point(-10, 10, -10)), PointLight(Color.new(0.4, 0.4, 0.6), Vector.point(1.0, 0.5, 0.2))], els)


def all_primitives():
    """Every Geometry variant incl. open/closed cylinders and cones, triangles, shadowless shapes, Debug pattern."""
    m = lambda r, g, b, **kw: Material(pattern=Pattern.plain(Color.new(r, g, b)), **kw)
    tri = Element.triangle(ShapeArgs(material=m(0.9, 0.2, 0.2)), Vector.point(-1, 0, 2), Vector.point(1, 0, 2), Vector.point(0, 2, 2))
    stri = Element.smooth_triangle(ShapeArgs(transform=Matrix.translation(2.5, 0, 1), material=m(0.2, 0.9, 0.2, reflective=0.3)),
                                   Vector.point(-1, 0, 0), Vector.point(1, 0, 0), Vector.point(0, 2, 0),
                                   Vector.vector(-0.5, 0, -1), Vector.vector(0.5, 0, -1), Vector.vector(0, 0.5, -1))
    els = [
        Element.plane(ShapeArgs(material=m(0.8, 0.8, 0.8, reflective=0.2))),
        Element.sphere(ShapeArgs(transform=Matrix.translation(-2.5, 1, 0), material=Material(pattern=Pattern.debug(), ambient=0.4))),
        Element.cube(ShapeArgs(transform=Matrix.translation(0, 0.5, -1) * Matrix.rotation_y(0.7) * Matrix.scaling(0.5, 0.5, 0.5), material=m(0.2, 0.3, 0.9))),
        Element.cylinder(ShapeArgs(transform=Matrix.translation(-1, 0, -2.5) * Matrix.scaling(0.4, 1, 0.4), material=m(0.9, 0.9, 0.2)), 0.0, 1.2, False),
        Element.cylinder(ShapeArgs(transform=Matrix.translation(1.2, 0, -2.5) * Matrix.scaling(0.4, 1, 0.4), material=m(0.9, 0.5, 0.2)), 0.0, 1.0, True),
        Element.cone(ShapeArgs(transform=Matrix.translation(2.5, 1.0, -1.5) * Matrix.scaling(0.6, 1, 0.6), material=m(0.6, 0.2, 0.9)), -1.0, 0.0, True),
        Element.cone(ShapeArgs(transform=Matrix.translation(-3.0, 0.8, -2.0) * Matrix.scaling(0.5, 0.8, 0.5), material=m(0.2, 0.8, 0.8)), -1.0, 1.0, False),
        Element.sphere(ShapeArgs(transform=Matrix.translation(0, 3.0, -1) * Matrix.scaling(0.5, 0.5, 0.5), material=m(1, 1, 1), casts_shadow=False)),
        tri, stri,
    ]
    return _cam(72, 48, 1.2, (0.3, 2.5, -8), (0, 0.8, 0)), World([PointLight(Color.white(), Vector.point(-6, 8, -8))], els)


def patterns_and_noise():
    """Every MixtureKind, both JitterKinds, Simplex and Fractal noise, nested pattern transforms, group materials."""
    W, K = Pattern.plain(Color.white()), Pattern.plain(Color.new(0.1, 0.1, 0.4))
    R, G = Pattern.plain(Color.new(0.9, 0.2, 0.1)), Pattern.plain(Color.new(0.1, 0.8, 0.3))
    pats = [
        Pattern.blend(Matrix.id(), Pattern.stripes(Matrix.scaling(0.3, 1, 1), W, K), Pattern.stripes(Matrix.rotation_y(PI / 2) * Matrix.scaling(0.3, 1, 1), R, G)),
        Pattern.ring_gradient(Matrix.scaling(0.5, 0.5, 0.5), W, R),
        Pattern.ring(Matrix.scaling(0.25, 1, 0.25), G, K),
        Pattern.gradient(Matrix.translation(0.3, 0, 0) * Matrix.scaling(2, 1, 1), R, K),
        Pattern.point_jitter(Noise.Fractal(0.3, 4), Pattern.stripes(Matrix.scaling(0.2, 0.2, 0.2), W, K)),
        Pattern.color_jitter(Noise.Simplex(0.2), Pattern.checkers(Matrix.scaling(0.5, 0.5, 0.5), R, G)),
        Pattern.point_jitter(Noise.Simplex(0.3), Pattern.checkers(Matrix.scaling(0.4, 0.4, 0.4), W, K)),
        Pattern.checkers(Matrix.scaling(0.7, 0.7, 0.7), Pattern.gradient(Matrix.id(), W, R), Pattern.ring(Matrix.scaling(0.2, 0.2, 0.2), K, G)),
    ]
    els = [Element.plane(ShapeArgs(material=Material(pattern=pats[7], reflective=0.1)))]
    for i, p in enumerate(pats[:7]):
        x, z = (i % 4) * 2.2 - 3.3, (i // 4) * 2.5
        els.append(Element.sphere(ShapeArgs(transform=Matrix.translation(x, 1, z) * Matrix.rotation_z(0.3 * i), material=Material(pattern=p, specular=0.3))))
    grouped = Element.composite(Matrix.translation(0, 0, 5) * Matrix.scaling(1.5, 1.5, 1.5), Material(pattern=pats[4], diffuse=0.8), GroupKind.Aggregation, [
        Element.cube(ShapeArgs(transform=Matrix.translation(-1.5, 1, 0))), Element.sphere(ShapeArgs(transform=Matrix.translation(1.5, 1, 0)))])
    els.append(grouped)
    return _cam(72, 48, 1.1, (0, 5, -9), (0, 1, 1)), World([PointLight(Color.white(), Vector.point(-5, 10, -8))], els)


def nested_groups():
    """Groups in groups with transforms, open cylinders (infinite / NaN-poisoned reference boxes), planes inside groups."""
    leg = lambda: Element.composite(Matrix.id(), None, GroupKind.Aggregation, [
        Element.sphere(ShapeArgs(transform=Matrix.translation(0, 0, -1) * Matrix.scaling(0.25, 0.25, 0.25))),
        Element.cylinder(ShapeArgs(transform=Matrix.translation(0, 0, -1) * Matrix.rotation_y(-PI / 6) * Matrix.rotation_z(-PI / 2) * Matrix.scaling(0.25, 1, 0.25)), 0.0, 1.0, False)])
    ring = Element.composite(Matrix.rotation_x(0.3), Material(pattern=Pattern.plain(Color.new(0.8, 0.6, 0.2)), reflective=0.2), GroupKind.Aggregation,
                             [Element.composite(Matrix.rotation_y(n * PI / 3), None, GroupKind.Aggregation, [leg()]) for n in range(6)])
    blob = Element.composite(Matrix.translation(2.5, 0.5, 0), None, GroupKind.Aggregation,
                             [Element.sphere(ShapeArgs(transform=Matrix.translation(0.4 * i, 0.3 * (i % 3), 0.2 * i) * Matrix.scaling(0.3, 0.3, 0.3),
                                                       material=Material(pattern=Pattern.plain(Color.new(0.2 + 0.08 * i, 0.5, 0.9 - 0.08 * i))))) for i in range(9)])
    floor_in_group = Element.composite(Matrix.translation(0, -1, 0), None, GroupKind.Aggregation, [Element.plane(ShapeArgs(material=Material(pattern=Pattern.plain(Color.new(0.6, 0.6, 0.6)))))])
    empty = Element.composite(Matrix.id(), None, GroupKind.Aggregation, [])
    return _cam(64, 40, 0.9, (3, 3, -6), (0.5, 0, 0)), World([PointLight(Color.white(), Vector.point(-4, 8, -6))], [ring, blob, floor_in_group, empty])


def cube_lattice():
    """4x4x4 axis-aligned cubes + a few cones: axis-parallel rays are 'quirk rays' (|d|<EPSILON slab rule, src/shape.rs:641)
    for every cube at once, which stresses the direction-grid culled scan (OP_QGRID)."""
    els = [Element.plane(ShapeArgs(transform=Matrix.translation(0, -4, 0)))]
    for i in range(4):
        for j in range(4):
            for k in range(4):
                t = Matrix.translation(2.0 * i - 3.0, 2.0 * j - 3.0, 2.0 * k - 3.0) * Matrix.scaling(0.6, 0.6, 0.6)
                mat = Material(pattern=Pattern.plain(Color.new(0.2 + 0.2 * i, 0.2 + 0.2 * j, 0.2 + 0.2 * k)), reflective=0.3 if (i + j + k) % 3 == 0 else 0.0,
                               transparency=0.8 if (i + j + k) % 5 == 0 else 0.0, refractive_index=1.3)
                els.append(Element.cube(ShapeArgs(transform=t, material=mat)))
    for i in range(4):
        els.append(Element.cone(ShapeArgs(transform=Matrix.translation(2.0 * i - 3.0, 4.5, 0.0) * Matrix.rotation_z(0.2 * i)), -1.0, 0.0, True))
    return _cam(64, 48, 1.0, (0.0, 0.0, -14.0), (0, 0, 0)), World([PointLight(Color.white(), Vector.point(-8, 12, -12))], els)


def synthetic_mesh_small():
    """Config-5 shaped scene at test size: 40x40 heightfield (3 042 smooth triangles, one OBJ group) with Fractal/Simplex
    point-jitter materials, a glass sphere, two lights."""
    import tempfile
    path = os.path.join(tempfile.gettempdir(), "rtc_heightfield_40x40_12345.obj")
    if not os.path.exists(path):
        scenes.write_heightfield_obj(path, 40, 40, 12345)
    return scenes.synthetic_mesh(path, hsize=48, vsize=27)


def csg_scene():
    """Union / Intersection / Difference groups (src/shape.rs:161-178, :230-269): the classic carved cube, a lens, a union
    with a glass member (containers see only retained intersections), a CSG nested in a CSG, a CSG inside a transformed
    aggregation group, and a CSG whose child is an aggregation group."""
    m = lambda r, g, b, **kw: Material(pattern=Pattern.plain(Color.new(r, g, b)), **kw)
    carved = Element.composite(Matrix.translation(-3, 1, 0) * Matrix.rotation_y(0.6), None, GroupKind.Difference, [
        Element.cube(ShapeArgs(material=m(0.9, 0.7, 0.2))),
        Element.sphere(ShapeArgs(transform=Matrix.scaling(1.3, 1.3, 1.3), material=m(0.8, 0.1, 0.1, reflective=0.3)))])
    lens = Element.composite(Matrix.translation(0, 1, 0), None, GroupKind.Intersection, [
        Element.sphere(ShapeArgs(transform=Matrix.translation(-0.5, 0, 0), material=m(0.1, 0.1, 0.2, transparency=0.9, reflective=0.5, refractive_index=1.5, diffuse=0.2))),
        Element.sphere(ShapeArgs(transform=Matrix.translation(0.5, 0, 0), material=m(0.1, 0.2, 0.1, transparency=0.9, reflective=0.5, refractive_index=1.5, diffuse=0.2)))])
    union_glass = Element.composite(Matrix.translation(3, 1, 0), None, GroupKind.Union, [
        Element.cylinder(ShapeArgs(transform=Matrix.scaling(0.6, 1, 0.6), material=m(0.2, 0.6, 0.9)), -1.0, 1.0, True),
        Element.sphere(ShapeArgs(transform=Matrix.translation(0, 0.8, 0), material=m(0.05, 0.05, 0.05, transparency=0.8, reflective=0.4, refractive_index=1.3, diffuse=0.3)))])
    inner = Element.composite(Matrix.id(), None, GroupKind.Union, [
        Element.sphere(ShapeArgs(transform=Matrix.translation(0, 0, -0.6) * Matrix.scaling(0.7, 0.7, 0.7), material=m(0.9, 0.9, 0.9))),
        Element.sphere(ShapeArgs(transform=Matrix.translation(0, 0, 0.6) * Matrix.scaling(0.7, 0.7, 0.7), material=m(0.9, 0.5, 0.9)))])
    nested = Element.composite(Matrix.translation(-1.5, 1, 3.5), None, GroupKind.Difference, [
        Element.cube(ShapeArgs(transform=Matrix.scaling(1.0, 0.8, 1.2), material=m(0.3, 0.8, 0.4))), inner])
    in_group = Element.composite(Matrix.translation(2, 0.8, 3.5) * Matrix.scaling(0.8, 0.8, 0.8), m(0.9, 0.4, 0.1, specular=0.5), GroupKind.Aggregation, [
        Element.composite(Matrix.rotation_x(0.5), None, GroupKind.Difference, [Element.cone(ShapeArgs(), -1.0, 0.0, True), Element.cube(ShapeArgs(transform=Matrix.translation(0.9, -0.5, 0)))]),
        Element.sphere(ShapeArgs(transform=Matrix.translation(0, 1.0, 0) * Matrix.scaling(0.4, 0.4, 0.4)))])
    agg_child = Element.composite(Matrix.translation(5.5, 1, 2), None, GroupKind.Intersection, [
        Element.composite(Matrix.id(), None, GroupKind.Aggregation, [
            Element.sphere(ShapeArgs(transform=Matrix.translation(-0.4, 0, 0), material=m(0.7, 0.7, 0.2))),
            Element.sphere(ShapeArgs(transform=Matrix.translation(0.4, 0, 0), material=m(0.2, 0.7, 0.7)))]),
        Element.cube(ShapeArgs(transform=Matrix.scaling(1.2, 0.5, 1.2), material=m(0.6, 0.3, 0.6)))])
    floor = Element.plane(ShapeArgs(material=Material(pattern=Pattern.checkers(Matrix.id(), Pattern.plain(Color.new(0.8, 0.8, 0.8)), Pattern.plain(Color.new(0.3, 0.3, 0.3))), reflective=0.1)))
    return _cam(96, 54, 1.0, (1.0, 5.0, -9.0), (1.0, 0.8, 1.0)), World(
        [PointLight(Color.white(), Vector.point(-6, 9, -7)), PointLight(Color.new(0.4, 0.4, 0.5), Vector.point(8, 6, -3))],
        [floor, carved, lens, union_glass, nested, in_group, agg_child])


def nested_pattern(kind, depth):
    """A pattern tree `depth` mixtures deep: checkers (no colour frame on the device's walk) or blends (one frame each)."""
    W, K = Pattern.plain(Color.white()), Pattern.plain(Color.new(0.1, 0.2, 0.6))
    p = Pattern.stripes(Matrix.scaling(0.3, 0.3, 0.3), W, K)
    for i in range(depth):
        t = Matrix.rotation_y(0.3 * i) * Matrix.scaling(0.9, 0.9, 0.9)
        p = Pattern.checkers(t, p, K) if kind == "checkers" else Pattern.blend(t, p, K)
    return p


def pattern_world(pattern):
    els = [Element.plane(ShapeArgs(material=Material(pattern=pattern))),
           Element.sphere(ShapeArgs(transform=Matrix.translation(0, 1, 0), material=Material(pattern=pattern, reflective=0.2)))]
    cam = Camera.new(48, 32, 1.0, Camera.transform(Vector.point(0, 2.0, -5), Vector.point(0, 0.8, 0), Vector.vector(0, 1, 0)))
    return cam, World([PointLight(Color.white(), Vector.point(-5, 8, -6))], els)


def csg_nested(levels):
    """Difference / Union groups nested `levels` deep around one sphere (the reference's Group::intersect recursion has no limit)."""
    m = lambda r, g, b: Material(pattern=Pattern.plain(Color.new(r, g, b)))
    node = Element.sphere(ShapeArgs(transform=Matrix.scaling(1.2, 1.2, 1.2), material=m(0.9, 0.3, 0.2)))
    for i in range(levels):
        other = Element.cube(ShapeArgs(transform=Matrix.translation(0.35 * ((i % 3) - 1), 0.3 * ((i % 2) * 2 - 1), 0.2 * (i % 4) - 0.3) * Matrix.scaling(0.5, 0.5, 0.5),
                                       material=m(0.2 + 0.05 * i, 0.8 - 0.04 * i, 0.5)))
        node = Element.composite(Matrix.rotation_y(0.2), None, GroupKind.Difference if i % 2 == 0 else GroupKind.Union, [node, other])
    floor = Element.plane(ShapeArgs(transform=Matrix.translation(0, -1.5, 0)))
    cam = Camera.new(64, 40, 0.9, Camera.transform(Vector.point(0.5, 1.5, -5), Vector.point(0, 0, 0), Vector.vector(0, 1, 0)))
    return cam, World([PointLight(Color.white(), Vector.point(-4, 6, -6))], [node, floor])


def edge_rays(n=4096, seed=7):
    """Rays for color_at parity: random, axis-parallel (the |d|<EPSILON slab rule), grazing, starting inside shapes, zero-ish components."""
    rng = np.random.default_rng(seed)
    o = rng.uniform(-6, 6, (n, 3))
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    k = n // 8
    d[:k] = np.eye(3)[rng.integers(0, 3, k)] * rng.choice([-1.0, 1.0], (k, 1))          # exactly axis-parallel
    d[k:2 * k, 0] = rng.uniform(-2e-5, 2e-5, k)                                             # around the EPSILON threshold
    d[2 * k:3 * k, 1] = rng.uniform(-2e-5, 2e-5, k)
    o[3 * k:4 * k] = rng.uniform(-0.9, 0.9, (k, 3))                                         # inside the unit shapes
    o[4 * k:5 * k, 1] = 1.0                                                                  # grazing y = 1 faces / caps
    d[4 * k:5 * k, 1] = 0.0
    return np.concatenate([o, d], axis=1)


def far_rays(n=1024, seed=11, distances=(1e3, 1e5, 1e7)):
    """Rays that start 10^3..10^7 units away and aim at random points inside the scene: the accelerator's f32 slab test works
    in a BVH-local frame with a per-ray error bound that grows with the origin's distance; hits must stay bit-exact."""
    rng = np.random.default_rng(seed)
    target = rng.uniform(-3, 3, (n, 3))
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    dist = np.array(distances)[rng.integers(0, len(distances), n)][:, None]
    o = target - d * dist
    return np.concatenate([o, d], axis=1)


SMALL_CASES = {
    "default_world": lambda: scenes.default_world(),
    "glass_air_bubble_200x100": lambda: scenes.chapter11_glass_air_bubble(200, 100),   # BASELINE config 1
    "chapter11_title": lambda: scenes.chapter11_title(96, 54),
    "chapter12_title": lambda: scenes.chapter12_title(96, 54),
    "chapter13_title": lambda: scenes.chapter13_title(96, 54),
    "chapter14_title": lambda: scenes.chapter14_title(96, 54),
    "cover": lambda: scenes.cover(64, 64),
    "chapter14_hexagon": lambda: scenes.chapter14_hexagon(96, 54),
    "chapter14_benchmark": lambda: scenes.chapter14_benchmark(96, 54),
    "teapot_low": lambda: scenes.chapter15_teapot("teapot_low.obj", 96, 54),
    "synthetic_analytic": lambda: scenes.synthetic_analytic(hsize=96, vsize=54),
    "synthetic_cones_grouped": lambda: scenes.synthetic_analytic(cones=True, grouped=True, hsize=96, vsize=54),
    "nested_glass": nested_glass,
    "all_primitives": all_primitives,
    "patterns_and_noise": patterns_and_noise,
    "nested_groups": nested_groups,
    "cube_lattice": cube_lattice,
    "synthetic_mesh_small": synthetic_mesh_small,
    "csg_scene": csg_scene,
}


def cone_apex_world(n_planes: int, glass_mirror: bool = False, planes_in_group: bool = False):
    """A ray through a double cone's apex: the local normal there is (0, 0, 0), its normalisation NaN, and so are over_point and the
    shadow rays.  A NaN ray makes no cone intersection (`disc >= 0` fails) and one NaN t per plane (`|dy| < EPSILON` fails):
    the reference's shadow list then holds `n_planes` NaNs, and its sort panics only when that is at least two (a one-element
    slice is never compared, src/intersection.rs:123-125) -- with one plane the pixel is the ambient term."""
    mat = Material(pattern=Pattern.plain(Color(0.2, 0.6, 0.3)), ambient=0.25)
    if glass_mirror:  # reflective AND transparent: the reference blends with the Schlick reflectance, NaN at the apex -> a NaN pixel
        mat = Material(pattern=Pattern.plain(Color(0.2, 0.6, 0.3)), ambient=0.25, reflective=0.5, transparency=0.5, refractive_index=1.5)
    els = [Element.cone(ShapeArgs(material=mat), -math.inf, math.inf, False)]
    planes = [Element.plane(ShapeArgs(transform=Matrix.translation(0.0, -3.0 - i, 0.0))) for i in range(n_planes)]
    # planes_in_group: Group::intersect's box test fails for a NaN ray (src/shape.rs:251, bounding_box.rs:80-92), so the planes are
    # never asked and the shadow list is empty
    els += [Element.composite(Matrix.id(), None, GroupKind.Aggregation, planes)] if planes_in_group else planes
    world = World(elements=els, lights=[PointLight(Color(1.0, 1.0, 1.0), Vector.point(-4.0, 6.0, -7.0)), PointLight(Color(0.5, 0.5, 0.5), Vector.point(3.0, 5.0, -2.0))])
    rays = np.array([[0.0, 0.0, -5.0, 0.0, 0.0, 1.0],      # the apex: t = 5 twice, object point (0, 0, 0)
                     [0.5, 0.25, -5.0, 0.0, 0.0, 1.0],     # an ordinary cone hit
                     [0.0, 4.0, -5.0, 0.0, -0.2, 1.0]])    # misses the cone, hits a plane
    return world, rays


def _special_points(world):
    """World-space points where the primitives' formulas change branch or lose a derivative: a cone's apex, rims of cylinder / cone caps,
    cube corners / edge midpoints / face centres, sphere poles, triangle vertices / edge midpoints, points of planes.  (shape, point)
    pairs plus each primitive's object->world matrix."""
    pts, mats = [], []

    def add(M, local):
        for p in local:
            w = M @ np.array([p[0], p[1], p[2], 1.0])
            if np.all(np.isfinite(w)):
                pts.append(w[:3])

    def walk(el, M):
        if el.tag == "composite":
            Mc = M @ np.array(el.transform.m, dtype=float)
            for c in el.children:
                walk(c, Mc)
            return
        if el.tag == "obj":   # vertices of the file and midpoints of consecutive ones (edges of its quads / strips), at most 64
            Mo = M @ np.array(el.transform.m, dtype=float)
            try:
                vs = [tuple(float(x) for x in ln.split()[1:4]) for ln in open(el.path) if ln.startswith("v ")][:64]
            except OSError:
                vs = []
            mats.append(Mo)
            add(Mo, vs + [tuple((np.array(a) + np.array(b)) / 2) for a, b in zip(vs, vs[1:])])
            return
        if el.tag != "shape":
            return
        Ms = M @ np.array(el.args.transform.m, dtype=float)
        mats.append(Ms)
        g = el.geometry
        if g == "sphere":
            add(Ms, [(0, 1, 0), (0, -1, 0), (1, 0, 0), (-1, 0, 0), (0, 0, 1), (0, 0, -1), (0, 0, 0)])
        elif g == "cube":
            add(Ms, [(x, y, z) for x in (-1, 0, 1) for y in (-1, 0, 1) for z in (-1, 0, 1)])
        elif g in ("cylinder", "cone"):
            lo, hi = el.params[0], el.params[1]
            ys = [y for y in (lo, hi, 0.0) if math.isfinite(y)]
            for y in ys:
                r = 1.0 if g == "cylinder" else abs(y)
                add(Ms, [(r, y, 0), (-r, y, 0), (0, y, r), (0, y, -r), (r * math.sqrt(0.5), y, r * math.sqrt(0.5)), (0, y, 0)])
        elif g in ("triangle", "smooth_triangle"):
            p = np.array(el.params[:9]).reshape(3, 3)
            add(Ms, [p[0], p[1], p[2], (p[0] + p[1]) / 2, (p[1] + p[2]) / 2, (p[0] + p[2]) / 2, (p[0] + p[1] + p[2]) / 3])
        elif g == "plane":
            add(Ms, [(0, 0, 0), (1, 0, 0), (0, 0, 1), (-3, 0, 2)])

    for el in world.elements:
        walk(el, np.eye(4))
    return np.array(pts).reshape(-1, 3), mats


def special_rays(world, n=2048, seed=11):
    """Rays for color_at parity aimed at `_special_points`: from a random origin through one; from one to another (what shadow and
    reflection rays between primitives look like); starting exactly on one; through one along an axis of its primitive's object space
    (exactly axis-parallel object rays where the transform is a translation / power-of-two scaling); towards the lights."""
    rng = np.random.default_rng(seed)
    pts, mats = _special_points(world)
    if len(pts) == 0:
        return edge_rays(n, seed)
    lo, hi = pts.min(axis=0), pts.max(axis=0)
    span = np.maximum(hi - lo, 1.0)
    rays = np.zeros((n, 6))
    lights = np.array([list(l.origin[:3]) for l in world.lights]) if world.lights else np.zeros((1, 3))
    for i in range(n):
        kind = i % 6
        tgt = pts[rng.integers(len(pts))]
        if kind == 0:      # random origin in the cloud's box (inflated) through the point
            o = lo - span + rng.random(3) * 3 * span
            d = tgt - o
        elif kind == 1:    # point to point
            o = pts[rng.integers(len(pts))]
            d = tgt - o
        elif kind == 2:    # starts exactly on the point, random direction
            o, d = tgt, rng.normal(size=3)
        elif kind == 3:    # through the point along an object-space axis of some primitive
            M = mats[rng.integers(len(mats))]
            a = M[:3, rng.integers(3)] * float(rng.choice([-1.0, 1.0]))
            d = a
            o = tgt - a * float(rng.choice([1.0, 2.0, 4.0, 0.5]))
        elif kind == 4:    # from the point towards a light / from a light to the point
            L = lights[rng.integers(len(lights))]
            o, d = (tgt, L - tgt) if rng.random() < 0.5 else (L, tgt - L)
        else:              # starts on the point, heads along an object-space axis (grazes faces / caps / the plane)
            M = mats[rng.integers(len(mats))]
            o, d = tgt, M[:3, rng.integers(3)] * float(rng.choice([-1.0, 1.0]))
        ln = float(np.linalg.norm(d))
        if not (ln > 1e-300 and math.isfinite(ln)):
            d, ln = np.array([0.0, 0.0, 1.0]), 1.0
        rays[i, :3], rays[i, 3:] = o, d / ln   # unit directions (an exactly axis-parallel one stays exactly so)
    return rays
