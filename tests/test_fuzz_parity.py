"""Randomised parity sweep: synthetic scenes with random primitive counts, kinds (cones on / off), grouping, lights (some inside the
cluster, some on the floor plane), CSG groups under transformed aggregation groups, smooth-triangle meshes under groups, fuels and
resolutions, both device paths against the oracle — the primary hit of every pixel and the digest of every closest hit of its ray
tree bit-exact, colours within 1e-5.  Light grids, the container passes' point test and the group gates all decide per ray which exact tests run; a scene generator
that nobody tuned the kernels on is the cheapest way to catch a wrong decision.  RTC_FUZZ_SEEDS=<n> runs more seeds on the GPU.

Six scene generators by seed range (random_case): < 3000 synthetic clouds with lights inside, CSG and meshes; >= 3000 + glass meshes,
shadowless primitives, procedural patterns, cameras inside the cloud; >= 5000 scenes from scratch (deep groups, five decades of scale);
>= 20000 CSG trees; >= 40000 LATTICE scenes (exact transforms, integer geometry: ties and thresholds everywhere); >= 60000 ordinary
geometry under extreme materials / lights / cameras.  Three ray generators beside the cameras: cases.edge_rays, cases.special_rays (aimed
at the primitives' apexes, rims, corners, poles, vertices), lattice_rays.  Rays and frames on which the reference PANICS are part of the
contract: the device must refuse them (tests/parity.py assert_ray_parity_with_panics).  DESIGN.md section 2 lists what each one found."""
import dataclasses
import math
import os

import numpy as np
import pytest

from parity import assert_parity, oracle_reference
from raytracer_challenge_amd import scenes
from raytracer_challenge_amd.scene import Camera, Color, Element, GroupKind, Material, Matrix, Noise, Pattern, PointLight, ShapeArgs, Vector, World

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _random_csg(rng):
    """A CSG group (random kind, children: two random primitives, one of them sometimes an aggregation group or a nested CSG) under a
    transformed aggregation group — the gate chain above an OP_CSG and the filter below it."""
    def prim():
        t = Matrix.translation(*rng.uniform(-0.6, 0.6, 3)) * Matrix.rotation_y(float(rng.uniform(0, 6.28))) * Matrix.scaling(*([float(rng.uniform(0.6, 1.2))] * 3))
        glass = rng.random() < 0.3
        mat = Material(pattern=Pattern.plain(Color.new(*rng.uniform(0.1, 0.9, 3))), reflective=float(rng.choice([0.0, 0.3])),
                       transparency=0.8 if glass else 0.0, refractive_index=1.4 if glass else 1.0)
        a = ShapeArgs(transform=t, material=mat)
        k = int(rng.integers(0, 4))
        return Element.sphere(a) if k == 0 else Element.cube(a) if k == 1 else Element.cylinder(a, -0.8, 0.8, True) if k == 2 else Element.cone(a, -1.0, 0.0, True)
    kinds = [GroupKind.Union, GroupKind.Intersection, GroupKind.Difference]
    left = prim()
    r = rng.random()
    if r < 0.25:
        right = Element.composite(Matrix.id(), None, GroupKind.Aggregation, [prim(), prim()])
    elif r < 0.5:
        right = Element.composite(Matrix.translation(0.2, 0, 0), None, kinds[int(rng.integers(0, 3))], [prim(), prim()])
    else:
        right = prim()
    csg = Element.composite(Matrix.rotation_x(float(rng.uniform(0, 1))), None, kinds[int(rng.integers(0, 3))], [left, right])
    where = Matrix.translation(float(rng.uniform(-8, 8)), float(rng.uniform(1.5, 6)), float(rng.uniform(-2, 10))) * Matrix.scaling(*([float(rng.uniform(1.0, 2.5))] * 3))
    return Element.composite(where, None, GroupKind.Aggregation, [csg, prim()])


def _random_mesh(rng, tmpdir):
    """A smooth-triangle mesh (a small heightfield OBJ: one group, vertex normals) under a transformed aggregation group."""
    n = int(rng.choice([6, 9, 12]))
    path = os.path.join(tmpdir, "rtc_fuzz_heightfield_%d.obj" % n)
    if not os.path.exists(path):
        scenes.write_heightfield_obj(path, n, n, 4242)
    mat = Material(pattern=Pattern.plain(Color.new(*rng.uniform(0.2, 0.9, 3))), reflective=float(rng.choice([0.0, 0.4])))
    t = Matrix.translation(float(rng.uniform(-6, 6)), float(rng.uniform(2, 5)), float(rng.uniform(0, 8))) * Matrix.rotation_z(float(rng.uniform(-0.5, 0.5))) * Matrix.scaling(*([float(rng.uniform(0.1, 0.3))] * 3))
    return Element.composite(Matrix.rotation_y(float(rng.uniform(0, 6.28))), None, GroupKind.Aggregation, [Element.obj(path, t, mat)])


def _third_wave(seed, sizes):
    """Seeds >= 5000: scenes built from scratch instead of from the BASELINE generator — aggregation groups nested up to four deep with
    their own transforms (more than 64 groups: the gate cache covers the first 64), primitive scales over five decades and clusters
    far from the origin (the accelerator's f32 boxes live in a BVH-local frame), 1-8 lights, all five analytic kinds and small meshes."""
    import tempfile
    rng = np.random.default_rng(seed)
    h, v = (int(x) for x in sizes[int(rng.integers(0, len(sizes)))])
    centre = rng.uniform(-1, 1, 3) * float(rng.choice([1.0, 1.0, 1e3, 1e5]))
    spread = float(rng.choice([0.05, 1.0, 1.0, 30.0]))

    def material():
        r = rng.random()
        col = Pattern.plain(Color.new(*rng.uniform(0.05, 1.0, 3)))
        if r < 0.5:
            return Material(pattern=col, specular=float(rng.choice([0.0, 0.3, 0.9])), shininess=float(rng.choice([5.0, 50.0, 200.0])))
        if r < 0.75:
            return Material(pattern=col, reflective=float(rng.uniform(0.1, 0.9)))
        return Material(pattern=col, diffuse=0.3, transparency=float(rng.uniform(0.3, 0.95)), reflective=float(rng.choice([0.0, 0.5, 0.9])), refractive_index=float(rng.choice([1.0, 1.1, 1.5, 2.4])))

    def prim():
        s = spread * float(10.0 ** rng.uniform(-1.5, 0.3))
        t = Matrix.translation(*(rng.uniform(-1, 1, 3) * spread * 6)) * Matrix.rotation_y(float(rng.uniform(0, 6.28))) * Matrix.rotation_x(float(rng.uniform(0, 6.28))) * \
            Matrix.scaling(s * float(rng.uniform(0.5, 2.0)), s * float(rng.uniform(0.5, 2.0)), s * float(rng.uniform(0.5, 2.0)))
        a = ShapeArgs(transform=t, material=material(), casts_shadow=bool(rng.random() > 0.05))
        k = int(rng.integers(0, 6))
        if k == 0:
            return Element.sphere(a)
        if k == 1:
            return Element.cube(a)
        if k == 2:
            return Element.cylinder(a, float(rng.uniform(-1.5, -0.2)), float(rng.uniform(0.2, 1.5)), bool(rng.integers(0, 2)))
        if k == 3:
            return Element.cone(a, float(rng.uniform(-1.2, -0.1)), float(rng.uniform(0.0, 1.0)), bool(rng.integers(0, 2)))
        if k == 4:
            p = rng.uniform(-1, 1, (3, 3))
            return Element.triangle(a, Vector.point(*p[0]), Vector.point(*p[1]), Vector.point(*p[2]))
        n = rng.normal(size=(3, 3))
        p = rng.uniform(-1, 1, (3, 3))
        return Element.smooth_triangle(a, Vector.point(*p[0]), Vector.point(*p[1]), Vector.point(*p[2]), Vector.vector(*n[0]), Vector.vector(*n[1]), Vector.vector(*n[2]))

    n_groups = [0]

    def group(depth):
        kids = []
        for _ in range(int(rng.integers(1, 5))):
            if depth < 4 and rng.random() < 0.45 and n_groups[0] < 120:
                kids.append(group(depth + 1))
            else:
                kids.append(prim())
        n_groups[0] += 1
        t = Matrix.translation(*(rng.uniform(-1, 1, 3) * spread * 2)) * Matrix.rotation_z(float(rng.uniform(-0.6, 0.6)))
        mat = material() if rng.random() < 0.2 else None
        return Element.composite(t, mat, GroupKind.Aggregation, kids)

    body = [group(0) if rng.random() < 0.7 else prim() for _ in range(int(rng.integers(3, 40)))]
    world_t = Matrix.translation(*centre)
    els = [Element.composite(world_t, None, GroupKind.Aggregation, body)]
    if rng.random() < 0.7:
        els.append(Element.plane(ShapeArgs(transform=Matrix.translation(float(centre[0]), float(centre[1] - spread * 8), float(centre[2])), material=Material(pattern=Pattern.plain(Color.new(0.5, 0.5, 0.5)), reflective=float(rng.choice([0.0, 0.4])), specular=0.0))))
    if rng.random() < 0.3:
        n = int(rng.choice([6, 9]))
        path = os.path.join(tempfile.gettempdir(), "rtc_fuzz_heightfield_%d.obj" % n)
        if not os.path.exists(path):
            scenes.write_heightfield_obj(path, n, n, 4242)
        els.append(Element.obj(path, world_t * Matrix.scaling(*([spread * 0.2] * 3)), material()))
    n_lights = int(rng.choice([1, 1, 2, 2, 3, 8]))
    lights = [PointLight(Color.new(*rng.uniform(0.2, 0.9, 3)), Vector.point(*(centre + rng.uniform(-1, 1, 3) * spread * float(rng.choice([3.0, 15.0, 60.0]))))) for _ in range(n_lights)]
    from raytracer_challenge_amd.scene import World
    world = World(lights, els)
    frm = centre + rng.uniform(-1, 1, 3) * spread * 14 + np.array([0.0, spread * 4, 0.0])
    cam = Camera.new(h, v, float(rng.uniform(0.5, 1.6)), Camera.transform(Vector.point(*frm), Vector.point(*centre), Vector.vector(0, 1, 0)))
    fuel = int(rng.choice([0, 2, 4, 6])) if n_lights <= 2 else (int(rng.choice([0, 2, 3])) if n_lights == 3 else int(rng.choice([0, 1])))
    label = "fuzz seed %d (third wave: %d groups, centre %.0e, spread %g, lights=%d fuel=%d %dx%d)" % (seed, n_groups[0], float(np.abs(centre).max()), spread, n_lights, fuel, h, v)
    return cam, world, fuel, label


def _fourth_wave(seed, sizes):
    """Seeds >= 20000: CSG-heavy scenes — Union / Intersection / Difference trees of random shape up to six deep (src/shape.rs:230-269), glass and
    mirror members, aggregation groups as CSG children and CSG groups under transformed aggregation groups, sometimes more than 32 possible
    intersections per subtree (the device's slab path), next to a few plain primitives and a floor."""
    from raytracer_challenge_amd.scene import World
    rng = np.random.default_rng(seed)
    h, v = (int(x) for x in sizes[int(rng.integers(0, len(sizes)))])
    kinds = [GroupKind.Union, GroupKind.Intersection, GroupKind.Difference]

    def material():
        r = rng.random()
        col = Pattern.plain(Color.new(*rng.uniform(0.1, 1.0, 3)))
        if r < 0.55:
            return Material(pattern=col)
        if r < 0.75:
            return Material(pattern=col, reflective=float(rng.uniform(0.2, 0.8)))
        return Material(pattern=col, diffuse=0.3, transparency=float(rng.uniform(0.4, 0.95)), reflective=float(rng.choice([0.0, 0.6])), refractive_index=float(rng.choice([1.0, 1.3, 1.5])))

    def prim(scale=1.0):
        t = Matrix.translation(*(rng.uniform(-0.7, 0.7, 3) * scale)) * Matrix.rotation_y(float(rng.uniform(0, 6.28))) * Matrix.rotation_z(float(rng.uniform(0, 6.28))) * \
            Matrix.scaling(*(rng.uniform(0.5, 1.3, 3) * scale))
        a = ShapeArgs(transform=t, material=material(), casts_shadow=bool(rng.random() > 0.1))
        k = int(rng.integers(0, 4))
        return Element.sphere(a) if k == 0 else Element.cube(a) if k == 1 else Element.cylinder(a, -0.9, 0.9, bool(rng.integers(0, 2))) if k == 2 else Element.cone(a, -1.0, float(rng.choice([0.0, 0.6])), True)

    def csg(depth):
        def child():
            r = rng.random()
            if depth < 6 and r < 0.4:
                return csg(depth + 1)
            if r < 0.55:
                return Element.composite(Matrix.rotation_x(float(rng.uniform(-0.5, 0.5))), None, GroupKind.Aggregation, [prim() for _ in range(int(rng.integers(1, 7)))])
            return prim()
        return Element.composite(Matrix.translation(*rng.uniform(-0.3, 0.3, 3)), material() if rng.random() < 0.15 else None, kinds[int(rng.integers(0, 3))], [child(), child()])

    els = []
    for _ in range(int(rng.integers(1, 6))):
        where = Matrix.translation(float(rng.uniform(-6, 6)), float(rng.uniform(0.5, 5)), float(rng.uniform(-3, 8))) * Matrix.scaling(*([float(rng.uniform(0.8, 2.2))] * 3))
        node = csg(0)
        els.append(Element.composite(where, None, GroupKind.Aggregation, [node, prim(0.5)]) if rng.random() < 0.5 else Element.composite(where, None, kinds[int(rng.integers(0, 3))], [node, prim()]))
    for _ in range(int(rng.integers(0, 12))):
        a = prim()
        els.append(Element.composite(Matrix.translation(float(rng.uniform(-8, 8)), float(rng.uniform(0.5, 6)), float(rng.uniform(-4, 10))), None, GroupKind.Aggregation, [a]))
    els.append(Element.plane(ShapeArgs(transform=Matrix.translation(0, -1.5, 0), material=Material(pattern=Pattern.checkers(Matrix.id(), Pattern.plain(Color.white()), Pattern.plain(Color.new(0.3, 0.3, 0.3))), reflective=float(rng.choice([0.0, 0.3]))))))
    n_lights = int(rng.choice([1, 2, 2, 3]))
    lights = [PointLight(Color.new(*rng.uniform(0.3, 0.9, 3)), Vector.point(float(rng.uniform(-10, 10)), float(rng.uniform(2, 12)), float(rng.uniform(-12, 4)))) for _ in range(n_lights)]
    cam = Camera.new(h, v, float(rng.uniform(0.8, 1.3)), Camera.transform(Vector.point(float(rng.uniform(-4, 4)), float(rng.uniform(1, 7)), float(rng.uniform(-14, -7))), Vector.point(0, 1.5, 2), Vector.vector(0, 1, 0)))
    fuel = int(rng.choice([0, 2, 4, 5])) if n_lights <= 2 else int(rng.choice([0, 2, 3]))
    return cam, World(lights, els), fuel, "fuzz seed %d (fourth wave: CSG, %d top elements, lights=%d fuel=%d %dx%d)" % (seed, len(els), n_lights, fuel, h, v)


def _fifth_wave(seed, sizes):
    """Seeds >= 40000: LATTICE scenes.  Every transform is translation(integers or halves) x signed axis permutation x scaling(powers of
    two), every limit and vertex an integer or a half, lights on lattice points: products, inverses and most intersections are EXACT, so
    the ties and thresholds random scenes never meet are met all the time -- tangent rays (discriminant exactly 0), rays through cube
    edges and corners (t_min == t_max), along faces, through triangle vertices and edges (u = 0, u + v = 1), a cone's apex, cap rims
    (x^2 + z^2 == r^2), y == limit, coincident faces of neighbours and of CSG operands (equal t: the sort's stability decides), lights
    on surfaces.  To be used with lattice_rays()."""
    rng = np.random.default_rng(seed)
    h, v = (int(x) for x in sizes[int(rng.integers(0, len(sizes)))])
    kinds = [GroupKind.Union, GroupKind.Intersection, GroupKind.Difference]

    def lattice(n=3, lo=-4, hi=4):
        return [float(x) for x in (rng.integers(2 * lo, 2 * hi + 1, n) / 2.0 if rng.random() < 0.3 else rng.integers(lo, hi + 1, n))]

    def transform(spread=4):
        perm = rng.permutation(3)
        P = [[0.0] * 4 for _ in range(4)]
        for r in range(3):
            P[r][int(perm[r])] = float(rng.choice([-1.0, 1.0]))
        P[3][3] = 1.0
        sc = [float(rng.choice([0.5, 1.0, 1.0, 2.0]))] * 3 if rng.random() < 0.6 else [float(x) for x in rng.choice([0.5, 1.0, 2.0], 3)]
        return Matrix.translation(*lattice(3, -spread, spread)) * Matrix(P) * Matrix.scaling(*sc)

    def pattern():
        r = rng.random()
        a, b = Pattern.plain(Color.new(*rng.uniform(0.1, 1.0, 3))), Pattern.plain(Color.new(*rng.uniform(0.0, 0.9, 3)))
        if r < 0.6:
            return a
        t = Matrix.scaling(*[float(rng.choice([0.5, 1.0, 2.0]))] * 3)
        base = [Pattern.checkers, Pattern.stripes, Pattern.ring, Pattern.gradient, Pattern.ring_gradient, Pattern.blend][int(rng.integers(0, 6))](t, a, b)
        if r < 0.85:
            return base
        # Simplex / Fractal noise evaluated at lattice points (the skew's cell corners), jittering the point or the colour
        noise = Noise.Simplex(float(rng.choice([0.25, 0.5, 1.0]))) if rng.random() < 0.5 else Noise.Fractal(float(rng.choice([0.25, 0.5])), int(rng.integers(1, 4)))
        return Pattern.point_jitter(noise, base) if rng.random() < 0.5 else Pattern.color_jitter(noise, base)

    def material():
        r = rng.random()
        refl = float(rng.choice([0.0, 0.0, 0.5, 1.0]))
        tr = float(rng.choice([0.0, 0.0, 0.5, 1.0]))
        return Material(pattern=pattern(), ambient=float(rng.choice([0.0, 0.1, 0.5])), diffuse=float(rng.choice([0.0, 0.5, 0.9])), specular=float(rng.choice([0.0, 0.5, 0.9])),
                        shininess=float(rng.choice([1.0, 10.0, 200.0])), reflective=refl, transparency=tr, refractive_index=float(rng.choice([1.0, 1.0, 1.5, 2.0])))

    def prim(spread=4):
        a = ShapeArgs(transform=transform(spread), material=material(), casts_shadow=bool(rng.random() > 0.15))
        k = int(rng.integers(0, 6))
        if k == 0:
            return Element.sphere(a)
        if k == 1:
            return Element.cube(a)
        if k in (2, 3):
            lo_, hi_ = sorted(float(x) for x in rng.choice([-2.0, -1.0, -0.5, 0.0, 0.5, 1.0, 2.0], 2, replace=False))
            if rng.random() < 0.15:
                lo_ = -math.inf
            if rng.random() < 0.15:
                hi_ = math.inf
            return (Element.cylinder if k == 2 else Element.cone)(a, lo_, hi_, bool(rng.integers(0, 2)))
        if k == 4:
            while True:
                p = [Vector.point(*lattice(3, -2, 2)) for _ in range(3)]
                e1, e2 = np.array(p[1][:3]) - np.array(p[0][:3]), np.array(p[2][:3]) - np.array(p[0][:3])
                if np.linalg.norm(np.cross(e1, e2)) > 0:
                    return Element.triangle(a, *p)
        return Element.plane(a)

    def csg(depth):
        def child():
            return csg(depth + 1) if depth < 2 and rng.random() < 0.3 else prim(1)
        return Element.composite(transform(3), material() if rng.random() < 0.2 else None, kinds[int(rng.integers(0, 3))], [child(), child()])

    def lattice_mesh():
        """An OBJ grid mesh with integer x, z and half-integer heights (quads, fan-triangulated by the parser; with or without vertex
        normals; sometimes two named groups): lattice rays pass exactly through its shared edges and through vertices six triangles
        share -- equal t on neighbouring triangles, the lower sequence number wins."""
        import tempfile
        n, smooth, two = int(rng.integers(2, 6)), bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        path = os.path.join(tempfile.gettempdir(), "rtc_fuzz_lattice_%d_%d.obj" % (os.getpid(), seed))
        hs = rng.choice([0.0, 0.0, 0.5, 1.0], (n + 1, n + 1))
        with open(path, "w") as f:
            for i in range(n + 1):
                for j in range(n + 1):
                    f.write("v %g %g %g\n" % (i - n // 2, hs[i, j], j - n // 2))
            if smooth:
                for i in range(n + 1):
                    for j in range(n + 1):
                        f.write("vn %g %g %g\n" % (float(rng.integers(-1, 2)), 1.0, float(rng.integers(-1, 2))))
            for i in range(n):
                if two and i == n // 2:
                    f.write("g second\n")
                for j in range(n):
                    a, b, c, d = i * (n + 1) + j + 1, (i + 1) * (n + 1) + j + 1, (i + 1) * (n + 1) + j + 2, i * (n + 1) + j + 2
                    f.write(("f %d//%d %d//%d %d//%d %d//%d\n" % (a, a, b, b, c, c, d, d)) if smooth else ("f %d %d %d %d\n" % (a, b, c, d)))
        return Element.obj(path, transform(3), material())

    els = []
    for _ in range(int(rng.integers(3, 25))):
        r = rng.random()
        if r < 0.08:
            m = lattice_mesh()
            els.append(m if rng.random() < 0.5 else Element.composite(transform(2), None, GroupKind.Aggregation, [m, prim(2)]))
        elif r < 0.6:
            els.append(prim())
        elif r < 0.8:
            els.append(Element.composite(transform(3), material() if rng.random() < 0.3 else None, GroupKind.Aggregation, [prim(2) for _ in range(int(rng.integers(1, 6)))]))
        else:
            els.append(csg(0))
    n_lights = int(rng.choice([0, 1, 1, 2, 3]))
    lights = [PointLight(Color.new(*rng.choice([0.0, 0.5, 1.0], 3)), Vector.point(*lattice(3, -6, 6))) for _ in range(n_lights)]
    cam = Camera.new(h, v, 1.0, Camera.transform(Vector.point(0, 2, -12), Vector.point(0, 0, 0), Vector.vector(0, 1, 0)))
    fuel = int(rng.choice([0, 1, 2, 3]))
    return cam, World(lights, els), fuel, "fuzz seed %d (fifth wave: lattice, %d top elements, lights=%d fuel=%d)" % (seed, len(els), n_lights, fuel)


def lattice_rays(n, seed):
    """Rays for _fifth_wave scenes: origins on the half-integer lattice of [-6, 6]^3, directions small integer vectors (exact tangents,
    edges, corners, apexes), every second one normalised."""
    rng = np.random.default_rng(seed + 77)
    o = rng.integers(-12, 13, (n, 3)) / 2.0
    o[: n // 2] = np.round(o[: n // 2])
    d = rng.integers(-2, 3, (n, 3)).astype(float)
    axis = rng.random(n) < 0.3
    d[axis] = np.eye(3)[rng.integers(0, 3, int(axis.sum()))] * rng.choice([-1.0, 1.0], (int(axis.sum()), 1))
    zero = ~d.any(axis=1)
    d[zero] = [0.0, 0.0, 1.0]
    norm = rng.random(n) < 0.5
    d[norm] /= np.linalg.norm(d[norm], axis=1, keepdims=True)
    return np.concatenate([o, d], axis=1)


def _sixth_wave(seed, sizes):
    """Seeds >= 60000: ORDINARY geometry, EXTREME parameters -- materials with shininess 0 / negative / 1e6 / fractional, specular and
    ambient above 1, negative diffuse, reflective and transparency above 1, refractive indices below 1, lights with negative, huge or
    zero intensity, a light at the camera; cameras of 1 x 1, 1 x N, N x 1 and other sizes below and off the 8 x 8 tile, fields of view
    from 0.01 to just under pi; fuels up to the device's 16.  `sizes` is ignored."""
    rng = np.random.default_rng(seed)
    n = int(rng.choice([6, 17, 40]))
    cam0, world = scenes.synthetic_analytic(n_primitives=n, seed=int(rng.integers(1, 1 << 30)), cones=bool(rng.integers(0, 2)), grouped=bool(rng.integers(0, 2)), hsize=32, vsize=18)

    def extreme(m):
        pick = lambda xs: float(xs[int(rng.integers(0, len(xs)))])
        return dataclasses.replace(
            m, ambient=pick([0.0, 0.1, 1.5, -0.2]), diffuse=pick([0.0, 0.9, 2.0, -0.5]), specular=pick([0.0, 0.9, 3.0, -1.0]),
            shininess=pick([0.0, 0.5, 1.0, 200.0, 1e6, 2e6, -1.0, -200.0, 7.3]), reflective=pick([0.0, 0.0, 0.5, 1.0, 1.7]),
            transparency=pick([0.0, 0.0, 0.5, 1.0, 1.3]), refractive_index=pick([1.0, 1.5, 0.5, 0.9, 3.0]))

    for e in _shapes(world.elements):
        if rng.random() < 0.7:
            object.__setattr__(e, "args", dataclasses.replace(e.args, material=extreme(e.args.material)))
    lights = []
    for _ in range(int(rng.choice([1, 1, 2, 3]))):
        inten = [float(x) for x in rng.choice([0.0, 0.3, 1.0, 5.0, -0.5, 1e6], 3)]
        lights.append(PointLight(Color.new(*inten), Vector.point(float(rng.uniform(-15, 15)), float(rng.uniform(1, 20)), float(rng.uniform(-15, 10)))))
    frm = Vector.point(float(rng.uniform(-6, 6)), float(rng.uniform(1, 8)), float(rng.uniform(-14, -6)))
    if rng.random() < 0.3:
        lights.append(PointLight(Color.new(0.5, 0.5, 0.5), frm))   # a light exactly at the camera
    world = World(lights, world.elements)
    h, v = [(1, 1), (1, 9), (9, 1), (7, 3), (13, 5), (8, 8), (17, 9), (33, 20), (64, 2)][int(rng.integers(0, 9))]
    fov = float(rng.choice([0.01, 0.5, 1.2, 3.0, 3.14, 3.1415]))
    cam = Camera.new(h, v, fov, Camera.transform(frm, Vector.point(0, 1, 2), Vector.vector(0, 1, 0)))
    fuel = int(rng.choice([0, 1, 5, 9, 16])) if len(lights) == 1 else int(rng.choice([0, 1, 3, 5]))
    return cam, world, fuel, "fuzz seed %d (sixth wave: extreme parameters, n=%d lights=%d fuel=%d %dx%d fov %g)" % (seed, n, len(lights), fuel, h, v, fov)


def random_case(seed, sizes=((64, 36), (96, 54), (128, 72)), counts=(17, 40, 96, 200, 512)):
    if seed >= 60000:
        return _sixth_wave(seed, sizes)
    if seed >= 40000:
        return _fifth_wave(seed, sizes)
    if seed >= 20000:
        return _fourth_wave(seed, sizes)
    if seed >= 5000:
        return _third_wave(seed, sizes)
    rng = np.random.default_rng(seed)
    n = int(rng.choice(counts))
    cones, grouped = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    h, v = (int(x) for x in sizes[int(rng.integers(0, len(sizes)))])
    cam, world = scenes.synthetic_analytic(n_primitives=n, seed=int(rng.integers(1, 1 << 30)), cones=cones, grouped=grouped, hsize=h, vsize=v)
    for _ in range(int(rng.integers(0, 3))):   # lights inside the cloud of primitives
        p = rng.uniform(-20, 20, 3) * np.array([1.0, 0.5, 1.0]) + np.array([0.0, 11.0, 5.0])
        world.lights.append(PointLight(Color.new(*rng.uniform(0.1, 0.6, 3)), Vector.point(*p)))
    if rng.random() < 0.3:                      # a light ON the floor plane
        world.lights.append(PointLight(Color.new(0.2, 0.2, 0.2), Vector.point(float(rng.uniform(-10, 10)), 0.0, float(rng.uniform(-10, 10)))))
    import tempfile
    n_csg = int(rng.integers(0, 3)) if rng.random() < 0.5 else 0
    n_mesh = 1 if rng.random() < 0.35 else 0
    for _ in range(n_csg):
        world.elements.append(_random_csg(rng))
    for _ in range(n_mesh):
        world.elements.append(_random_mesh(rng, tempfile.gettempdir()))
    fuel = int(rng.choice([0, 1, 3, 5, 7]))
    if len(world.lights) > 3:
        fuel = min(fuel, 5)   # (the oracle re-traces every subtree once per light: lights ** depth)
    extra = _second_wave(rng, world, cam, tempfile.gettempdir()) if seed >= 3000 else ""
    if isinstance(extra, tuple):
        cam, extra = extra
    label = "fuzz seed %d (n=%d cones=%s grouped=%s csg=%d mesh=%d lights=%d fuel=%d %dx%d%s)" % (seed, n, cones, grouped, n_csg, n_mesh, len(world.lights), fuel, h, v, extra)
    return cam, world, fuel, label


def _shapes(elements):
    for e in elements:
        if e.tag == "shape":
            yield e
        elif e.tag == "composite":
            yield from _shapes(e.children)


def _second_wave(rng, world, cam, tmpdir):
    """Seeds >= 3000 (round 3, after seed 1058): what the first generator never varied — shadowless primitives (the closest-hit shadow
    rule, src/world.rs:43-47), a GLASS mesh (every triangle is a container of its own in the reference), procedural patterns on some
    primitives (colours only), and a camera inside the cloud of primitives (rays that start inside glass)."""
    tags = []
    shapes = list(_shapes(world.elements))
    if rng.random() < 0.4:
        for e in shapes:
            if rng.random() < 0.1:
                object.__setattr__(e, "args", dataclasses.replace(e.args, casts_shadow=False))   # (the description classes are frozen)
        tags.append("shadowless")
    if rng.random() < 0.35:
        n = int(rng.choice([6, 9]))
        path = os.path.join(tmpdir, "rtc_fuzz_heightfield_%d.obj" % n)
        if not os.path.exists(path):
            scenes.write_heightfield_obj(path, n, n, 4242)
        glass = Material(pattern=Pattern.plain(Color.new(0.05, 0.1, 0.1)), diffuse=0.3, transparency=0.85, reflective=float(rng.choice([0.0, 0.5])), refractive_index=float(rng.choice([1.0, 1.3, 1.5])))
        t = Matrix.translation(float(rng.uniform(-8, 8)), float(rng.uniform(3, 12)), float(rng.uniform(-5, 10))) * Matrix.rotation_x(float(rng.uniform(-1.5, 1.5))) * Matrix.scaling(*([float(rng.uniform(0.15, 0.4))] * 3))
        world.elements.append(Element.obj(path, t, glass))
        tags.append("glass-mesh")
    if rng.random() < 0.5:
        W, K = Pattern.plain(Color.white()), Pattern.plain(Color.new(0.1, 0.2, 0.5))
        pats = [Pattern.stripes(Matrix.scaling(0.3, 0.3, 0.3), W, K), Pattern.checkers(Matrix.rotation_y(0.4), W, K), Pattern.gradient(Matrix.scaling(2, 1, 1), W, K),
                Pattern.blend(Matrix.id(), Pattern.ring(Matrix.scaling(0.2, 1, 0.2), W, K), Pattern.stripes(Matrix.rotation_z(0.5), K, W)),
                Pattern.point_jitter(Noise.Simplex(0.3), Pattern.checkers(Matrix.scaling(0.4, 0.4, 0.4), W, K)), Pattern.color_jitter(Noise.Fractal(0.2, 3), Pattern.ring_gradient(Matrix.id(), W, K))]
        for e in shapes:
            if rng.random() < 0.2:
                object.__setattr__(e, "args", dataclasses.replace(e.args, material=dataclasses.replace(e.args.material, pattern=pats[int(rng.integers(0, len(pats)))])))
        tags.append("patterns")
    if rng.random() < 0.3:
        frm = Vector.point(float(rng.uniform(-15, 15)), float(rng.uniform(2, 18)), float(rng.uniform(-10, 20)))
        to = Vector.point(float(rng.uniform(-5, 5)), float(rng.uniform(0, 10)), float(rng.uniform(0, 10)))
        cam = Camera.new(cam.hsize, cam.vsize, 1.2, Camera.transform(frm, to, Vector.vector(0, 1, 0)))
        tags.append("camera-inside")
        return cam, " " + "+".join(tags)
    return (" " + "+".join(tags)) if tags else ""


@pytest.fixture(scope="module")
def emu():
    from emu_lib import emu as _emu
    return _emu()


@pytest.mark.parametrize("seed", [2001, 2002, 2003, 2004, 2006, 2007])
def test_random_scenes_in_the_emulator(emu, orc, seed, monkeypatch):
    cam, world, fuel, label = random_case(seed, sizes=((48, 27), (64, 36)), counts=(17, 40, 96))
    ref = oracle_reference(orc, world, cam, min(fuel, 3))
    for path in ("1", "4"):
        monkeypatch.setenv("RTC_KERNEL", path)
        assert_parity(emu, orc, world, cam, min(fuel, 3), label=label + " path " + path, ref=ref)


def test_container_pass_counts_triangles_of_multi_triangle_leaves(emu, orc, monkeypatch):
    """Regression (found by the hit-tree digest in a 400-seed GPU fuzz run, seed 1058): a ray inside a glass sphere starts on a mesh
    triangle it has just been reflected off; the line crosses that triangle once BEHIND the origin, so the reference's container walk
    (src/intersection.rs:70-103) ends with the triangle as the last container and n1 = n2 = 1.  The container pass used to apply its
    sphere / cube point test to mesh leaves whose triangle count happened to set the same bit, skipped the leaf, and answered n1 = 1.5:
    total internal reflection instead of a refracted ray — a pixel off by 1.5 with every primary hit still exact."""
    cam, world, fuel, label = random_case(1058)
    idx = np.arange(1317 - 32, 1317 + 32, dtype=np.uint64)
    ref = oracle_reference(orc, world, cam, fuel, idx)
    for path in ("1", "4"):
        monkeypatch.setenv("RTC_KERNEL", path)
        assert_parity(emu, orc, world, cam, fuel, idx, label=label + " path " + path, ref=ref)


def _gpu_seeds():
    """Default: 30 seeds of the first generator, 10 of the second wave (>= 3000), 10 of the third (>= 5000), 10 of the fourth (>= 20000: CSG), and 1058
    (the container-pass regression).
    RTC_FUZZ_SEEDS=<n> [RTC_FUZZ_FIRST=<seed>] runs n consecutive seeds instead (hunting runs: 400 seeds take 2.5 minutes)."""
    if "RTC_FUZZ_SEEDS" in os.environ:
        first = int(os.environ.get("RTC_FUZZ_FIRST", "1000"))
        return list(range(first, first + int(os.environ["RTC_FUZZ_SEEDS"])))
    return list(range(1000, 1030)) + list(range(3000, 3010)) + list(range(5000, 5010)) + list(range(20000, 20010)) + [1058]


@pytest.mark.gpu
@pytest.mark.parametrize("seed", _gpu_seeds())
def test_hip_random_scenes(hip, orc, seed, monkeypatch):
    cam, world, fuel, label = random_case(seed)
    ref = oracle_reference(orc, world, cam, fuel)   # one oracle pass, both device paths against it
    for path in ("1", "4"):
        monkeypatch.setenv("RTC_KERNEL", path)
        assert_parity(hip, orc, world, cam, fuel, label=label + " path " + path, ref=ref)


def _ray_seeds():
    if "RTC_FUZZ_RAY_SEEDS" in os.environ:
        first = int(os.environ.get("RTC_FUZZ_FIRST", "1000"))
        return list(range(first, first + int(os.environ["RTC_FUZZ_RAY_SEEDS"])))
    return [1000, 1005, 3001, 5003, 5007, 20002, 5125]   # 5125: a hit at a cone's apex (NaN shadow rays, one NaN t in the list)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", _ray_seeds())
def test_hip_random_scenes_edge_rays(hip, orc, seed, monkeypatch):
    """World::color_at on rays no camera makes, in the fuzz generators' scenes: exactly axis-parallel directions, components around the
    EPSILON threshold of the cube / bounding-box slab rule, origins inside shapes, grazing rays — every ray's own nearest hit bit-exact,
    colours within 1e-5, both device paths (rtc_trace_rays)."""
    import cases
    from parity import assert_ray_parity
    cam, world, fuel, label = random_case(seed, sizes=((32, 18),))
    rays = cases.edge_rays(2048, seed=seed)
    # the generators' scenes are not all around the origin: aim the ray set at what the camera looks at
    inv = np.array(cam.transform_matrix.m, dtype=float)
    eye = np.linalg.inv(inv)[:3, 3]
    fwd = -np.linalg.inv(inv)[:3, 2]
    centre = eye + fwd * (np.linalg.norm(eye) * 0.0 + 1.0) * max(1.0, float(np.linalg.norm(fwd))) * 10.0
    scale = 1.0
    if seed >= 5000 and seed < 20000:
        rng = np.random.default_rng(seed)
        rng.integers(0, 1)                       # (same first draws as _third_wave)
        c = rng.uniform(-1, 1, 3) * float(rng.choice([1.0, 1.0, 1e3, 1e5]))
        centre, scale = c, float(rng.choice([0.05, 1.0, 1.0, 30.0]))
    else:
        centre = np.array([0.0, 3.0, 3.0])
    rays = rays.copy()
    rays[:, :3] = rays[:, :3] * scale + centre
    f = min(fuel, 3)
    for path in ("1", "4"):
        monkeypatch.setenv("RTC_KERNEL", path)
        assert_ray_parity(hip, orc, world, rays, f, label=label + " edge rays, path " + path)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", _ray_seeds())
def test_hip_random_scenes_special_rays(hip, orc, seed, monkeypatch):
    """The generators' scenes under rays aimed at their primitives' special points (cases.special_rays: cone apexes, cap rims, cube
    corners / edges, poles, triangle vertices / edges, from and to the lights, along object-space axes) -- the ray set that found the
    NaN-reflectance blend and the sign of a zero t on cylinders.  Both device paths; rays the reference panics on are refused singly."""
    import cases
    from parity import assert_ray_parity_with_panics
    cam, world, fuel, label = random_case(seed, sizes=((32, 18),))
    rays = cases.special_rays(world, 1536, seed=seed)
    for path in ("1", "4"):
        monkeypatch.setenv("RTC_KERNEL", path)
        assert_ray_parity_with_panics(hip, orc, world, rays, min(fuel, 3), label=label + " special rays, path " + path)


def _lattice_seeds():
    if "RTC_FUZZ_LATTICE_SEEDS" in os.environ:
        first = int(os.environ.get("RTC_FUZZ_FIRST", "40000"))
        return list(range(first, first + int(os.environ["RTC_FUZZ_LATTICE_SEEDS"])))
    return list(range(40000, 40024))


def _lattice_check(backend, orc, seed, label_suffix=""):
    """One lattice scene (_fifth_wave): lattice rays and special-point rays through World::color_at (relative colour tolerance: every
    second direction is not a unit vector), panicking rays refused one by one; then the scene through its camera with the digest, unless
    the reference panics on that frame (then the device must refuse it too)."""
    import cases
    from parity import assert_ray_parity_with_panics, _raises
    cam, world, fuel, label = random_case(seed, sizes=((48, 27),))
    label += label_suffix
    assert_ray_parity_with_panics(backend, orc, world, lattice_rays(1536, seed), fuel, label=label + " lattice rays", rel=True, max_panics=1536)
    assert_ray_parity_with_panics(backend, orc, world, cases.special_rays(world, 768, seed=seed), fuel, label=label + " special rays", rel=True, max_panics=768)
    try:
        ref = oracle_reference(orc, world, cam, fuel)
    except Exception as ex:  # noqa: BLE001
        assert "NaN" in str(ex)
        with pytest.raises(Exception, match="NaN"):
            backend.render(backend.build_world(world), cam, fuel)
        return
    assert_parity(backend, orc, world, cam, fuel, label=label + " camera", ref=ref)


@pytest.mark.parametrize("seed", list(range(40000, 40008)))
def test_lattice_scenes_in_the_emulator(emu, orc, seed, monkeypatch):
    for path in ("1", "4"):
        monkeypatch.setenv("RTC_KERNEL", path)
        _lattice_check(emu, orc, seed, " emulated path " + path)


def test_world_without_lights_traces_no_secondary_rays(emu, orc):
    """Found by the lattice scenes (the first generator with light-less worlds): World::shade_hit calls reflected_color and
    refracted_color INSIDE its loop over the lights (src/world.rs:58-79), so a world without lights traces no secondary ray: every hit
    is black, and a mirror facing two planes -- whose reflected NaN-free rays are harmless, but whose digest would count them -- has
    the digest of its primary hits only."""
    import cases
    world = cases.SMALL_CASES["nested_glass"]()[1]
    cam = cases.SMALL_CASES["nested_glass"]()[0]
    dark = World([], world.elements)
    nw = emu.build_world(dark)
    rgb, hits = emu.render(nw, cam, 5)
    assert not rgb.any() and (hits["prim"] >= 0).any()
    assert np.array_equal(emu.render_digest(nw, cam, 5), emu.render_digest(nw, cam, 0))
    assert_parity(emu, orc, dark, cam, 5, label="nested_glass without lights")


@pytest.mark.gpu
@pytest.mark.parametrize("seed", _lattice_seeds())
def test_hip_lattice_scenes(hip, orc, seed, monkeypatch):
    for path in ("1", "4"):
        monkeypatch.setenv("RTC_KERNEL", path)
        _lattice_check(hip, orc, seed, " path " + path)


def _extreme_seeds():
    if "RTC_FUZZ_EXTREME_SEEDS" in os.environ:
        first = int(os.environ.get("RTC_FUZZ_FIRST", "60000"))
        return list(range(first, first + int(os.environ["RTC_FUZZ_EXTREME_SEEDS"])))
    return list(range(60000, 60016))


def _extreme_check(backend, orc, seed, monkeypatch, tag):
    import cases
    from parity import assert_ray_parity_with_panics
    cam, world, fuel, label = random_case(seed)
    ref = oracle_reference(orc, world, cam, fuel)
    rays = cases.special_rays(world, 512, seed=seed)
    for path in ("1", "4"):
        monkeypatch.setenv("RTC_KERNEL", path)
        assert_parity(backend, orc, world, cam, fuel, label="%s %s path %s" % (label, tag, path), ref=ref, rel=True)
        assert_ray_parity_with_panics(backend, orc, world, rays, min(fuel, 3), label="%s %s special rays path %s" % (label, tag, path), rel=True)


@pytest.mark.parametrize("seed", list(range(60000, 60006)))
def test_extreme_parameters_in_the_emulator(emu, orc, seed, monkeypatch):
    """_sixth_wave: ordinary geometry under extreme materials, lights and cameras (colours compared relative to max(1, |reference|):
    an intensity of 1e6 and a specular coefficient of 3 leave the unit range)."""
    _extreme_check(emu, orc, seed, monkeypatch, "emulated")


@pytest.mark.gpu
@pytest.mark.parametrize("seed", _extreme_seeds())
def test_hip_extreme_parameters(hip, orc, seed, monkeypatch):
    _extreme_check(hip, orc, seed, monkeypatch, "")
