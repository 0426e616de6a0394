"""Scene programs: the reference's `construct_world()` bodies restated as data, plus the synthetic
benchmark scenes of BASELINE.json / SURVEY.md §8(d).

Each function returns ``(Camera, World)`` built from :mod:`raytracer_challenge_amd.scene` values only —
no native state — so the same scene can be handed to any backend.  Cameras default to the reference's
native resolution; pass ``hsize``/``vsize`` to re-target (the BASELINE configs use 1920x1080 / 3840x2160).
"""
from __future__ import annotations

import math
import os
from typing import Tuple

from .scene import (Camera, Color, Element, GroupKind, Material, Matrix, Noise, Pattern, PointLight, ShapeArgs, Vector, World)

PI = math.pi
ASSETS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "assets")


def _cam(h, v, fov, frm, to, up, hsize=None, vsize=None) -> Camera:
    return Camera.new(hsize or h, vsize or v, fov, Camera.transform(Vector.point(*frm), Vector.point(*to), Vector.vector(*up)))


def default_world(hsize=11, vsize=11) -> Tuple[Camera, World]:
    """World::default (src/world.rs:152-183) with the camera of src/image.rs:128-146."""
    return _cam(11, 11, PI / 2.0, (0, 0, -5), (0, 0, 0), (0, 1, 0), hsize, vsize), World.default()


def chapter11_glass_air_bubble(hsize=None, vsize=None) -> Tuple[Camera, World]:
    """src/bin/chapter11_glass_air_bubble.rs:15-77 (BASELINE config 1 renders it at 200x100)."""
    floor = Element.plane(ShapeArgs(
        transform=Matrix.translation(0.0, -10.0, 0.0),
        material=Material(pattern=Pattern.checkers(Matrix.id(), Pattern.plain(Color.white()), Pattern.plain(Color.black())), specular=0.0)))
    glass = Element.sphere(ShapeArgs(material=Material(
        pattern=Pattern.plain(Color.black()), diffuse=0.1, shininess=300.0, reflective=1.0, transparency=1.0, refractive_index=1.52)))
    air = Element.sphere(ShapeArgs(transform=Matrix.scaling(0.5, 0.5, 0.5), material=Material(
        pattern=Pattern.plain(Color.black()), diffuse=0.1, shininess=300.0, reflective=1.0, transparency=1.0, refractive_index=1.0)))
    world = World([PointLight(Color.new(0.7, 0.7, 0.7), Vector.point(20.0, 10.0, 0.0))], [floor, glass, air])
    return _cam(4096, 4096, PI / 3.0, (0.0, 2.5, 0.0), (0.0, 0.0, 0.0), (0.0, 0.0, 1.0), hsize, vsize), world


def chapter11_title(hsize=None, vsize=None) -> Tuple[Camera, World]:
    """src/bin/chapter11_title.rs:14-200 — six planes, seven spheres, stripes/checkers, two glass spheres."""
    wall = Material(
        pattern=Pattern.stripes(Matrix.scaling(0.25, 0.25, 0.25) * Matrix.rotation_y(PI / 2.0),
                                Pattern.plain(Color.new(0.45, 0.45, 0.45)), Pattern.plain(Color.new(0.55, 0.55, 0.55))),
        ambient=0.0, diffuse=0.4, specular=0.0, reflective=0.3)
    floor = Element.plane(ShapeArgs(
        transform=Matrix.rotation_y(0.31415),
        material=Material(pattern=Pattern.checkers(Matrix.id(), Pattern.plain(Color.new(0.35, 0.35, 0.35)), Pattern.plain(Color.new(0.65, 0.65, 0.65))),
                          specular=0.0, reflective=0.4)))
    ceiling = Element.plane(ShapeArgs(transform=Matrix.translation(0.0, 5.0, 0.0),
                                      material=Material(pattern=Pattern.plain(Color.new(0.8, 0.8, 0.8)), ambient=0.3, specular=0.0)))
    west = Element.plane(ShapeArgs(transform=Matrix.translation(-5.0, 0.0, 0.0) * Matrix.rotation_z(PI / 2.0) * Matrix.rotation_y(PI / 2.0), material=wall))
    east = Element.plane(ShapeArgs(transform=Matrix.translation(5.0, 0.0, 0.0) * Matrix.rotation_z(PI / 2.0) * Matrix.rotation_y(PI / 2.0), material=wall))
    north = Element.plane(ShapeArgs(transform=Matrix.translation(0.0, 0.0, 5.0) * Matrix.rotation_x(PI / 2.0), material=wall))
    south = Element.plane(ShapeArgs(transform=Matrix.translation(0.0, 0.0, -5.0) * Matrix.rotation_x(PI / 2.0), material=wall))

    def ball(tx, ty, tz, s, rgb, **kw):
        t = Matrix.translation(tx, ty, tz) if s is None else Matrix.translation(tx, ty, tz) * Matrix.scaling(s, s, s)
        return Element.sphere(ShapeArgs(transform=t, material=Material(pattern=Pattern.plain(Color.new(*rgb)), **kw)))

    glass = dict(ambient=0.0, diffuse=0.4, specular=0.9, shininess=300.0, reflective=0.9, transparency=0.9, refractive_index=1.5)
    elements = [
        floor, ceiling, west, east, north, south,
        ball(4.6, 0.4, 1.0, 0.4, (0.8, 0.5, 0.3), shininess=50.0),
        ball(4.7, 0.3, 0.4, 0.3, (0.9, 0.4, 0.5), shininess=50.0),
        ball(-1.0, 0.5, 4.5, 0.5, (0.4, 0.9, 0.6), shininess=50.0),
        ball(-1.7, 0.3, 4.7, 0.3, (0.4, 0.6, 0.9), shininess=50.0),
        ball(-0.6, 1.0, 0.6, None, (1.0, 0.3, 0.2), specular=0.4, shininess=5.0),
        ball(0.6, 0.7, -0.6, 0.7, (0.0, 0.0, 0.2), **glass),
        ball(-0.7, 0.5, -0.8, 0.5, (0.0, 0.2, 0.0), **glass),
    ]
    world = World([PointLight(Color.white(), Vector.point(-4.9, 4.9, -1.0))], elements)
    return _cam(4096, 2160, 1.152, (-2.6, 1.5, -3.9), (-0.6, 1.0, -0.8), (0.0, 1.0, 0.0), hsize, vsize), world


def chapter12_title(hsize=None, vsize=None) -> Tuple[Camera, World]:
    """src/bin/chapter12_title.rs:14-268 — a room of 18 cubes: checkered floor/walls, table, glass cube, mirror."""
    def cube(t, rgb=None, pattern=None, **kw):
        pat = pattern if pattern is not None else Pattern.plain(Color.new(*rgb))
        return Element.cube(ShapeArgs(transform=t, material=Material(pattern=pat, **kw)))

    T, S, RY = Matrix.translation, Matrix.scaling, Matrix.rotation_y
    leg_mat = dict(rgb=(0.5529, 0.4235, 0.3255), ambient=0.2, diffuse=0.7)
    elements = [
        cube(S(20.0, 7.0, 20.0) * T(0.0, 1.0, 0.0), pattern=Pattern.checkers(S(0.07, 0.07, 0.07), Pattern.plain(Color.black()), Pattern.plain(Color.new(0.25, 0.25, 0.25))),
             ambient=0.25, diffuse=0.7, specular=0.9, shininess=300.0, reflective=0.1),
        cube(S(10.0, 10.0, 10.0), pattern=Pattern.checkers(S(0.05, 20.0, 0.05), Pattern.plain(Color.new(0.4863, 0.3765, 0.2941)), Pattern.plain(Color.new(0.3725, 0.2902, 0.2275))),
             ambient=0.1, diffuse=0.7, specular=0.9, shininess=300.0, reflective=0.1),
        cube(T(0.0, 3.1, 0.0) * S(3.0, 0.1, 2.0), pattern=Pattern.stripes(RY(0.1) * S(0.05, 0.05, 0.05), Pattern.plain(Color.new(0.5529, 0.4235, 0.3255)), Pattern.plain(Color.new(0.6588, 0.5098, 0.4000))),
             ambient=0.1, diffuse=0.7, specular=0.9, shininess=300.0, reflective=0.2),
        cube(T(2.7, 1.5, -1.7) * S(0.1, 1.5, 0.1), **leg_mat),
        cube(T(2.7, 1.5, 1.7) * S(0.1, 1.5, 0.1), **leg_mat),
        cube(T(-2.7, 1.5, -1.7) * S(0.1, 1.5, 0.1), **leg_mat),
        cube(T(-2.7, 1.5, 1.7) * S(0.1, 1.5, 0.1), **leg_mat),
        cube(T(0.0, 3.45001, 0.0) * RY(0.2) * S(0.25, 0.25, 0.25), (1.0, 1.0, 0.8), ambient=0.0, diffuse=0.3, specular=0.9, shininess=300.0,
             reflective=0.7, transparency=0.7, refractive_index=1.5),
        cube(T(1.0, 3.35, -0.9) * RY(-0.4) * S(0.15, 0.15, 0.15), (1.0, 0.5, 0.5), reflective=0.6, diffuse=0.4),
        cube(T(-1.5, 3.27, 0.3) * RY(0.4) * S(0.15, 0.07, 0.15), (1.0, 1.0, 0.5)),
        cube(T(0.0, 3.25, 1.0) * RY(0.4) * S(0.2, 0.05, 0.05), (0.5, 1.0, 0.5)),
        cube(T(-0.6, 3.4, -1.0) * RY(0.8) * S(0.05, 0.2, 0.05), (0.5, 0.5, 1.0)),
        cube(T(2.0, 3.4, 1.0) * RY(0.8) * S(0.05, 0.2, 0.05), (0.5, 1.0, 1.0)),
        cube(T(-10.0, 4.0, 1.0) * S(0.05, 1.0, 1.0), (0.7098, 0.2471, 0.2196), diffuse=0.6),
        cube(T(-10.0, 3.4, 2.7) * S(0.05, 0.4, 0.4), (0.2667, 0.2706, 0.6902), diffuse=0.6),
        cube(T(-10.0, 4.6, 2.7) * S(0.05, 0.4, 0.4), (0.3098, 0.5961, 0.3098), diffuse=0.6),
        cube(T(-2.0, 3.5, 9.95) * S(5.0, 1.5, 0.05), (0.3882, 0.2627, 0.1882), diffuse=0.6),
        cube(T(-2.0, 3.5, 9.95) * S(4.8, 1.4, 0.06), (0.0, 0.0, 0.0), ambient=0.0, diffuse=0.0, specular=1.0, shininess=300.0, reflective=1.0),
    ]
    world = World([PointLight(Color.new(1.0, 1.0, 0.9), Vector.point(0.0, 6.9, -5.0))], elements)
    return _cam(4096, 2160, 0.785, (8.0, 6.0, -8.0), (0.0, 3.0, 0.0), (0.0, 1.0, 0.0), hsize, vsize), world


def chapter13_title(hsize=None, vsize=None) -> Tuple[Camera, World]:
    """src/bin/chapter13_title.rs:14-245 — a checkered plane and ten cylinders (open, closed, concentric, glass)."""
    T, S, RY = Matrix.translation, Matrix.scaling, Matrix.rotation_y

    def cyl(t, rgb, mn, mx, closed, **kw):
        return Element.cylinder(ShapeArgs(transform=t, material=Material(pattern=Pattern.plain(Color.new(*rgb)), **kw)), mn, mx, closed)

    shiny = dict(ambient=0.1, specular=0.9, shininess=300.0)
    floor = Element.plane(ShapeArgs(material=Material(
        pattern=Pattern.checkers(RY(0.3) * S(0.25, 0.25, 0.25), Pattern.plain(Color.new(0.5, 0.5, 0.5)), Pattern.plain(Color.new(0.75, 0.75, 0.75))),
        ambient=0.2, diffuse=0.9, specular=0.0)))
    elements = [
        floor,
        cyl(T(-1.0, 0.0, 1.0) * S(0.5, 1.0, 0.5), (0.0, 0.0, 0.6), 0.0, 0.75, True, diffuse=0.1, specular=0.9, shininess=300.0, reflective=0.9),
        cyl(T(1.0, 0.0, 0.0) * S(0.8, 1.0, 0.8), (1.0, 1.0, 0.3), 0.0, 0.2, False, diffuse=0.8, **shiny),
        cyl(T(1.0, 0.0, 0.0) * S(0.6, 1.0, 0.6), (1.0, 0.9, 0.4), 0.0, 0.3, False, diffuse=0.8, **shiny),
        cyl(T(1.0, 0.0, 0.0) * S(0.4, 1.0, 0.4), (1.0, 0.8, 0.5), 0.0, 0.4, False, diffuse=0.8, **shiny),
        cyl(T(1.0, 0.0, 0.0) * S(0.2, 1.0, 0.2), (1.0, 0.7, 0.6), 0.0, 0.5, True, diffuse=0.8, **shiny),
        cyl(T(0.0, 0.0, -0.75) * S(0.05, 1.0, 0.05), (1.0, 0.0, 0.0), 0.0, 0.3, True, diffuse=0.9, **shiny),
        cyl(T(0.0, 0.0, -2.25) * RY(-0.15) * T(0.0, 0.0, 1.5) * S(0.05, 1.0, 0.05), (1.0, 1.0, 0.0), 0.0, 0.3, True, diffuse=0.9, **shiny),
        cyl(T(0.0, 0.0, -2.25) * RY(-0.3) * T(0.0, 0.0, 1.5) * S(0.05, 1.0, 0.05), (0.0, 1.0, 0.0), 0.0, 0.3, True, diffuse=0.9, **shiny),
        cyl(T(0.0, 0.0, -2.25) * RY(-0.45) * T(0.0, 0.0, 1.5) * S(0.05, 1.0, 0.05), (0.0, 1.0, 1.0), 0.0, 0.3, True, diffuse=0.9, **shiny),
        cyl(T(0.0, 0.0, -1.5) * S(0.33, 1.0, 0.33), (0.25, 0.0, 0.0), 0.0001, 0.5, True, diffuse=0.1, specular=0.9, shininess=300.0,
            reflective=0.9, transparency=0.9, refractive_index=1.5),
    ]
    world = World([PointLight(Color.white(), Vector.point(1.0, 6.9, -4.9))], elements)
    return _cam(4096, 2160, 0.314, (8.0, 3.5, -9.0), (0.0, 0.3, 0.0), (0.0, 1.0, 0.0), hsize, vsize), world


def chapter14_title(hsize=None, vsize=None) -> Tuple[Camera, World]:
    """src/bin/chapter14_title.rs:14-150 — three "wacky" objects: nested groups of spheres, open cylinders and open cones
    (all-infinite reference boxes), group materials, four lights, a far backdrop plane."""
    T, S, RX, RY, RZ = Matrix.translation, Matrix.scaling, Matrix.rotation_x, Matrix.rotation_y, Matrix.rotation_z

    def leg(t):
        sphere = Element.sphere(ShapeArgs(transform=T(0.0, 0.0, -1.0) * S(0.25, 0.25, 0.25)))
        cylinder = Element.cylinder(ShapeArgs(transform=T(0.0, 0.0, -1.0) * RY(-0.5236) * RZ(-1.5708) * S(0.25, 1.0, 0.25)), 0.0, 1.0, False)
        return Element.composite(t, None, GroupKind.Aggregation, [sphere, cylinder])

    def cap(t):
        cones = [Element.cone(ShapeArgs(transform=RY(PI / 3.0 * float(i)) * RX(-0.7854) * S(0.24606, 1.37002, 0.24606)), -1.0, 0.0, False) for i in range(6)]
        return Element.composite(t, None, GroupKind.Aggregation, cones)

    def wacky(t, material):
        els = [leg(RY(PI / 3.0 * float(i))) for i in range(6)]
        els.append(cap(T(0.0, 1.0, 0.0)))
        els.append(cap(RX(PI) * T(0.0, 1.0, 0.0)))
        return Element.composite(t, material, GroupKind.Aggregation, els)

    mat = lambda r, g, b: Material(pattern=Pattern.plain(Color.new(r, g, b)), ambient=0.2, diffuse=0.8, specular=0.7, shininess=20.0)
    elements = [
        Element.plane(ShapeArgs(transform=T(0.0, 0.0, 100.0) * RX(1.5708), material=Material(pattern=Pattern.plain(Color.white()), ambient=1.0, diffuse=0.0, specular=0.0))),
        wacky(T(-2.8, 0.0, 0.0) * RX(0.4363) * RY(0.1745), mat(0.9, 0.2, 0.4)),
        wacky(RY(0.1745), mat(0.2, 0.9, 0.6)),
        wacky(T(2.8, 0.0, 0.0) * RX(-0.4363) * RY(-0.1745), mat(0.2, 0.3, 1.0)),
    ]
    lights = [PointLight(Color.new(0.25, 0.25, 0.25), Vector.point(x, y, -10000.0)) for (x, y) in ((10000.0, 10000.0), (-10000.0, 10000.0), (10000.0, -10000.0), (-10000.0, -10000.0))]
    return _cam(4096, 2160, 0.9, (0.0, 0.0, -9.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), hsize, vsize), World(lights, elements)


def cover(hsize=None, vsize=None) -> Tuple[Camera, World]:
    """src/bin/cover.rs:14-215 — the book cover: a glass sphere and 17 cubes in nested groups, two lights, tilted camera."""
    T, S, RX = Matrix.translation, Matrix.scaling, Matrix.rotation_x
    white = dict(diffuse=0.7, ambient=0.1, specular=0.0, reflective=0.1)
    m_white = Material(pattern=Pattern.plain(Color.white()), **white)
    m_blue = Material(pattern=Pattern.plain(Color.new(0.537, 0.831, 0.914)), **white)
    m_red = Material(pattern=Pattern.plain(Color.new(0.941, 0.322, 0.388)), **white)
    m_purple = Material(pattern=Pattern.plain(Color.new(0.373, 0.404, 0.550)), **white)
    standard = S(0.5, 0.5, 0.5) * T(1.0, -1.0, 1.0)
    large, medium, small = S(3.5, 3.5, 3.5) * standard, S(3.0, 3.0, 3.0) * standard, S(2.0, 2.0, 2.0) * standard
    backdrop = Element.plane(ShapeArgs(transform=T(0.0, 0.0, 500.0) * RX(PI / 2.0), material=Material(pattern=Pattern.plain(Color.white()), ambient=1.0, diffuse=0.0, specular=0.0)))
    sphere = Element.sphere(ShapeArgs(transform=large, material=Material(pattern=Pattern.plain(Color.new(0.373, 0.404, 0.550)), diffuse=0.2, ambient=0.0, specular=1.0,
                                                                         shininess=200.0, reflective=0.7, transparency=0.7, refractive_index=1.5)))
    cube = lambda t, m: Element.cube(ShapeArgs(transform=t, material=m))
    top = [sphere,
           cube(T(4.0, 0.0, 0.0) * medium, m_white), cube(T(8.5, 1.5, -0.5) * large, m_blue), cube(T(0.0, 0.0, 4.0) * large, m_red),
           cube(T(4.0, 0.0, 4.0) * small, m_white), cube(T(7.5, 0.5, 4.0) * medium, m_purple), cube(T(-0.25, 0.25, 8.0) * medium, m_white),
           cube(T(4.0, 1.0, 7.5) * large, m_blue), cube(T(10.0, 2.0, 7.5) * medium, m_red), cube(T(8.0, 2.0, 12.0) * small, m_white),
           cube(T(20.0, 1.0, 9.0) * small, m_white)]
    bot = [cube(T(-0.5, -5.0, 0.25) * large, m_blue), cube(T(4.0, -4.0, 0.0) * large, m_red), cube(T(8.5, -4.0, 0.0) * large, m_white),
           cube(T(0.0, -4.0, 4.0) * large, m_white), cube(T(-0.5, -4.5, 8.0) * large, m_purple), cube(T(0.0, -8.0, 4.0) * large, m_white),
           cube(T(-0.5, -8.5, 8.0) * large, m_white)]
    group_all = Element.composite(Matrix.id(), None, GroupKind.Aggregation, [
        Element.composite(Matrix.id(), None, GroupKind.Aggregation, top), Element.composite(Matrix.id(), None, GroupKind.Aggregation, bot)])
    world = World([PointLight(Color.white(), Vector.point(50.0, 100.0, -50.0)), PointLight(Color.new(0.2, 0.2, 0.2), Vector.point(-400.0, 50.0, -10.0))],
                  [backdrop, group_all])
    return _cam(4096, 4096, 0.785, (-6.0, 6.0, -10.0), (6.0, 0.0, 6.0), (-0.45, 1.0, 0.0), hsize, vsize), world


def chapter14_hexagon(hsize=None, vsize=None) -> Tuple[Camera, World]:
    """src/bin/chapter14_hexagon.rs:16-88 — nested groups of spheres + open cylinders, Simplex point-jitter."""
    def corner():
        return Element.sphere(ShapeArgs(transform=Matrix.translation(0.0, 0.0, -1.0) * Matrix.scaling(0.25, 0.25, 0.25)))

    def edge():
        return Element.cylinder(ShapeArgs(transform=Matrix.translation(0.0, 0.0, -1.0) * Matrix.rotation_y(-PI / 6.0)
                                          * Matrix.rotation_z(-PI / 2.0) * Matrix.scaling(0.25, 1.0, 0.25)), 0.0, 1.0, False)

    def side(t):
        return Element.composite(t, None, GroupKind.Aggregation, [corner(), edge()])

    material = Material(pattern=Pattern.point_jitter(
        Noise.Simplex(0.3),
        Pattern.stripes(Matrix.rotation_z(PI / 2.0) * Matrix.scaling(0.05, 0.05, 0.05), Pattern.plain(Color.white()), Pattern.plain(Color.black()))))
    sides = [side(Matrix.rotation_y(float(n) * PI / 3.0)) for n in range(6)]
    hexagon = Element.composite(Matrix.id(), material, GroupKind.Aggregation, sides)
    world = World([PointLight(Color.white(), Vector.point(1.0, 6.9, -4.9))], [hexagon])
    return _cam(4096, 2160, 0.314, (8.0, 5.0, 8.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), hsize, vsize), world


def chapter14_benchmark(hsize=None, vsize=None) -> Tuple[Camera, World]:
    """src/bin/chapter14_benchmark.rs:14-111 — plane + 8 nested groups x 125 spheres, gradient material."""
    def sphere_block(n, h):
        out = []
        for x in range(n):
            for z in range(n):
                for y in range(n):
                    out.append(Element.sphere(ShapeArgs(
                        transform=Matrix.translation(0.1 + float(x) * h, 0.1 + float(y) * h, 0.1 + float(z) * h) * Matrix.scaling(0.1, 0.1, 0.1),
                        material=Material(pattern=Pattern.plain(Color.black())))))
        return out

    backdrop = Element.plane(ShapeArgs(material=Material(pattern=Pattern.plain(Color.white()), specular=0.0)))
    n, h = 5, 0.3
    offset = float(n - 1) * (0.1 + h) - 0.1
    groups = []
    for x in range(2):
        for z in range(2):
            for y in range(2):
                groups.append(Element.composite(Matrix.translation(float(x) * offset, float(y) * offset, float(z) * offset),
                                                None, GroupKind.Aggregation, sphere_block(n, h)))
    mat = Material(pattern=Pattern.gradient(Matrix.scaling(1.0, 2.9, 1.0) * Matrix.rotation_z(PI / 2.0),
                                            Pattern.plain(Color.new(1.0, 0.0, 0.0)), Pattern.plain(Color.new(0.0, 0.0, 1.0))))
    elements = [backdrop, Element.composite(Matrix.id(), mat, GroupKind.Aggregation, groups)]
    world = World([PointLight(Color.white(), Vector.point(-5.0, 7.0, -1.0))], elements)
    return _cam(4096, 2160, 1.0, (-8.0, 8.0, -8.0), (2.0, 2.0, 2.0), (0.0, 1.0, 0.0), hsize, vsize), world


def chapter15_teapot(obj: str = "teapot_high.obj", hsize=None, vsize=None) -> Tuple[Camera, World]:
    """src/bin/chapter15_teapot.rs:16-92.  BASELINE config 3 = teapot_low @1920x1080, config 4 = teapot_high @3840x2160."""
    path = obj if os.path.isabs(obj) else os.path.join(ASSETS, "obj", obj)
    teapot = Element.obj(path, Matrix.rotation_x(-PI / 2.0), Material(
        pattern=Pattern.plain(Color.new(0.7, 0.7, 1.0)), ambient=0.1, diffuse=0.6, specular=0.4, reflective=0.1, shininess=5.0))
    material = Material(pattern=Pattern.plain(Color.black()), ambient=0.02, diffuse=0.7, specular=0.0, reflective=0.5)
    floor = Element.plane(ShapeArgs(material=material))
    left = Element.plane(ShapeArgs(material=material, transform=Matrix.rotation_y(-PI / 4.0) * Matrix.translation(0.0, 0.0, 30.0) * Matrix.rotation_x(PI / 2.0)))
    right = Element.plane(ShapeArgs(material=material, transform=Matrix.rotation_y(PI / 4.0) * Matrix.translation(0.0, 0.0, 30.0) * Matrix.rotation_x(PI / 2.0)))
    world = World([PointLight(Color.new(0.7, 0.7, 0.7), Vector.point(-100.0, 100.0, -100.0)),
                   PointLight(Color.new(0.7, 0.7, 0.7), Vector.point(100.0, 100.0, -100.0))],
                  [floor, left, right, teapot])
    return _cam(4096, 2160, 1.4, (0.0, 30.0, -50.0), (0.0, 0.0, 10.0), (0.0, 1.0, 0.0), hsize, vsize), world


class SplitMix64:
    """Documented generator for the synthetic scenes (seed 12345)."""

    def __init__(self, seed: int):
        self.s = seed & 0xFFFFFFFFFFFFFFFF

    def next(self) -> int:
        self.s = (self.s + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        return z ^ (z >> 31)

    def uniform(self, lo=0.0, hi=1.0) -> float:
        return lo + (hi - lo) * ((self.next() >> 11) * (1.0 / 9007199254740992.0))


def synthetic_analytic(n_primitives=512, seed=12345, cones=False, grouped=False, hsize=1920, vsize=1080) -> Tuple[Camera, World]:
    """BASELINE config 2: 3 planes as in the teapot scene + ``n_primitives`` analytic primitives
    placed uniformly in a 40x20x40 box above the floor; 60% matte / 25% reflective / 15% glass; 2 lights; fov 1.4.
    cones=False: 50% spheres, 25% cubes, 25% closed cylinders (the headline variant);
    cones=True: 40/25/25 + 10% closed cones.  grouped=True nests them into an 8-cell grid of Aggregation groups."""
    rng = SplitMix64(seed)
    plane_mat = Material(pattern=Pattern.plain(Color.black()), ambient=0.02, diffuse=0.7, specular=0.0, reflective=0.5)
    floor = Element.plane(ShapeArgs(material=plane_mat))
    left = Element.plane(ShapeArgs(material=plane_mat, transform=Matrix.rotation_y(-PI / 4.0) * Matrix.translation(0.0, 0.0, 30.0) * Matrix.rotation_x(PI / 2.0)))
    right = Element.plane(ShapeArgs(material=plane_mat, transform=Matrix.rotation_y(PI / 4.0) * Matrix.translation(0.0, 0.0, 30.0) * Matrix.rotation_x(PI / 2.0)))
    prims, cells = [], {}
    for _ in range(n_primitives):
        u = rng.uniform()
        if cones:
            kind = "sphere" if u < 0.40 else "cube" if u < 0.65 else "cylinder" if u < 0.90 else "cone"
        else:
            kind = "sphere" if u < 0.50 else "cube" if u < 0.75 else "cylinder"
        px, py, pz = rng.uniform(-20.0, 20.0), rng.uniform(1.5, 21.5), rng.uniform(-15.0, 25.0)
        s = rng.uniform(0.3, 1.5)
        rx, ry, rz = rng.uniform(0.0, 2 * PI), rng.uniform(0.0, 2 * PI), rng.uniform(0.0, 2 * PI)
        t = Matrix.translation(px, py, pz) * Matrix.rotation_z(rz) * Matrix.rotation_y(ry) * Matrix.rotation_x(rx) * Matrix.scaling(s, s, s)
        col = Color.new(rng.uniform(0.1, 1.0), rng.uniform(0.1, 1.0), rng.uniform(0.1, 1.0))
        m = rng.uniform()
        if m < 0.60:
            mat = Material(pattern=Pattern.plain(col), specular=0.3, shininess=50.0)
        elif m < 0.85:
            mat = Material(pattern=Pattern.plain(col), reflective=rng.uniform(0.2, 0.9))
        else:
            mat = Material(pattern=Pattern.plain(Color.new(col.r * 0.2, col.g * 0.2, col.b * 0.2)), diffuse=0.3, transparency=0.9, refractive_index=1.5, reflective=0.9)
        args = ShapeArgs(transform=t, material=mat)
        if kind == "sphere":
            e = Element.sphere(args)
        elif kind == "cube":
            e = Element.cube(args)
        elif kind == "cylinder":
            e = Element.cylinder(args, 0.0, 1.0, True)
        else:
            e = Element.cone(args, -1.0, 0.0, True)
        prims.append(e)
        cells.setdefault((px >= 0.0, py >= 11.5, pz >= 5.0), []).append(e)
    if grouped:
        body = [Element.composite(Matrix.id(), None, GroupKind.Aggregation, v) for _, v in sorted(cells.items())]
    else:
        body = prims
    world = World([PointLight(Color.new(0.7, 0.7, 0.7), Vector.point(-100.0, 100.0, -100.0)),
                   PointLight(Color.new(0.7, 0.7, 0.7), Vector.point(100.0, 100.0, -100.0))],
                  [floor, left, right] + body)
    return _cam(1920, 1080, 1.4, (0.0, 30.0, -50.0), (0.0, 0.0, 10.0), (0.0, 1.0, 0.0), hsize, vsize), world


def write_heightfield_obj(path: str, nx: int, nz: int, seed: int = 12345) -> int:
    """Writes a displaced nx x nz vertex grid as ONE OBJ group with analytic smooth normals
    (`f a//a b//b c//c`, two triangles per cell) and returns the triangle count.  The surface is
    y = sum of three seeded sine waves over x, z in [-20, 20]; deterministic for (nx, nz, seed)."""
    import numpy as np
    rng = SplitMix64(seed)
    waves = [(rng.uniform(0.15, 0.5), rng.uniform(0.15, 0.5), rng.uniform(0.0, 2 * PI), rng.uniform(0.4, 1.2)) for _ in range(3)]
    xs = np.linspace(-20.0, 20.0, nx)
    zs = np.linspace(-20.0, 20.0, nz)
    X, Z = np.meshgrid(xs, zs, indexing="xy")  # (nz, nx)
    Y = np.zeros_like(X)
    dYdx = np.zeros_like(X)
    dYdz = np.zeros_like(X)
    for (kx, kz, ph, amp) in waves:
        arg = kx * X + kz * Z + ph
        Y += amp * np.sin(arg)
        dYdx += amp * kx * np.cos(arg)
        dYdz += amp * kz * np.cos(arg)
    N = np.stack([-dYdx, np.ones_like(X), -dYdz], axis=-1)
    N /= np.linalg.norm(N, axis=-1, keepdims=True)
    V = np.stack([X, Y, Z], axis=-1).reshape(-1, 3)
    Nf = N.reshape(-1, 3)
    idx = (np.arange(nz - 1)[:, None] * nx + np.arange(nx - 1)[None, :]).reshape(-1) + 1  # 1-based index of each cell's corner
    a, b, c, d = idx, idx + 1, idx + nx, idx + nx + 1
    tris = np.concatenate([np.stack([a, c, b], axis=1), np.stack([b, c, d], axis=1)], axis=0)
    with open(path, "w") as f:
        f.write("# synthetic heightfield %dx%d seed %d\ng Heightfield\n" % (nx, nz, seed))
        np.savetxt(f, V, fmt="v %.17g %.17g %.17g")
        np.savetxt(f, Nf, fmt="vn %.17g %.17g %.17g")
        np.savetxt(f, np.concatenate([tris, tris], axis=1)[:, [0, 3, 1, 4, 2, 5]], fmt="f %d//%d %d//%d %d//%d")
    return int(tris.shape[0])


def synthetic_mesh(obj_path: str, nx: int = 708, nz: int = 708, seed: int = 12345, hsize: int = 3840, vsize: int = 2160) -> Tuple[Camera, World]:
    """BASELINE config 5: a ~10^6-triangle synthetic smooth mesh (708 x 708 grid -> 999 698 triangles)
    with `point_jitter(Fractal{0.3, 4}, stripes)`, a second object with `Noise::Simplex{0.3}`, reflective 0.1, floor plane,
    two lights; 4K, fuel 8.  `obj_path` is written if it does not exist (the reference's on-disk format is the interface)."""
    if not os.path.exists(obj_path):
        write_heightfield_obj(obj_path, nx, nz, seed)
    mesh_mat = Material(pattern=Pattern.point_jitter(Noise.Fractal(0.3, 4), Pattern.stripes(Matrix.scaling(0.7, 0.7, 0.7), Pattern.plain(Color.new(0.9, 0.85, 0.6)), Pattern.plain(Color.new(0.3, 0.45, 0.25)))),
                        diffuse=0.8, specular=0.2, shininess=20.0, reflective=0.1)
    mesh = Element.obj(obj_path, Matrix.translation(0.0, 2.0, 10.0), mesh_mat)
    ball = Element.sphere(ShapeArgs(transform=Matrix.translation(-6.0, 9.0, 2.0) * Matrix.scaling(3.0, 3.0, 3.0), material=Material(
        pattern=Pattern.point_jitter(Noise.Simplex(0.3), Pattern.checkers(Matrix.scaling(0.4, 0.4, 0.4), Pattern.plain(Color.new(0.9, 0.2, 0.2)), Pattern.plain(Color.white()))),
        reflective=0.1)))
    glass = Element.sphere(ShapeArgs(transform=Matrix.translation(7.0, 8.0, 0.0) * Matrix.scaling(2.5, 2.5, 2.5), material=Material(
        pattern=Pattern.plain(Color.new(0.05, 0.05, 0.1)), diffuse=0.2, transparency=0.9, reflective=0.9, refractive_index=1.5)))
    floor = Element.plane(ShapeArgs(transform=Matrix.translation(0.0, -3.0, 0.0), material=Material(pattern=Pattern.plain(Color.new(0.2, 0.2, 0.25)), reflective=0.3, specular=0.0)))
    world = World([PointLight(Color.new(0.7, 0.7, 0.7), Vector.point(-100.0, 100.0, -100.0)), PointLight(Color.new(0.5, 0.5, 0.5), Vector.point(60.0, 80.0, -40.0))],
                  [floor, mesh, ball, glass])
    return _cam(3840, 2160, 1.2, (0.0, 22.0, -32.0), (0.0, 2.0, 8.0), (0.0, 1.0, 0.0), hsize, vsize), world
