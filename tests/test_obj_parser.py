"""The product's OBJ loader (csrc/host_scene.hpp: parse_obj_text, reached through rtw_parse_obj — host code, no GPU needed) on
the reference's own parser cases (src/obj.rs:289-673) and on grammar corner cases (tests/golden/obj_cases.json): ignored-line and
triangle counts, the group tree and the triangle records it flattens to, against (i) the fixture's expected values, (ii) the
oracle's separately written parser, (iii) the tree the reference's test expects, built with Element::composite / Element::triangle
and flattened by the foreign flattener; and rendered (emulated kernels vs oracle) for the hits."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import foreign_flattener as ff
import raytracer_challenge_amd as rt
from raytracer_challenge_amd.scene import Camera, Color, Element, GroupKind, Material, Matrix, Pattern, PointLight, ShapeArgs, Vector, World

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = json.load(open(os.path.join(ROOT, "tests", "golden", "obj_cases.json")))["cases"]


@pytest.fixture(scope="module")
def product():
    return rt.Backend(os.path.join(ROOT, "raytracer_challenge_amd", "csrc", "librtc_amd.so"))   # host side only: no device is touched


def parse(be, path, transform=None, material=None):
    """rtw_parse_obj -> (ignored, triangles, NativeWorld holding the element)."""
    from raytracer_challenge_amd.backend import _MaterialC, _d16
    cache, owned = {}, []
    mat = be._material(material or Material(), cache, owned)
    ign, tris = C.c_uint64(0), C.c_uint64(0)
    h = be.lib.rtw_parse_obj(path.encode(), _d16(transform or Matrix.id()), C.byref(mat), C.byref(ign), C.byref(tris))
    for p in owned:
        be.lib.rtw_pattern_release(p)
    assert h, be._err()
    w = be.lib.rtw_world_create()
    assert be.lib.rtw_world_add_element(w, h) == 0
    from raytracer_challenge_amd.backend import NativeWorld
    return int(ign.value), int(tris.value), NativeWorld(be, w, 0)


def flat_desc(be, nw):
    be.lib.rtw_world_flatten_desc.restype = C.c_int
    be.lib.rtw_world_flatten_desc.argtypes = [C.c_void_p, C.POINTER(ff.RtcSceneDesc)]
    d = ff.RtcSceneDesc()
    assert be.lib.rtw_world_flatten_desc(nw.handle, C.byref(d)) == 0, be._err()
    return d


def expected_element(case):
    """The tree src/obj.rs:237-257 builds: one Aggregation composite per non-empty group (transform, material), wrapped in an outer
    identity group if there are several (the reference's `obj` test spells the two-group case out, :610-672)."""
    tris, k = [], 0
    for f, smooth, nrm in zip(case["faces"], case["smooth"], case.get("normals") or [None] * len(case["faces"])):
        p = [Vector.point(*v) for v in f]
        tris.append(Element.smooth_triangle(ShapeArgs(), *p, *[Vector.vector(*n) for n in nrm]) if smooth else Element.triangle(ShapeArgs(), *p))
    groups = []
    for n in case["groups"]:
        groups.append(Element.composite(Matrix.id(), Material(), GroupKind.Aggregation, tris[k:k + n]))
        k += n
    return groups[0] if len(groups) == 1 else Element.composite(Matrix.id(), None, GroupKind.Aggregation, groups)


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_product_obj_parser_on_the_reference_cases(product, orc, case, tmp_path):
    path = str(tmp_path / (case["name"] + ".obj"))
    open(path, "w").write(case["text"])
    ign, tris, nw = parse(product, path)
    o_ign, o_tris, _ = parse(orc, path)
    assert (ign, tris) == (case["ignored"], case["triangles"]), "product vs fixture"
    assert (o_ign, o_tris) == (case["ignored"], case["triangles"]), "oracle vs fixture"
    d = flat_desc(product, nw)
    kinds = [d.nodes[i].kind for i in range(d.n_nodes)]
    n_groups = len(case["groups"])
    # group tree shape: [outer] + per group: group node + its triangles (an OBJ without faces is one empty group)
    want_kinds = ([3] if n_groups != 1 else []) + sum(([3] + [-1] * n for n in case["groups"]), [])
    assert kinds == want_kinds
    assert d.n_prims == case["triangles"] == d.n_tris
    if not case["triangles"]:
        return
    # the flattened records are, bit for bit, those of the tree the reference's test expects
    exp = ff.flatten(World([], [expected_element(case)])).desc()
    assert d.n_nodes == exp.n_nodes and d.n_prims == exp.n_prims and d.n_tris == exp.n_tris
    n9 = 9 * d.n_tris
    assert list(d.tri_p1e1e2[:n9]) == list(exp.tri_p1e1e2[:n9])
    for i in range(d.n_prims):
        assert (d.prims[i].geometry, d.prims[i].flags, d.prims[i].data) == (exp.prims[i].geometry, exp.prims[i].flags, exp.prims[i].data)
        g = d.prims[i].geometry
        nn = 9 if g == 6 else 3   # flat triangles: {n, -, -}
        assert list(d.tri_normals[9 * i: 9 * i + nn]) == list(exp.tri_normals[9 * i: 9 * i + nn])
    for i in range(d.n_nodes):
        assert (d.nodes[i].kind, d.nodes[i].skip) == (exp.nodes[i].kind, exp.nodes[i].skip)
        if d.nodes[i].kind >= 0:
            assert list(d.nodes[i].bbox_min) == list(exp.nodes[i].bbox_min) and list(d.nodes[i].bbox_max) == list(exp.nodes[i].bbox_max)


def test_multi_group_obj_renders_like_the_oracle(orc, tmp_path):
    """Two groups, flat and smooth faces, a polygon, ignored lines, under a transform and a material: the product's loader +
    flattener + (emulated) kernels against the oracle's loader + the reference algorithm: hits bit-exact, colours <= 1e-5."""
    from emu_lib import emu
    from parity import assert_parity
    txt = ("# two groups\nv -1 0 0\nv 1 0 0\nv 1 2 0\nv -1 2 0\nv 0 3 0.5\nv 0 1 -1.5\nvn 0 0 -1\nvn 0.3 0.2 -1\nvn -0.3 0.2 -1\n"
           "g Wall\nf 1 2 3 4 5\nusemtl x\ng Fin\nf 1//1 2//2 6//3\nf 2/9/2 3/9/1 6/9/3\nf 4 1 6\n")
    path = str(tmp_path / "two_groups.obj")
    open(path, "w").write(txt)
    mat = Material(pattern=Pattern.plain(Color.new(0.8, 0.4, 0.3)), reflective=0.2)
    obj = Element.obj(path, Matrix.translation(0.1, -0.5, 0.3) * Matrix.rotation_y(0.5) * Matrix.scaling(1.2, 1.0, 1.2), mat)
    floor = Element.plane(ShapeArgs(transform=Matrix.translation(0, -1, 0)))
    world = World([PointLight(Color.white(), Vector.point(-4, 6, -6))], [floor, obj])
    cam = Camera.new(80, 60, 1.0, Camera.transform(Vector.point(0.5, 2.0, -6), Vector.point(0, 1, 0), Vector.vector(0, 1, 0)))
    assert_parity(emu(), orc, world, cam, 5, label="two-group OBJ")


def _random_obj_text(rng):
    """A syntactically ordinary OBJ file written in as many spellings as the grammar has (src/obj.rs:54-149): number formats, runs of blanks
    and tabs, trailing blanks, CRLF, comments and statements the parser ignores, faces as v, v/t, v//n and v/t/n, polygons, several groups."""
    def num(x):
        f = int(rng.integers(0, 6))
        return [repr(float(x)), "%.4f" % x, "%e" % x, "%+.3f" % x, "%g" % x, ("%d" % round(x)) if abs(x - round(x)) < 1e-9 else "%.17g" % x][f]

    def sep():
        return [" ", "  ", "\\t", " \\t "][int(rng.integers(0, 4))]

    lines, n_v, n_n = [], 0, 0
    junk = ["# a comment", "", "vt 0.5 0.25", "usemtl shiny", "s off", "o thing", "mtllib x.mtl", "   ", "vp 0.1 0.2"]
    for _ in range(int(rng.integers(4, 14))):
        v = rng.uniform(-2, 2, 3)
        lines.append("v" + sep() + sep().join(num(c) for c in v) + ("  " if rng.random() < 0.2 else ""))
        n_v += 1
        if rng.random() < 0.15:
            lines.append(junk[int(rng.integers(0, len(junk)))])
    for _ in range(int(rng.integers(0, 6))):
        n = rng.normal(size=3)
        lines.append("vn" + sep() + sep().join(num(c) for c in n))
        n_n += 1
    for g in range(int(rng.integers(1, 4))):
        if g or rng.random() < 0.7:
            lines.append("g" + sep() + "Group%d" % g)
        for _ in range(int(rng.integers(0, 5))):
            k = int(rng.integers(3, 7))
            idx = [int(i) + 1 for i in rng.choice(n_v, size=k, replace=k > n_v)]
            style = int(rng.integers(0, 4)) if n_n else int(rng.integers(0, 2))
            def ref(i):
                if style == 0:
                    return "%d" % i
                if style == 1:
                    return "%d/%d" % (i, int(rng.integers(1, 9)))
                nn = int(rng.integers(1, n_n + 1))
                return "%d//%d" % (i, nn) if style == 2 else "%d/%d/%d" % (i, int(rng.integers(1, 9)), nn)
            lines.append("f" + sep() + sep().join(ref(i) for i in idx))
            if rng.random() < 0.2:
                lines.append(junk[int(rng.integers(0, len(junk)))])
    eol = "\\r\\n" if rng.random() < 0.25 else "\\n"
    return eol.join(lines) + (eol if rng.random() < 0.8 else "")


@pytest.mark.parametrize("seed", range(int(os.environ.get("RTC_OBJ_FUZZ_SEEDS", "24"))))
def test_obj_parser_fuzz_product_vs_oracle(product, orc, seed, tmp_path):
    """Random ordinary OBJ files: the product's loader and the oracle's separately written one agree on what they ignore and how many
    triangles they make, and the parsed mesh renders to the same hits (emulated kernels vs the reference algorithm over the oracle's parse)."""
    from emu_lib import emu
    from parity import assert_parity
    rng = np.random.default_rng(9000 + seed)
    path = str(tmp_path / ("fuzz_%d.obj" % seed))
    with open(path, "w", newline="") as f:
        f.write(_random_obj_text(rng))
    ign, tris, _ = parse(product, path)
    o_ign, o_tris, _ = parse(orc, path)
    assert (ign, tris) == (o_ign, o_tris), open(path).read()
    if tris == 0:
        return
    mat = Material(pattern=Pattern.plain(Color.new(0.7, 0.5, 0.3)), reflective=0.2)
    world = World([PointLight(Color.white(), Vector.point(-4, 6, -6))], [Element.plane(ShapeArgs(transform=Matrix.translation(0, -2.5, 0))), Element.obj(path, Matrix.rotation_y(0.4), mat)])
    cam = Camera.new(48, 32, 1.0, Camera.transform(Vector.point(0.5, 1.0, -7), Vector.point(0, 0, 0), Vector.vector(0, 1, 0)))
    assert_parity(emu(), orc, world, cam, 3, label="fuzzed OBJ %d" % seed)


def test_threaded_flatten_of_a_large_obj_group_equals_the_serial_walk(product, tmp_path, monkeypatch):
    """host_scene.hpp Flattener: a group of >= 32 768 triangles with one material and one set of matrices (an OBJ group) is written by
    several threads; the descriptor must be byte for byte the one the element-by-element walk emits (RTC_FLATTEN_SERIAL=1), for flat and
    smooth triangles, with a second, small group beside it (which takes the ordinary walk)."""
    import numpy as np
    from raytracer_challenge_amd import scenes
    path = str(tmp_path / "big.obj")
    scenes.write_heightfield_obj(path, 150, 150, 77)           # 44 402 smooth triangles, one group
    with open(path, "a") as f:                                  # + a small second group of flat triangles
        f.write("g tail\nv 0 0 0\nv 1 0 0\nv 0 1 0\nv 1 1 0\nf 22501 22502 22503\nf 22502 22504 22503\n")
    t = Matrix.translation(1.0, 2.0, 3.0) * Matrix.scaling(0.5, 0.5, 0.5)
    _, tris, nw = parse(product, path, t, Material(reflective=0.25))
    assert tris > 32768

    def snapshot():
        d = flat_desc(product, nw)
        arr = lambda p, n, size: bytes(C.string_at(C.cast(p, C.c_void_p), n * size))
        return (d.n_nodes, d.n_prims, d.n_xforms, d.n_tris, d.n_materials,
                arr(d.nodes, d.n_nodes, C.sizeof(ff.RtcNode)), arr(d.prims, d.n_prims, C.sizeof(ff.RtcPrim)),
                arr(d.tri_p1e1e2, d.n_tris, 72), arr(d.tri_normals, d.n_tris, 72), arr(d.xforms, d.n_xforms, C.sizeof(ff.RtcXform)))

    threaded = snapshot()
    monkeypatch.setenv("RTC_FLATTEN_SERIAL", "1")
    serial = snapshot()
    assert threaded[:5] == serial[:5]
    for a, b, name in zip(threaded[5:], serial[5:], ("nodes", "prims", "tri_p1e1e2", "tri_normals", "xforms")):
        assert a == b, name
