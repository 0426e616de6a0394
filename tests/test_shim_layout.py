"""The Rust shim (shim/gpu.rs) is not compiled in this image, so the one thing that can silently rot — the `#[repr(C)]` mirrors of the
structs of include/rtc.h — is checked textually: same fields, same order, same widths."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
C_TO_RUST = {"uint64_t": "u64", "uint32_t": "u32", "int32_t": "i32", "double": "f64", "int": "c_int", "float": "f32"}


def c_struct(text, name):
    m = re.search(r"typedef struct %s\s*\{(.*?)\}\s*%s\s*;" % (name, name), text, re.S)
    assert m, name
    body = re.sub(r"/\*.*?\*/", "", m.group(1), flags=re.S)
    fields = []
    for stmt in body.split(";"):
        stmt = " ".join(stmt.split())
        if not stmt:
            continue
        ty, rest = stmt.split(" ", 1)
        if ty == "const":
            ty, rest = rest.split(" ", 1)
        for decl in rest.split(","):
            decl = decl.strip()
            arr = re.search(r"\[(\d+)\]", decl)
            nm = re.sub(r"\[.*", "", decl).lstrip("*")
            fields.append((nm, "ptr" if ("*" in decl or "*" in ty) else C_TO_RUST[ty], int(arr.group(1)) if arr else 0))
    return fields


def rust_struct(text, name):
    text = re.sub(r"//[^\n]*", "", text)
    m = re.search(r"pub struct %s\s*\{(.*?)\}" % name, text, re.S)
    assert m, name
    fields = []
    for line in m.group(1).splitlines():
        line = line.split("//")[0].strip().rstrip(",")
        if not line:
            continue
        nm, ty = [x.strip() for x in line.replace("pub ", "").split(":", 1)]
        arr = re.match(r"\[(\w+);\s*(\d+)\]", ty)
        if arr:
            fields.append((nm, arr.group(1), int(arr.group(2))))
        else:
            fields.append((nm, "ptr" if ty.startswith("*") else ty, 0))
    return fields


def test_shim_structs_mirror_the_header():
    h = open(os.path.join(ROOT, "include", "rtc.h")).read()
    rs = open(os.path.join(ROOT, "shim", "gpu.rs")).read()
    for c_name, rust_name in (("rtc_stats", "RtcStats"), ("rtc_camera", "RtcCamera"), ("rtc_hit", "RtcHit"), ("rtc_prim", "RtcPrim"), ("rtc_xform", "RtcXform"),
                              ("rtc_material", "RtcMaterial"), ("rtc_pattern_node", "RtcPatternNode"), ("rtc_light", "RtcLight"), ("rtc_node", "RtcNode"),
                              ("rtc_scene_desc", "RtcSceneDesc")):
        c, r = c_struct(h, c_name), rust_struct(rs, rust_name)
        keywords = {"ref", "type", "match", "move", "box", "fn", "in", "loop"}   # C field names Rust cannot spell: the shim renames them
        assert all(a[0] == b[0] or a[0] in keywords for a, b in zip(c, r)) and len(c) == len(r), (c_name, c, r)
        assert [f[1:] for f in c] == [f[1:] for f in r], (c_name, c, r)
