"""crates/skinrt-sys/src/lib.rs against include/skinrt.h.  The build image has no Rust
toolchain, so the bindings cannot be compiled here; instead this test
  * compiles a C program that prints sizeof / offsetof of EVERY struct and field the header declares,
  * parses every `#[repr(C)] pub struct` of the Rust file, lays it out by the C rules (natural alignment),
    and requires the same field names, order, offsets and sizes,
  * requires the same of the ctypes mirror (renderer_amd/abi.py),
  * requires the `extern "C"` block to declare exactly the header's functions with the same arity,
  * requires the Rust constants to equal the header's #defines.
CPU only."""
import ctypes as C
import os
import re
import subprocess

from renderer_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "skinrt.h")
RUST = os.path.join(ROOT, "crates", "skinrt-sys", "src", "lib.rs")

PRIM = {"u8": (1, 1), "i8": (1, 1), "u16": (2, 2), "i16": (2, 2), "u32": (4, 4), "i32": (4, 4), "f32": (4, 4),
        "u64": (8, 8), "i64": (8, 8), "f64": (8, 8), "c_int": (4, 4)}


def header_text():
    return re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)


def header_structs():
    """{struct: [field, ...]} for every `typedef struct X { ... } X;` with a body"""
    out = {}
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*\1\s*;", header_text(), flags=re.S):
        fields = []
        for decl in m.group(2).split(";"):
            decl = decl.strip()
            if not decl:
                continue
            names = decl.split(",")
            first = re.search(r"(\w+)\s*$", names[0]).group(1)
            fields.append(first)
            fields += [n.strip().lstrip("*").strip() for n in names[1:]]
        out[m.group(1)] = fields
    return out


def c_layout(structs, tmp_path):
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "skinrt.h"', "int main(void) {"]
    for s, fields in structs.items():
        lines.append(f'printf("{s} %zu\\n", sizeof({s}));')
        for f in fields:
            lines.append(f'printf("{s}.{f} %zu\\n", offsetof({s}, {f}));')
    lines += ["return 0; }"]
    src = tmp_path / "layout_all.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout_all"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout
    return {k: int(v) for k, v in (l.rsplit(" ", 1) for l in out.splitlines())}


def rust_structs():
    text = re.sub(r"//.*", "", open(RUST).read())
    out = {}
    for m in re.finditer(r"#\[repr\(C\)\]\s*(?:#\[derive\([^\]]*\)\]\s*)?pub struct (\w+)\s*\{(.*?)\}", text, flags=re.S):
        fields = []
        for f in m.group(2).split(","):
            f = f.strip()
            if not f:
                continue
            fm = re.match(r"(?:pub\s+)?(\w+)\s*:\s*(.+)$", f, flags=re.S)
            fields.append((fm.group(1), fm.group(2).strip()))
        out[m.group(1)] = fields
    return out


def rust_layout(fields):
    """C layout of a #[repr(C)] struct: each field at the next multiple of its alignment, size rounded up to the largest"""
    off, align_max, offs = 0, 1, {}
    for name, ty in fields:
        if ty.startswith("*const") or ty.startswith("*mut"):
            size, align = 8, 8
        elif ty in PRIM:
            size, align = PRIM[ty]
        else:
            raise AssertionError(f"type {ty!r} of field {name} not understood")
        off = (off + align - 1) // align * align
        offs[name] = off
        off += size
        align_max = max(align_max, align)
    return offs, (off + align_max - 1) // align_max * align_max


def test_every_header_struct_is_bound_field_for_field(tmp_path):
    hs = header_structs()
    assert set(hs) == {"skinrt_config", "skinrt_channel_desc", "skinrt_clip_desc", "skinrt_morph_target_desc", "skinrt_rig_desc",
                       "skinrt_rig_info", "skinrt_placement_info", "skinrt_skin_desc", "skinrt_section_desc", "skinrt_skin_layout",
                       "skinrt_section_layout"}
    want = c_layout(hs, tmp_path)
    rs = rust_structs()
    for s, fields in hs.items():
        assert s in rs, f"{s} has no #[repr(C)] struct in skinrt-sys"
        assert [n for n, _ in rs[s]] == fields, f"{s}: field names / order differ from the header"
        offs, size = rust_layout(rs[s])
        assert size == want[s], f"{s}: size {size} (Rust) != {want[s]} (C)"
        for f in fields:
            assert offs[f] == want[f"{s}.{f}"], f"{s}.{f}: offset {offs[f]} (Rust) != {want[f'{s}.{f}']} (C)"
    # opaque handles
    for h in ("skinrt_ctx", "skinrt_rig", "skinrt_crowd"):
        assert h in rs and [t for _, t in rs[h]] == ["[u8; 0]"] or rs[h] == [("_private", "[u8; 0]")]


def test_ctypes_mirror_matches_every_field_too(tmp_path):
    hs = header_structs()
    want = c_layout(hs, tmp_path)
    mirror = {"skinrt_config": abi.Config, "skinrt_channel_desc": abi.ChannelDesc, "skinrt_clip_desc": abi.ClipDesc,
              "skinrt_morph_target_desc": abi.MorphTargetDesc, "skinrt_rig_desc": abi.RigDesc, "skinrt_rig_info": abi.RigInfo,
              "skinrt_placement_info": abi.PlacementInfo, "skinrt_skin_desc": abi.SkinDesc, "skinrt_section_desc": abi.SectionDesc,
              "skinrt_skin_layout": abi.SkinLayout, "skinrt_section_layout": abi.SectionLayout}
    for s, fields in hs.items():
        t = mirror[s]
        assert [n for n, _ in t._fields_] == fields, s
        assert C.sizeof(t) == want[s], s
        for f in fields:
            assert getattr(t, f).offset == want[f"{s}.{f}"], f"{s}.{f}"


def split_args(a):
    return [x for x in (p.strip() for p in a.split(",")) if x and x != "void"]


def test_extern_block_declares_the_headers_functions():
    h = {m.group(1): len(split_args(m.group(2)))
         for m in re.finditer(r"SKINRT_API\s+[\w\s\*]+?\b(skinrt_\w+)\s*\(([^)]*)\)\s*;", header_text(), flags=re.S)}
    text = re.sub(r"//.*", "", open(RUST).read())
    block = re.search(r'extern "C" \{(.*)\}', text, flags=re.S).group(1)
    r = {m.group(1): len(split_args(m.group(2))) for m in re.finditer(r"pub fn (skinrt_\w+)\s*\(([^)]*)\)", block, flags=re.S)}
    assert len(h) >= 32
    assert r == h


def test_rust_constants_equal_the_headers_defines():
    defs = {m.group(1): m.group(2) for m in re.finditer(r"#define\s+(SKINRT_\w+)\s+\(?(-?\d+)u?\)?\s", open(HEADER).read())}
    text = open(RUST).read()
    consts = {m.group(1): int(m.group(2).replace("_", "")) for m in re.finditer(r"pub const (SKINRT_\w+)\s*:\s*\w+\s*=\s*(-?[\d_]+)\s*;", text)}
    for name, val in consts.items():
        assert name in defs, name
        assert int(defs[name]) == val, name
    for name in ("SKINRT_ABI_VERSION", "SKINRT_PATH_WEIGHTS", "SKINRT_E_LIMIT", "SKINRT_STAGE_ALL", "SKINRT_SKIN_MATRIX_BYTES",
                 "SKINRT_TANGENT_NOALLOC_MAX_INDICES"):
        assert name in consts, name
    m = re.search(r'^version = "0\.(\d+)\.\d+"', open(os.path.join(ROOT, "crates", "skinrt-sys", "Cargo.toml")).read(), flags=re.M)
    assert int(m.group(1)) == int(defs["SKINRT_ABI_VERSION"])


def test_skin_matrix_mirror_in_the_safe_crate_is_112_bytes():
    text = open(os.path.join(ROOT, "crates", "skinrt", "src", "lib.rs")).read()
    m = re.search(r"pub struct SkinMatrix \{\s*pub transform: \[\[f32; 4\]; 4\],\s*pub normal: \[\[f32; 4\]; 3\],\s*\}", text)
    assert m, "SkinMatrix must mirror renderer/src/renderer/uniform/skin.rs:12-17"
    assert 4 * 4 * 4 + 3 * 4 * 4 == 112
