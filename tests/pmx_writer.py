"""Test helper: serialise a small skinned model as a PMX 2.0 file, field by field in the layout the
reference's reader declares (renderer-asset/src/loader/pmx/format.rs:159-780).  Written from that
layout; it is not a general exporter."""
import struct

import numpy as np


def _text(s, utf16):
    b = s.encode("utf-16-le" if utf16 else "utf-8")
    return struct.pack("<I", len(b)) + b


def _idx(v, width):
    return struct.pack({1: "<b", 2: "<h", 4: "<i"}[width], -1 if v is None else int(v))


def write_pmx(positions, normals, indices, bones, deforms, morphs=(), materials=None, utf16=True, extra_vec4=0,
              widths=(2, 1, 1, 2, 1, 1), globals_count=8, version=2.0, other_morphs=()):
    """bones: [(position[3], parent or None, flags dict)]; deforms: per vertex ("bdef1", b) | ("bdef2", b1, b2, w) |
    ("bdef4", [b]*4, [w]*4) | ("sdef", b1, b2, w) | ("qdef", [b]*4, [w]*4); morphs: [[(vertex, (dx, dy, dz)), ...]];
    widths: vertex, texture, material, bone, morph, rigidbody index bytes."""
    wv, wt, wm, wb, wmo, wr = widths
    out = bytearray(b"PMX ") + struct.pack("<f", version)
    out += struct.pack("<bBb", globals_count, 0 if utf16 else 1, extra_vec4) + bytes(widths) + b"\0" * (globals_count - 8)
    for s in ("model", "model", "comment", "comment"):
        out += _text(s, utf16)
    V = len(positions)
    out += struct.pack("<i", V)
    for v in range(V):
        out += struct.pack("<3f3f2f", *positions[v], *normals[v], 0.25, 0.75) + b"\0" * (16 * extra_vec4)
        d = deforms[v]
        kind = d[0]
        if kind == "bdef1":
            out += b"\0" + _idx(d[1], wb)
        elif kind in ("bdef2", "sdef"):
            out += (b"\1" if kind == "bdef2" else b"\3") + _idx(d[1], wb) + _idx(d[2], wb) + struct.pack("<f", d[3])
            if kind == "sdef":
                out += struct.pack("<9f", *range(9))
        else:
            out += (b"\2" if kind == "bdef4" else b"\4") + b"".join(_idx(b, wb) for b in d[1]) + struct.pack("<4f", *d[2])
        out += struct.pack("<f", 1.0)
    out += struct.pack("<i", len(indices)) + b"".join(_idx(i, wv) for i in indices)
    out += struct.pack("<i", 1) + _text("tex.png", utf16)
    if materials is None:
        materials = [len(indices)]
    out += struct.pack("<i", len(materials))
    for k, sc in enumerate(materials):
        out += _text(f"m{k}", utf16) + _text(f"m{k}", utf16)
        out += struct.pack("<4f3ff3fB4ff", 1, 1, 1, 1, 0, 0, 0, 5, 0.2, 0.2, 0.2, 0x1F, 0, 0, 0, 1, 1.0)
        out += _idx(0, wt) + _idx(None, wt) + b"\0"
        out += (b"\1" + b"\3") if k % 2 else (b"\0" + _idx(None, wt))
        out += _text("meta", utf16) + struct.pack("<i", sc)
    out += struct.pack("<i", len(bones))
    for k, (pos, parent, fl) in enumerate(bones):
        flags = 0x001E
        body = bytearray()
        if fl.get("indexed_tail"):
            flags |= 0x0001; body += _idx(None, wb)
        else:
            body += struct.pack("<3f", 0, 1, 0)
        if fl.get("inherit"):
            flags |= 0x0100; body += _idx(0, wb) + struct.pack("<f", 0.5)
        if fl.get("fixed_axis"):
            flags |= 0x0400; body += struct.pack("<3f", 1, 0, 0)
        if fl.get("local"):
            flags |= 0x0800; body += struct.pack("<6f", 1, 0, 0, 0, 0, 1)
        if fl.get("external"):
            flags |= 0x2000; body += _idx(0, wb)
        if fl.get("ik"):
            flags |= 0x0020
            body += _idx(0, wb) + struct.pack("<ifi", 8, 0.5, 2) + _idx(0, wb) + b"\0" + _idx(0, wb) + b"\1" + struct.pack("<6f", *[0] * 6)
        out += _text(f"b{k}", utf16) + _text(f"b{k}", utf16) + struct.pack("<3f", *pos) + _idx(parent, wb) + struct.pack("<iH", 0, flags) + body
    out += struct.pack("<i", len(morphs) + len(other_morphs))
    # other morph kinds first and in between: the loader has to step over them exactly
    for kind in other_morphs:
        out += _text("o", utf16) + _text("o", utf16) + b"\4"
        if kind == "group":
            out += b"\0" + struct.pack("<i", 2) + (_idx(0, wmo) + struct.pack("<f", 0.5)) * 2
        elif kind == "bone":
            out += b"\2" + struct.pack("<i", 1) + _idx(0, wb) + struct.pack("<6f", *[0.1] * 6)
        elif kind == "uv":
            out += b"\3" + struct.pack("<i", 3) + (_idx(0, wv) + struct.pack("<4f", *[0.1] * 4)) * 3
        elif kind == "material":
            out += b"\10" + struct.pack("<i", 1) + _idx(0, wm) + b"\0" + struct.pack("<28f", *[0.5] * 28)
        elif kind == "impulse":
            out += b"\12" + struct.pack("<i", 1) + _idx(0, wr) + b"\1" + struct.pack("<6f", *[0.5] * 6)
        elif kind == "flip":
            out += b"\11" + struct.pack("<i", 1) + _idx(0, wmo) + struct.pack("<f", 1.0)
    for m in morphs:
        out += _text("v", utf16) + _text("v", utf16) + b"\1\1" + struct.pack("<i", len(m))
        for v, d in m:
            out += _idx(v, wv) + struct.pack("<3f", *d)
    out += struct.pack("<iii", 0, 0, 0)  # display frames, rigid bodies, joints: none (and not read)
    return bytes(out)
