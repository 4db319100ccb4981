"""integration/rust/bullet-mi355/src/lib.rs cannot be compiled here (no Rust toolchain), so its declarations are
checked textually against include/bullet_amd.h: every #[repr(C)] struct has the header's fields in the header's order,
and every extern fn names a declared entry point with the same number of parameters."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def strip_comments(text, rust=False):
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return re.sub(r"//.*", "", text) if rust else text


def c_structs(hdr):
    out = {}
    for body, name in re.findall(r"typedef struct \{(.*?)\}\s*(\w+);", hdr, flags=re.S):
        fields = []
        for decl in body.split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            fp = re.match(r".*\(\*\s*(\w+)\)\s*\(", decl)  # function pointer member
            if fp:
                fields.append(fp.group(1))
                continue
            decl = re.sub(r"^(const\s+)?(unsigned\s+)?\w+\s+", "", decl, count=1)  # drop the type
            for part in decl.split(","):
                fields.append(re.sub(r"\[.*", "", part.replace("*", "").strip()))
        out[name] = fields
    return out


def c_functions(hdr):
    hdr = re.sub(r"typedef[^;{]*\{.*?\}[^;]*;", "", hdr, flags=re.S)
    out = {}
    for name, args in re.findall(r"\b(bt_\w+)\s*\(([^;{]*?)\)\s*;", hdr, flags=re.S):
        if "(*" in name:
            continue
        args = args.strip()
        out[name] = 0 if args in ("", "void") else len(split_top(args))
    return out


def split_top(args):
    parts, depth, cur = [], 0, ""
    for ch in args:
        depth += ch in "(<[" 
        depth -= ch in ")>]"
        if ch == "," and depth == 0:
            parts.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        parts.append(cur)
    return parts


def test_rust_declarations_match_the_header():
    hdr = strip_comments(open(os.path.join(ROOT, "include", "bullet_amd.h")).read())
    rs = strip_comments(open(os.path.join(ROOT, "integration", "rust", "bullet-mi355", "src", "lib.rs")).read(), rust=True)
    ffi = rs[rs.index("pub mod ffi"):rs.index("pub struct Error")]
    cs, cf = c_structs(hdr), c_functions(hdr)
    seen = 0
    for name, body in re.findall(r"#\[repr\(C\)\][^{]*?pub struct (\w+) \{(.*?)\n    \}", ffi, flags=re.S):
        if name in ("bt_net", "bt_reader"):  # opaque
            continue
        fields = re.findall(r"pub (\w+)\s*:", body)
        assert name in cs, name
        assert fields == cs[name], (name, fields, cs[name])
        seen += 1
    assert seen >= 8
    ext = ffi[ffi.index('extern "C" {'):]
    fns = re.findall(r"pub fn (bt_\w+)\s*\((.*?)\)\s*(?:->[^;]*)?;", ext, flags=re.S)
    assert len(fns) >= 35
    for name, args in fns:
        assert name in cf, f"{name} is not declared in bullet_amd.h"
        n = len([a for a in split_top(args) if a.strip()])
        assert n == cf[name], (name, n, cf[name])
