#!/usr/bin/env python3
"""Generate rust/srx_sys.rs — the `extern "C"` module a SingleRust maintainer adds — from
include/srx.h, so that the Rust declarations cannot drift from the C ABI.

The image has no rustc: the output is NOT compile-checked.  What is checked (tests/test_abi_cpu.py) is that the
committed file is exactly what this script produces from the current header, and that it declares every function
the shared library exports for include/srx.h.

    python scripts/gen_rust_bindings.py            # rewrite rust/srx_sys.rs
    python scripts/gen_rust_bindings.py --check    # exit 1 if the committed file is stale
"""
from __future__ import annotations

import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "srx.h")
OUT = os.path.join(ROOT, "rust", "srx_sys.rs")

SCALARS = {
    "int32_t": "i32", "uint32_t": "u32", "int64_t": "i64", "uint64_t": "u64", "size_t": "usize", "double": "f64",
    "float": "f32", "uint8_t": "u8", "int8_t": "i8", "uint16_t": "u16", "int16_t": "i16", "char": "c_char", "int": "i32",
}
RUST_KEYWORDS = {"type", "fn", "in", "ref", "mod", "use", "box", "loop", "match", "move", "self", "super", "where"}


def camel(name: str) -> str:
    return "".join(p.capitalize() for p in name.split("_"))


def strip_comments(src: str) -> str:
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"//[^\n]*", " ", src)
    src = re.sub(r"^\s*#.*$", " ", src, flags=re.M)            # preprocessor lines
    src = re.sub(r'extern\s+"C"\s*\{', " ", src)
    return src


class Types:
    def __init__(self):
        self.opaque: list[str] = []          # typedef struct X X;
        self.structs: dict[str, list[tuple[str, str]]] = {}
        self.enums: set[str] = set()
        self.fnptrs: dict[str, str] = {}

    def rust(self, ctype: str) -> str:
        """C type (no declarator name) -> Rust type."""
        t = " ".join(ctype.replace("*", " * ").split())
        n_ptr = t.count("*")
        base = t.replace("*", " ").split()
        const = "const" in base
        base = [w for w in base if w not in ("const", "struct", "enum")]
        assert len(base) == 1, ctype
        b = base[0]
        if b in self.fnptrs and n_ptr == 0:
            return camel(b)                                      # the `pub type` alias below
        if b == "void":
            r = "c_void"
            assert n_ptr >= 1, ctype
        elif b in SCALARS:
            r = SCALARS[b]
        elif b in self.enums:
            r = "i32"                                            # C enums cross the ABI as int
        elif b in self.structs or b in self.opaque:
            r = camel(b)
        else:
            raise ValueError(f"unknown C type {ctype!r}")
        for i in range(n_ptr):
            # `const T*` -> *const T; the const binds to the pointee of the innermost pointer only
            r = ("*const " if (const and i == 0) else "*mut ") + r
        return r


def split_decl(decl: str) -> tuple[str, str]:
    """'const uint64_t* sel' -> ('const uint64_t*', 'sel')"""
    m = re.match(r"^(.*?)([A-Za-z_][A-Za-z0-9_]*)\s*(\[[^\]]*\])?$", decl.strip(), flags=re.S)
    assert m, decl
    ctype, name, arr = m.group(1).strip(), m.group(2), m.group(3)
    assert ctype, decl
    return ctype + (f" [{arr[1:-1]}]" if arr else ""), name


def ident(name: str) -> str:
    name = name.rstrip("_") if name.endswith("_") and name[:-1] not in RUST_KEYWORDS else name
    return name + "_" if name in RUST_KEYWORDS else name


def parse(src: str):
    ty = Types()
    consts: list[tuple[str, str]] = []
    src = strip_comments(src)
    for m in re.finditer(r"#define\s+(SRX_[A-Z0-9_]+)\s+(\d+)", open(HEADER).read()):
        consts.append((m.group(1), m.group(2)))
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s+(\w+)\s*;", src):
        ty.opaque.append(m.group(2))
    # enums (named typedefs and anonymous groups of constants)
    for m in re.finditer(r"(?:typedef\s+)?enum\s*(\w*)\s*\{(.*?)\}\s*(\w*)\s*;", src, flags=re.S):
        if m.group(3):
            ty.enums.add(m.group(3))
        nxt = 0
        for item in m.group(2).split(","):
            item = item.strip()
            if not item:
                continue
            if "=" in item:
                k, v = (s.strip() for s in item.split("="))
                nxt = int(v, 0)
            else:
                k = item
            consts.append((k, str(nxt)))
            nxt += 1
    src_no_enum = re.sub(r"(?:typedef\s+)?enum\s*\w*\s*\{.*?\}\s*\w*\s*;", " ", src, flags=re.S)
    for m in re.finditer(r"typedef\s+(\w+)\s*\(\s*\*\s*(\w+)\s*\)\s*\((.*?)\)\s*;", src_no_enum, flags=re.S):
        ty.fnptrs[m.group(2)] = (m.group(1), m.group(3))             # resolved below (may use struct types)
    struct_src = list(re.finditer(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*(\w+)\s*;", src_no_enum, flags=re.S))
    for m in struct_src:
        ty.structs[m.group(3)] = []
    for name, (ret, params) in list(ty.fnptrs.items()):
        ps = ", ".join(f"{ident(n)}: {ty.rust(t)}" for t, n in (split_decl(p) for p in params.split(",")))
        ty.fnptrs[name] = f"extern \"C\" fn({ps}) -> {ty.rust(ret)}"
    for m in struct_src:
        fields = []
        for stmt in m.group(2).split(";"):
            stmt = stmt.strip()
            if not stmt:
                continue
            # `uint64_t a, b, c` -> three fields of the same type
            first, *rest = [s.strip() for s in stmt.split(",")]
            ctype, fname = split_decl(first)
            names = [fname] + rest
            for n in names:
                arr = re.search(r"\[(\w+)\]", ctype)
                if arr:
                    base = ty.rust(ctype[: arr.start()].strip())
                    fields.append((ident(n), f"[{base}; {arr.group(1)} as usize]"))
                else:
                    fields.append((ident(n), ty.rust(ctype)))
        ty.structs[m.group(3)] = fields
    body = re.sub(r"typedef\s+struct\s+\w+\s*\{.*?\}\s*\w+\s*;", " ", src_no_enum, flags=re.S)
    body = re.sub(r"typedef[^;]*;", " ", body)
    funcs = []
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(srx_\w+)\s*\(([^()]*)\)\s*;", body, flags=re.S):
        ret, name, params = " ".join(m.group(1).split()), m.group(2), " ".join(m.group(3).split())
        args = []
        if params and params != "void":
            for p in params.split(","):
                t, n = split_decl(p)
                args.append((ident(n), ty.rust(t)))
        funcs.append((name, args, None if ret == "void" else ty.rust(ret)))
    return ty, consts, funcs


def render() -> str:
    ty, consts, funcs = parse(open(HEADER).read())
    o = []
    o.append("// rust/srx_sys.rs — GENERATED from include/srx.h by scripts/gen_rust_bindings.py; do not edit.")
    o.append("// `extern \"C\"` declarations of libsrx_hip.so for the shim of INTEGRATION.md (src/gpu/ffi.rs in a SingleRust")
    o.append("// checkout, behind a cargo feature).  Not compile-checked here: the build image has no rustc.  The hand-written")
    o.append("// part of the binding (the `upload` helper and the replaced function bodies) is rust/shim.rs.")
    o.append("#![allow(non_camel_case_types, dead_code)]")
    o.append("use std::os::raw::{c_char, c_void};")
    o.append("")
    for k, v in consts:
        t = "usize" if k in ("SRX_UNIQUE_ID_BYTES",) else "i32"
        o.append(f"pub const {k}: {t} = {v};")
    o.append("")
    for name in ty.opaque:
        if name in ty.structs:
            continue
        o.append(f"#[repr(C)] pub struct {camel(name)} {{ _private: [u8; 0] }}      // opaque `{name}`")
    o.append("")
    for name, fields in ty.structs.items():
        o.append("#[repr(C)]")
        o.append("#[derive(Clone, Copy)]")
        o.append(f"pub struct {camel(name)} {{      // `{name}`")
        for fname, ftype in fields:
            o.append(f"    pub {fname}: {ftype},")
        o.append("}")
        o.append("")
    for name, sig in ty.fnptrs.items():
        o.append(f"pub type {camel(name)} = {sig};      // `{name}`")
    o.append("")
    o.append('#[link(name = "srx_hip")]')
    o.append('extern "C" {')
    for name, args, ret in funcs:
        a = ", ".join(f"{n}: {t}" for n, t in args)
        line = f"    pub fn {name}({a})" + (f" -> {ret}" if ret else "") + ";"
        if len(line) > 118:
            pad = " " * (len(f"    pub fn {name}("))
            chunks, cur = [], ""
            for i, (n, t) in enumerate(args):
                piece = f"{n}: {t}" + (", " if i + 1 < len(args) else "")
                if len(pad) + len(cur) + len(piece) > 116 and cur:
                    chunks.append(cur.rstrip())
                    cur = ""
                cur += piece
            chunks.append(cur)
            line = f"    pub fn {name}(" + ("\n" + pad).join(chunks) + ")" + (f" -> {ret}" if ret else "") + ";"
        o.append(line)
    o.append("}")
    o.append("")
    return "\n".join(o)


def declared_functions() -> list[str]:
    return [f[0] for f in parse(open(HEADER).read())[2]]


if __name__ == "__main__":
    text = render()
    if "--check" in sys.argv:
        ok = os.path.exists(OUT) and open(OUT).read() == text
        print("rust/srx_sys.rs is", "up to date" if ok else "STALE: run scripts/gen_rust_bindings.py")
        sys.exit(0 if ok else 1)
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open(OUT, "w") as f:
        f.write(text)
    print(f"wrote {OUT}: {text.count('pub fn ')} functions")
