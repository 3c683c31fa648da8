"""The Rust shim (bindings/rust) cannot be compiled here (no cargo / rustc in the image), so it is held to the C header
mechanically: the raw extern block must declare exactly the header's symbols with the header's arity and pointer
shapes, it must equal what tools/gen_rust_sys.py generates today, and every sys:: call in the safe wrappers must name a
declared symbol and pass the declared number of arguments."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_rust_sys  # noqa: E402

SYS_RS = os.path.join(ROOT, "bindings", "rust", "src", "sys.rs")
LIB_RS = os.path.join(ROOT, "bindings", "rust", "src", "lib.rs")


def _rust_decls():
    out = {}
    for m in re.finditer(r"pub fn (pqv_\w+)\((.*?)\)( -> ([^;]+))?;", open(SYS_RS).read()):
        args = [a.strip() for a in m.group(2).split(",") if a.strip()]
        out[m.group(1)] = ([a.split(":", 1)[1].strip() for a in args], (m.group(4) or "").strip())
    return out


def test_sys_rs_is_the_generated_file():
    assert open(SYS_RS).read() == gen_rust_sys.generate(), "run tools/gen_rust_sys.py"


def test_extern_block_matches_header_symbol_for_symbol():
    hdr = {name: (ret, params) for name, ret, params in gen_rust_sys.parse_header()}
    rs = _rust_decls()
    assert set(hdr) == set(rs)
    from pq_vector_amd import _ffi
    assert set(hdr) == set(_ffi.SIGNATURES), "ctypes table, header and Rust block must list the same symbols"
    for name, (ret, params) in hdr.items():
        rtypes, rret = rs[name]
        assert len(rtypes) == len(params), name
        for (ctype, pname), rtype in zip(params, rtypes):
            assert rtype.count("*") == ctype.count("*"), (name, pname)
            if ctype.count("*") == 1:
                assert rtype.startswith("*const") == ctype.startswith("const"), (name, pname, ctype, rtype)
        assert (rret == "") == (ret == "void"), name
        assert len(_ffi.SIGNATURES[name][1]) == len(params), name


def _split_args(s):
    depth, cur, out = 0, "", []
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur); cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return out


def test_safe_wrappers_call_declared_symbols_with_declared_arity():
    rs = _rust_decls()
    src = open(LIB_RS).read()
    calls = 0
    for m in re.finditer(r"sys::(pqv_\w+)\(", src):
        name = m.group(1)
        assert name in rs, f"lib.rs calls undeclared {name}"
        i, depth = m.end(), 1
        while depth:
            depth += {"(": 1, ")": -1}.get(src[i], 0)
            i += 1
        args = _split_args(src[m.end():i - 1])
        assert len(args) == len(rs[name][0]), (name, len(args), len(rs[name][0]))
        calls += 1
    assert calls >= 25
    # the reference's builder surface is present with its validation texts
    for needle in ("pub struct IndexBuilder", "pub struct TopkBuilder", "pub struct SearchResult", "impl Drop for Searcher",
                   "impl Drop for Corpus", "impl Drop for Index", '"k must be > 0"', '"nprobe must be > 0"', '"k must be set"',
                   '"nprobe must be set"', '"max_iters must be > 0"', '"n_clusters must be > 0"'):
        assert needle in src, needle
