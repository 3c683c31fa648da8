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


FILE_RS = os.path.join(ROOT, "bindings", "rust", "src", "file.rs")


def _impl_methods(src, type_name):
    """names of the `pub fn` / `pub async fn` of every `impl ... type_name ...` block of a Rust source"""
    out = []
    for m in re.finditer(r"impl(?:<[^>]*>)?\s+" + type_name + r"(?:<[^>]*>)?\s*\{", src):
        i, depth = m.end(), 1
        while depth and i < len(src):
            depth += {"{": 1, "}": -1}.get(src[i], 0)
            i += 1
        out += re.findall(r"pub\s+(?:async\s+)?fn\s+(\w+)", src[m.end():i])
    return out


def test_path_taking_builders_keep_the_reference_surface():
    """bindings/rust/src/file.rs: IndexBuilder::new(source, column) ... build_inplace() / build_new(out) and
    TopkBuilder::new(path, &query) ... search() -- every public method of the reference's two builders
    (src/ivf/parquet.rs:23-103, src/ivf/search.rs:49-81) exists under the same name, with the same validation texts; every
    sys::pqv_* call in the file is declared with that arity."""
    src = open(FILE_RS).read()
    ours_ib, ours_tb = set(_impl_methods(src, "IndexBuilder")), set(_impl_methods(src, "TopkBuilder"))
    want_ib = {"new", "n_clusters", "max_iters", "seed", "build_inplace", "build_new"}
    want_tb = {"new", "k", "nprobe", "search"}
    ref_parquet = "/root/reference/src/ivf/parquet.rs"
    ref_search = "/root/reference/src/ivf/search.rs"
    if os.path.exists(ref_parquet) and os.path.exists(ref_search):         # (this container only; the GPU box has no reference)
        assert set(_impl_methods(open(ref_parquet).read(), "IndexBuilder")) == want_ib
        assert set(_impl_methods(open(ref_search).read(), "TopkBuilder")) == want_tb
    assert want_ib <= ours_ib, want_ib - ours_ib
    assert want_tb <= ours_tb, want_tb - ours_tb
    assert re.search(r"pub fn new\(source: impl AsRef<Path>, embedding_column: impl AsRef<str>\) -> Self", src)
    assert re.search(r"pub fn new\(parquet_path: impl AsRef<Path>, query: &'a \[f32\]\) -> Self", src)
    assert "pub async fn search(self)" in src
    for needle in ('"k must be > 0"', '"nprobe must be > 0"', '"k must be set"', '"nprobe must be set"',
                   '"Missing pq-vector index metadata in parquet footer"', '"Invalid pq-vector index magic"',
                   '"pq-vector index payload is truncated"', '"pq-vector index bytes are truncated"',
                   '"Parquet file too small to contain a footer"', '"Encrypted parquet footers are not supported for in-place indexing"',
                   '"Embedding column contains null rows"', '"Embedding values contain nulls"', '"Embedding row has zero length"',
                   '"Embedding vectors have inconsistent dimensions"', '"Embedding values are not float32/float64"',
                   'b"PQ_VECTOR1"', '"pq_vector_index_offset"', '"pq_vector_embedding_column"', "shard_row_groups"):
        assert needle in src, needle
    rs = _rust_decls()
    calls = 0
    for m in re.finditer(r"sys::(pqv_\w+)\(", src):
        name = m.group(1)
        assert name in rs, f"file.rs calls undeclared {name}"
        i, depth = m.end(), 1
        while depth:
            depth += {"(": 1, ")": -1}.get(src[i], 0)
            i += 1
        assert len(_split_args(src[m.end():i - 1])) == len(rs[name][0]), name
        calls += 1
    assert calls >= 3
    lib = open(LIB_RS).read()
    assert "pub mod file;" in lib and 'feature = "parquet-files"' in lib
    # the constants the file uses exist in the generated sys.rs
    sysrs = open(os.path.join(ROOT, "bindings", "rust", "src", "sys.rs")).read()
    for c in ("PQV_RELEASE_IF_COPIED", "PQV_LAYOUT_IVF_ORDERED", "PQV_L2SQ_REF4"):
        assert f"pub const {c}" in sysrs


def test_rust_row_group_shard_rule_matches_python():
    """file.rs::shard_row_groups restates sharding.shard_row_groups: same cut rule (checked textually: nearest boundary, lower on
    a tie, monotone) -- and the Python rule's outputs on a few layouts are pinned here so a change to either side shows."""
    from pq_vector_amd.sharding import shard_row_groups
    assert [shard_row_groups(r, 2, [1000] * 10 + [37]) for r in range(2)] == [(0, 5, 0, 5000), (5, 11, 5000, 5037)]
    assert [shard_row_groups(r, 3, [5, 1, 1, 1, 100]) for r in range(3)] == [(0, 4, 0, 8), (4, 5, 8, 100), (5, 5, 108, 0)]
    assert shard_row_groups(0, 1, []) == (0, 0, 0, 0)
    src = open(FILE_RS).read()
    assert "sys::pqv_shard_row_groups(" in src                      # the Rust shim asks the library: one implementation
    # ... which must agree with the rule restated in plain Python on random layouts (ties, empty ranges, huge row groups)
    import random
    from pq_vector_amd.sharding import _shard_row_groups_py
    rnd = random.Random(7)
    for _ in range(300):
        n_rg = rnd.randint(0, 40)
        rows = [rnd.choice([0, 1, 5, 1000, 1000, 4096, rnd.randint(1, 10**7), 2**40]) for _ in range(n_rg)]
        world = rnd.randint(1, 12)
        ranges = [shard_row_groups(r, world, rows) for r in range(world)]
        assert ranges == [_shard_row_groups_py(r, world, rows) for r in range(world)], (rows, world)
        assert ranges[0][0] == 0 and ranges[-1][1] == n_rg and all(ranges[i][1] == ranges[i + 1][0] for i in range(world - 1))
        assert sum(x[3] for x in ranges) == sum(rows)
