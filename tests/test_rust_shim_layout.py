"""The Rust shim (rust/border-amd-agent/src/ffi.rs) against include/border_amd.h, textually: every `#[repr(C)]` struct must have the
header's fields in the header's order with the header's types, every `extern "C"` function the header's arguments and return
type, every callback typedef the header's signature - and the struct sizes / field offsets computed from the Rust declarations
(repr(C) = the C layout rules) must equal what gcc computes from the header.  There is no cargo in this image, so this is the
guard that keeps shim and ABI from drifting apart (round 2's INTEGRATION.md snippet built a 40-byte `bdr_replay_config` for a
48-byte ABI; this test fails on that)."""
import os
import re
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "border_amd.h")
FFI = os.path.join(ROOT, "rust", "border-amd-agent", "src", "ffi.rs")

PRIM = {"uint64_t": "u64", "int64_t": "i64", "uint32_t": "u32", "int32_t": "i32", "uint16_t": "u16", "int16_t": "i16",
        "uint8_t": "u8", "int8_t": "i8", "float": "f32", "double": "f64", "char": "c_char", "void": "c_void", "size_t": "usize"}
SIZE = {"u64": 8, "i64": 8, "u32": 4, "i32": 4, "u16": 2, "i16": 2, "u8": 1, "i8": 1, "f32": 4, "f64": 8, "c_char": 1, "usize": 8}


# ------------------------------------------------------------------------------------------------ C side
def strip_c_comments(s):
    s = re.sub(r"/\*.*?\*/", " ", s, flags=re.S)
    return re.sub(r"//[^\n]*", " ", s)


def c_defines(src):
    return {m.group(1): int(m.group(2)) for m in re.finditer(r"#define\s+(\w+)\s+(\d+)\s*$", src, flags=re.M)}


def c_type(tokens, stars):
    """tokens: type words (may contain 'const'); stars: pointer depth -> canonical Rust-style spelling."""
    const = "const" in tokens
    words = [t for t in tokens if t not in ("const", "struct")]
    assert len(words) == 1, tokens
    t = PRIM.get(words[0], words[0])
    for k in range(stars):
        t = ("*const " if const else "*mut ") + t
        const = False if k == 0 else const   # qualifiers written before the base type apply to the pointee of the first star
    return t


def split_top(s, sep=","):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([<{":
            depth += 1
        elif ch in ")]>}" and not (ch == ">" and cur.endswith("-")):   # `->` is not a bracket
            depth -= 1
        if ch == sep and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return [x.strip() for x in out]


def c_param(p, defs):
    """one C parameter / simple declarator -> (name, canonical type)"""
    p = p.strip()
    m = re.match(r"^(.*?)\(\s*\*\s*(\w*)\s*\)\s*\((.*)\)$", p, flags=re.S)
    if m:   # function pointer
        ret = canon(c_param(m.group(1).strip() + " _", defs)[1]) if m.group(1).strip() != "void" else "()"
        params = [] if m.group(3).strip() in ("", "void") else [canon(c_param(q, defs)[1]) for q in split_top(m.group(3))]
        return m.group(2), "fn(" + ", ".join(params) + ") -> " + ret
    arr = re.search(r"\[\s*(\w+)\s*\]\s*$", p)
    n_arr = None
    if arr:
        n_arr = int(arr.group(1)) if arr.group(1).isdigit() else defs[arr.group(1)]
        p = p[:arr.start()].strip()
    stars = p.count("*")
    words = p.replace("*", " ").split()
    name, tokens = words[-1], words[:-1]
    t = c_type(tokens, stars)
    return name, (t, n_arr)


def canon(tn, as_param=False):
    """(type, array) -> spelling; an array PARAMETER decays to a pointer"""
    if isinstance(tn, str):
        return tn
    t, n = tn
    if n is None:
        return t
    if as_param:
        raise AssertionError("use c_array_param")
    return f"[{t}; {n}]"


def parse_c(src):
    defs = c_defines(src)
    s = strip_c_comments(src)
    structs, funcs, fnptr_types = {}, {}, {}
    for m in re.finditer(r"typedef\s+struct\s*(\w*)\s*\{(.*?)\}\s*(\w+)\s*;", s, flags=re.S):
        fields = []
        for decl in [d.strip() for d in m.group(2).split(";") if d.strip()]:
            if "(" in decl:
                name, t = c_param(decl, defs)
                fields.append((name, t))
                continue
            # `type a, b[8], c;`
            first, *rest = split_top(decl)
            n0, t0 = c_param(first, defs)
            fields.append((n0, canon(t0)))
            base = first.replace("*", " ").split()[:-1] if "[" not in first else re.sub(r"\[.*\]", "", first).replace("*", " ").split()[:-1]
            for r in rest:
                nm, t = c_param(" ".join(base) + " " + r, defs)
                fields.append((nm, canon(t)))
        structs[m.group(3)] = fields
    for m in re.finditer(r"typedef\s+(\w[\w\s\*]*?)\(\s*\*\s*(\w+)\s*\)\s*\((.*?)\)\s*;", s, flags=re.S):
        ret = "()" if m.group(1).strip() == "void" else canon(c_param(m.group(1).strip() + " _", defs)[1])
        params = [canon(c_param(q, defs)[1]) for q in split_top(m.group(3))]
        fnptr_types[m.group(2)] = "fn(" + ", ".join(params) + ") -> " + ret
    for m in re.finditer(r"BDR_API\s+([\w\s\*]+?)\b(bdr_\w+)\s*\((.*?)\)\s*;", s, flags=re.S):
        rt = m.group(1).strip()
        ret = "()" if rt == "void" else canon(c_param(rt + " _", defs)[1])
        params = []
        if m.group(3).strip() not in ("", "void"):
            for q in split_top(m.group(3)):
                nm, t = c_param(q, defs)
                if isinstance(t, tuple) and t[1] is not None:     # array parameter: decays to a pointer to its element
                    const = "const" in q.split()
                    t = ("*const " if const else "*mut ") + t[0]
                params.append(canon(t))
        funcs[m.group(2)] = (params, ret)
    return structs, funcs, fnptr_types, defs


# ------------------------------------------------------------------------------------------------ Rust side
def strip_rs_comments(s):
    return re.sub(r"//[^\n]*", " ", s)


def rs_type(t):
    t = " ".join(t.split())
    m = re.match(r'^Option<\s*unsafe extern "C" fn\((.*)\)\s*(?:->\s*(.+?))?\s*,?\s*>$', t, flags=re.S)
    if m:
        params = []
        for p in split_top(m.group(1)):
            if p:
                params.append(rs_type(p.split(":", 1)[1]))
        return "fn(" + ", ".join(params) + ") -> " + (rs_type(m.group(2)) if m.group(2) else "()")
    return t


def parse_rs(src):
    s = strip_rs_comments(src)
    structs, funcs, fnptr_types, consts = {}, {}, {}, {}
    for m in re.finditer(r"#\[repr\(C\)\]\s*(?:#\[derive\([^\]]*\)\]\s*)?pub struct (\w+)\s*\{(.*?)\n\}", s, flags=re.S):
        fields = []
        for f in split_top(m.group(2)):
            if not f:
                continue
            name, t = f.split(":", 1)
            fields.append((re.sub(r"^\s*pub\s+", "", name).strip(), rs_type(t)))
        structs[m.group(1)] = fields
    for m in re.finditer(r"pub type (\w+)\s*=\s*(.*?);", s, flags=re.S):
        fnptr_types[m.group(1)] = rs_type(m.group(2))
    ext = re.search(r'extern "C" \{(.*)\n\}', s, flags=re.S).group(1)
    for m in re.finditer(r"pub fn (\w+)\s*\((.*?)\)\s*(?:->\s*([^;]+?))?\s*;", ext, flags=re.S):
        params = [rs_type(p.split(":", 1)[1]) for p in split_top(m.group(2)) if p]
        funcs[m.group(1)] = (params, rs_type(m.group(3)) if m.group(3) else "()")
    for m in re.finditer(r"pub const (\w+): \w+ = (\d+);", s):
        consts[m.group(1)] = int(m.group(2))
    return structs, funcs, fnptr_types, consts


# ------------------------------------------------------------------------------------------------ fixtures
@pytest.fixture(scope="module")
def both():
    return parse_c(open(HEADER).read()), parse_rs(open(FFI).read())


OPAQUE = {"bdr_replay", "bdr_agent", "bdr_comm", "bdr_model_mailbox", "bdr_atari_prep"}


def test_every_struct_of_the_header_is_declared_with_the_same_fields(both):
    (cs, _, _, _), (rs, _, _, _) = both
    assert len(cs) >= 20, sorted(cs)
    missing = sorted(set(cs) - set(rs))
    assert not missing, f"structs of border_amd.h without a #[repr(C)] twin in ffi.rs: {missing}"
    extra = sorted(set(rs) - set(cs) - OPAQUE)
    assert not extra, f"structs in ffi.rs the header does not declare: {extra}"
    for name, cf in cs.items():
        rf = rs[name]
        assert [n for n, _ in rf] == [n for n, _ in cf], f"{name}: field names / order differ:\n  header {[n for n, _ in cf]}\n  ffi.rs {[n for n, _ in rf]}"
        for (n, ct), (_, rt) in zip(cf, rf):
            assert canon(ct) == rt, f"{name}.{n}: header `{canon(ct)}` vs ffi.rs `{rt}`"


def test_every_function_of_the_header_is_declared_with_the_same_signature(both):
    (_, cf, _, _), (_, rf, _, _) = both
    assert len(cf) >= 90, len(cf)
    missing = sorted(set(cf) - set(rf))
    assert not missing, f"functions of border_amd.h missing from ffi.rs: {missing}"
    extra = sorted(set(rf) - set(cf))
    assert not extra, f"functions in ffi.rs the header does not declare: {extra}"
    for name, (cp, cr) in cf.items():
        rp, rr = rf[name]
        assert cp == rp, f"{name}: parameters differ:\n  header {cp}\n  ffi.rs {rp}"
        assert cr == rr, f"{name}: return type: header `{cr}` vs ffi.rs `{rr}`"


def test_callback_typedefs_and_constants_match(both):
    (_, _, ct, defs), (_, _, rt, consts) = both
    assert set(ct) == {"bdr_trainer_observer", "bdr_async_observer"} and set(ct) <= set(rt)
    for k in ct:
        assert ct[k] == rt[k], (k, ct[k], rt[k])
    for k, v in defs.items():
        if k in consts:
            assert consts[k] == v, (k, v, consts[k])
    for k in ("BDR_MAX_UNITS", "BDR_UNIQUE_ID_BYTES", "BDR_CKPT_SAFETENSORS", "BDR_ASYNC_EVENT_ACTOR_SYNC", "BDR_TRAINER_EVENT_COST"):
        assert k in consts, k
    # enum values of the header
    hdr = strip_c_comments(open(HEADER).read())
    enum_vals = {}
    for body in re.findall(r"enum\s*\{(.*?)\}", hdr, flags=re.S):
        nxt = 0
        for item in [x.strip() for x in body.split(",") if x.strip()]:
            if "=" in item:
                nm, v = [x.strip() for x in item.split("=")]
                nxt = int(v)
            else:
                nm = item
            enum_vals[nm] = nxt
            nxt += 1
    assert len(enum_vals) >= 24
    for k, v in enum_vals.items():
        assert consts.get(k) == v, f"enum {k} = {v} in the header, {consts.get(k)} in ffi.rs"


def _layout(fields, structs, cache):
    """C layout rules on canonical field types -> (size, align, [offsets])"""
    off, align, offs = 0, 1, []
    for _, t in fields:
        sz, al = _size_align(t, structs, cache)
        off = (off + al - 1) // al * al
        offs.append(off)
        off += sz
        align = max(align, al)
    return (off + align - 1) // align * align, align, offs


def _size_align(t, structs, cache):
    if t.startswith("*") or t.startswith("fn("):
        return 8, 8
    m = re.match(r"^\[(.+); (\d+)\]$", t)
    if m:
        s, a = _size_align(m.group(1), structs, cache)
        return s * int(m.group(2)), a
    if t in SIZE:
        return SIZE[t], SIZE[t]
    if t not in cache:
        cache[t] = _layout(structs[t], structs, cache)[:2]
    return cache[t]


def test_sizes_and_offsets_from_the_rust_declarations_equal_gccs(both):
    (cs, _, _, _), (rs, _, _, _) = both
    names = sorted(cs)
    prog = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{HEADER}"', "int main(void) {"]
    for n in names:
        prog.append(f'  printf("{n} %zu", sizeof({n}));')
        for f, _ in cs[n]:
            prog.append(f'  printf(" %zu", offsetof({n}, {f}));')
        prog.append('  printf("\\n");')
    prog += ["  return 0;", "}"]
    with tempfile.TemporaryDirectory() as d:
        src, exe = os.path.join(d, "l.c"), os.path.join(d, "l")
        open(src, "w").write("\n".join(prog))
        subprocess.check_call(["gcc", "-std=c11", "-o", exe, src])
        out = subprocess.check_output([exe], text=True)
    cache = {}
    for line in out.splitlines():
        n, size, *offs = line.split()
        r_size, _, r_offs = _layout(rs[n], rs, cache)
        assert r_size == int(size), f"{n}: {r_size} bytes from ffi.rs, {size} from the header"
        assert r_offs == [int(o) for o in offs], f"{n}: field offsets {r_offs} from ffi.rs, {offs} from the header"
    assert int(dict((l.split()[0], l.split()[1]) for l in out.splitlines())["bdr_replay_config"]) == 56


def test_the_crate_has_complete_sources_and_integration_md_points_at_them():
    crate = os.path.join(ROOT, "rust", "border-amd-agent")
    for f in ("Cargo.toml", "build.rs", "src/lib.rs", "src/ffi.rs", "src/error.rs", "src/bytes.rs", "src/config.rs", "src/handle.rs", "src/replay.rs",
              "src/dqn.rs", "src/iqn.rs", "src/sac.rs", "src/async_trainer.rs", "src/comm.rs"):
        p = os.path.join(crate, f)
        assert os.path.exists(p), f
        body = open(p).read()
        assert "/* ... */" not in body and "todo!()" not in body and "unimplemented!()" not in body.replace("_ => unimplemented!(), // iqn/model/config.rs", ""), f
    # every trait the boundary names is implemented for every agent kind
    for f, agent in (("dqn.rs", "AmdDqn"), ("iqn.rs", "AmdIqn"), ("sac.rs", "AmdSac")):
        body = open(os.path.join(crate, "src", f)).read()
        for trait in ("Policy<E>", "Configurable", "Agent<E, AmdReplayBuffer<O, A>>", "SyncModel"):
            assert re.search(rf"impl<E, O, A> {re.escape(trait)} for {agent}<E, O, A>", body), (f, trait)
        for method in ("fn train(", "fn eval(", "fn is_train(", "fn opt(", "fn opt_with_record(", "fn save_params(", "fn load_params(", "fn as_any_ref(",
                       "fn as_any_mut(", "fn sample(", "fn build(", "fn model_info(", "fn sync_model("):
            assert method in body, (f, method)
    rb = open(os.path.join(crate, "src", "replay.rs")).read()
    for s in ("impl<O, A> ExperienceBufferBase for AmdReplayBuffer<O, A>", "impl<O, A> ReplayBufferBase for AmdReplayBuffer<O, A>", "fn push(", "fn len(",
              "fn build(", "fn batch(", "fn update_priority("):
        assert s in rb, s
    assert "bdr_async_train(" in open(os.path.join(crate, "src", "async_trainer.rs")).read()
    integ = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    assert "rust/border-amd-agent/src/ffi.rs" in integ and "rust/border-amd-agent/src/async_trainer.rs" in integ
    # no Rust struct literal of the ABI survives in the document (it went stale once): the crate is the source
    assert "reserved: 0 };" not in integ and "device: 0, reserved: 0" not in integ


def _rust_code_only(src: str) -> str:
    """Rust source with comments, string / char literals removed (lifetimes kept): enough for delimiter accounting."""
    out, i, n = [], 0, len(src)
    while i < n:
        c = src[i]
        if src.startswith("//", i):
            i = src.find("\n", i) if src.find("\n", i) >= 0 else n
        elif src.startswith("/*", i):
            depth, i = 1, i + 2
            while i < n and depth:
                if src.startswith("/*", i): depth += 1; i += 2
                elif src.startswith("*/", i): depth -= 1; i += 2
                else: i += 1
        elif c == '"' or (c == "r" and re.match(r'r#*"', src[i:])) or (c == "b" and src.startswith('b"', i)):
            m = re.match(r'b?r(#*)"', src[i:])
            if m:   # raw string: ends at "#*
                end = src.find('"' + m.group(1), i + m.end())
                i = (end + 1 + len(m.group(1))) if end >= 0 else n
            else:
                i += 2 if c == "b" else 1
                while i < n and src[i] != '"':
                    i += 2 if src[i] == "\\" else 1
                i += 1
        elif c == "'":
            m = re.match(r"'(\\.[^']*|[^'\\])'", src[i:])
            if m: i += m.end()           # a char literal
            else: out.append(c); i += 1  # a lifetime
        else:
            out.append(c); i += 1
    return "".join(out)


def test_every_rust_source_has_balanced_delimiters_and_declared_modules():
    """No cargo here: the cheapest structural check a compiler would make first - (), [], {} balance in every .rs file of the shim
    crate and of tools/upstream_kat, and every `mod x;` of lib.rs has its file."""
    roots = [os.path.join(ROOT, "rust", "border-amd-agent", "src"), os.path.join(ROOT, "tools", "upstream_kat", "src")]
    files = [os.path.join(r, f) for r in roots for f in sorted(os.listdir(r)) if f.endswith(".rs")]
    assert len(files) >= 13
    pairs = {")": "(", "]": "[", "}": "{"}
    for path in files:
        code = _rust_code_only(open(path).read())
        stack = []
        for k, ch in enumerate(code):
            if ch in "([{": stack.append(ch)
            elif ch in ")]}":
                assert stack and stack[-1] == pairs[ch], (path, code[max(0, k - 80):k + 1])
                stack.pop()
        assert not stack, (path, stack[-3:])
    lib = open(os.path.join(roots[0], "lib.rs")).read()
    for m in re.finditer(r"^\s*(?:pub\s+)?mod\s+(\w+)\s*;", lib, re.M):
        assert os.path.exists(os.path.join(roots[0], m.group(1) + ".rs")), m.group(1)
