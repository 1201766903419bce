// Reader / writer for the named-tensor archives libtorch produces for tch's `VarStore::save` / `VarStore::load` when the
// path does not end in ".safetensors" - the reference's default checkpoint names end in ".pt.tch"
// (border-tch-agent/src/dqn/base.rs:348-362, dqn/model/base.rs:134-148, sac/base.rs:313-345).
//
// tch 0.16 (third-party, not under /root/reference) routes those paths through torch-sys `at_save_multi`:
//     torch::serialize::OutputArchive ar;  ar.write(name_i, tensor_i);  ar.save_to(path);
// and reads them back with torch::jit::load + named_parameters().  What that writes is a TorchScript module archive:
//   an uncompressed ZIP (entries "<stem>/..." where <stem> is the file name up to its last '.') holding
//     data/<k>             raw little-endian storage of tensor k, 64-byte aligned in the file
//     data.pkl             pickle protocol 2: `__torch__.Module` object whose state dict maps each variable name to
//                          torch._utils._rebuild_tensor_v2(pers_id('storage', torch.FloatStorage, '<k>', 'cpu', numel),
//                                                          offset, sizes, strides, requires_grad, OrderedDict())
//     code/__torch__.py    the class stub listing the names under __parameters__
//     constants.pkl, version ("3"), byteorder ("little")
// Pinned against libtorch itself (oracle/libtorch_archive.cpp builds fixtures with exactly that call sequence and loads
// what this file writes; tests/test_checkpoint_archive.py), not against a tch-written file (none exists offline).
//
// Host-only code; no torch types, no device work.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <exception>

#include <map>
#include <memory>
#include <string>
#include <vector>

namespace bdr {
namespace tcha {

struct Tensor {
    std::string name;
    std::vector<uint64_t> dims;
    std::vector<float> data;   // contiguous, row-major
};

inline uint32_t crc32(const uint8_t* p, size_t n)
{
    static uint32_t table[256];
    static bool init = false;
    if (!init) {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
        init = true;
    }
    uint32_t c = 0xFFFFFFFFu;
    for (size_t i = 0; i < n; ++i) c = table[(c ^ p[i]) & 0xFF] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}

// ------------------------------------------------------------------------------------------------ writer
struct Buf {
    std::string s;
    void u8(uint8_t v) { s.push_back((char)v); }
    void u16(uint16_t v) { u8(v & 0xFF); u8(v >> 8); }
    void u32(uint32_t v) { u16(v & 0xFFFF); u16(v >> 16); }
    void u64(uint64_t v) { u32((uint32_t)v); u32((uint32_t)(v >> 32)); }
    void raw(const void* p, size_t n) { s.append((const char*)p, n); }
    void str(const std::string& t) { s += t; }
};

struct Pickler {
    Buf b;
    uint32_t memo = 0;
    uint32_t put()
    {
        if (memo < 256) { b.u8('q'); b.u8((uint8_t)memo); } else { b.u8('r'); b.u32(memo); }
        return memo++;
    }
    void get(uint32_t id)
    {
        if (id < 256) { b.u8('h'); b.u8((uint8_t)id); } else { b.u8('j'); b.u32(id); }
    }
    void global(const char* mod, const char* name) { b.u8('c'); b.str(mod); b.u8('\n'); b.str(name); b.u8('\n'); }
    void unicode(const std::string& t) { b.u8('X'); b.u32((uint32_t)t.size()); b.str(t); }
    void integer(uint64_t v)
    {
        if (v < 256) { b.u8('K'); b.u8((uint8_t)v); }
        else if (v < 65536) { b.u8('M'); b.u16((uint16_t)v); }
        else if (v < 0x80000000ull) { b.u8('J'); b.u32((uint32_t)v); }
        else { b.u8(0x8a); b.u8(8); b.u64(v); }   // LONG1, 8 little-endian bytes (v < 2^63)
    }
    void int_tuple(const std::vector<uint64_t>& v)
    {
        b.u8('(');
        for (auto x : v) integer(x);
        b.u8('t');
    }
};

inline std::string make_data_pkl(const std::vector<Tensor>& ts)
{
    Pickler p;
    p.b.u8(0x80); p.b.u8(2);
    p.global("__torch__", "Module"); p.put();
    p.b.u8(')'); p.b.u8(0x81);   // NEWOBJ
    p.b.u8('}'); p.b.u8('(');
    uint32_t m_rebuild = 0, m_storage = 0, m_float = 0, m_cpu = 0, m_odict = 0;
    for (size_t k = 0; k < ts.size(); ++k) {
        const Tensor& t = ts[k];
        p.unicode(t.name); p.put();
        if (k == 0) { p.global("torch._utils", "_rebuild_tensor_v2"); m_rebuild = p.put(); } else p.get(m_rebuild);
        p.b.u8('('); p.b.u8('(');
        if (k == 0) { p.unicode("storage"); m_storage = p.put(); p.global("torch", "FloatStorage"); m_float = p.put(); }
        else { p.get(m_storage); p.get(m_float); }
        p.unicode(std::to_string(k)); p.put();
        if (k == 0) { p.unicode("cpu"); m_cpu = p.put(); } else p.get(m_cpu);
        p.integer(t.data.size());
        p.b.u8('t'); p.b.u8('Q'); p.put();   // BINPERSID
        p.integer(0);                          // storage offset
        p.int_tuple(t.dims);
        std::vector<uint64_t> strides(t.dims.size(), 1);
        for (size_t d = t.dims.size(); d-- > 1;) strides[d - 1] = strides[d] * t.dims[d];
        p.int_tuple(strides);
        p.b.u8(0x89);                          // requires_grad = False
        if (k == 0) { p.global("collections", "OrderedDict"); m_odict = p.put(); } else p.get(m_odict);
        p.b.u8(')'); p.b.u8('R');
        p.b.u8('t'); p.b.u8('R'); p.put();
    }
    p.b.u8('u'); p.b.u8('b'); p.put(); p.b.u8('.');
    return p.b.s;
}

inline std::string make_code(const std::vector<Tensor>& ts)
{
    std::string c = "class Module(Module):\n  __parameters__ = [";
    for (const auto& t : ts) c += "\"" + t.name + "\", ";
    c += "]\n  __buffers__ = []\n  __annotations__ = []\n";
    for (const auto& t : ts) c += "  __annotations__[\"" + t.name + "\"] = Tensor\n";
    return c;
}

struct ZipWriter {
    Buf out, central;
    uint16_t n = 0;
    // align = 64 for tensor data (what PyTorchStreamWriter does, so the file can be mmapped), 0 otherwise
    void add(const std::string& name, const void* data, size_t size, bool align)
    {
        const uint32_t crc = crc32((const uint8_t*)data, size);
        const uint32_t off = (uint32_t)out.s.size();
        std::string extra;
        if (align) {
            // "FB" extra field padded with 'Z' so that the payload starts on a 64-byte boundary
            const size_t start = out.s.size() + 30 + name.size() + 4;
            const size_t pad = (64 - start % 64) % 64;
            extra = "FB";
            extra.push_back((char)(pad & 0xFF)); extra.push_back((char)(pad >> 8));
            extra.append(pad, 'Z');
        }
        auto header = [&](Buf& b, bool is_central) {
            b.u32(is_central ? 0x02014b50u : 0x04034b50u);
            if (is_central) b.u16(0);   // version made by
            b.u16(0);                    // version needed
            b.u16(0x0800);               // UTF-8 names
            b.u16(0);                    // stored
            b.u16(0); b.u16(0);          // time, date
            b.u32(crc); b.u32((uint32_t)size); b.u32((uint32_t)size);
            b.u16((uint16_t)name.size());
            b.u16(is_central ? 0 : (uint16_t)extra.size());
            if (is_central) { b.u16(0); b.u16(0); b.u16(0); b.u32(0); b.u32(off); }
            b.str(name);
            if (!is_central) b.str(extra);
        };
        header(out, false);
        out.raw(data, size);
        header(central, true);
        ++n;
    }
    std::string finish()
    {
        const uint32_t cd_off = (uint32_t)out.s.size(), cd_size = (uint32_t)central.s.size();
        out.str(central.s);
        out.u32(0x06054b50u); out.u16(0); out.u16(0); out.u16(n); out.u16(n); out.u32(cd_size); out.u32(cd_off); out.u16(0);
        return std::move(out.s);
    }
};

inline std::string archive_stem(const std::string& path)
{
    const size_t slash = path.find_last_of('/');
    std::string base = slash == std::string::npos ? path : path.substr(slash + 1);
    const size_t dot = base.find_last_of('.');
    if (dot != std::string::npos && dot > 0) base = base.substr(0, dot);
    return base.empty() ? std::string("archive") : base;
}

// returns "" on success, else the error text
inline std::string write_archive(const std::string& path, const std::vector<Tensor>& ts)
{
    uint64_t total = 0;
    for (const auto& t : ts) total += t.data.size() * 4 + 256;
    if (total >= 0xF0000000ull) return "checkpoint larger than 4 GiB (zip64 is not written)";
    const std::string stem = archive_stem(path);
    ZipWriter z;
    for (size_t k = 0; k < ts.size(); ++k) z.add(stem + "/data/" + std::to_string(k), ts[k].data.data(), ts[k].data.size() * 4, true);
    const std::string pkl = make_data_pkl(ts), code = make_code(ts);
    z.add(stem + "/data.pkl", pkl.data(), pkl.size(), false);
    z.add(stem + "/code/__torch__.py", code.data(), code.size(), false);
    z.add(stem + "/constants.pkl", "\x80\x02).", 4, false);
    z.add(stem + "/version", "3\n", 2, false);
    z.add(stem + "/byteorder", "little", 6, false);
    const std::string bytes = z.finish();
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) return "cannot open " + path + " for writing";
    bool ok = fwrite(bytes.data(), 1, bytes.size(), f) == bytes.size();
    ok = fflush(f) == 0 && ok;
    fclose(f);
    return ok ? "" : "write to " + path + " failed";
}

// ------------------------------------------------------------------------------------------------ reader
struct ZipEntry { uint64_t data_off = 0, size = 0; bool stored = true; };

struct ZipReader {
    std::string bytes;
    std::map<std::string, ZipEntry> entries;
    std::string err;
    uint16_t r16(size_t o) const { return (uint16_t)((uint8_t)bytes[o] | ((uint8_t)bytes[o + 1] << 8)); }
    uint32_t r32(size_t o) const { return (uint32_t)r16(o) | ((uint32_t)r16(o + 2) << 16); }
    uint64_t r64(size_t o) const { return (uint64_t)r32(o) | ((uint64_t)r32(o + 4) << 32); }
    bool open(const std::string& path)
    {
        FILE* f = fopen(path.c_str(), "rb");
        if (!f) { err = "cannot open " + path; return false; }
        fseek(f, 0, SEEK_END);
        const long len = ftell(f);
        fseek(f, 0, SEEK_SET);
        if (len < 22) { fclose(f); err = path + " is not a zip archive"; return false; }
        bytes.resize((size_t)len);
        const bool ok = fread(&bytes[0], 1, (size_t)len, f) == (size_t)len;
        fclose(f);
        if (!ok) { err = "short read of " + path; return false; }
        // end-of-central-directory record: scan back over a possible comment
        size_t eocd = std::string::npos;
        for (size_t o = bytes.size() - 22;; --o) {
            if (r32(o) == 0x06054b50u) { eocd = o; break; }
            if (o == 0 || bytes.size() - o > 22 + 65535) break;
        }
        if (eocd == std::string::npos) { err = path + " is not a zip archive (libtorch .pt / .pt.tch files are)"; return false; }
        uint64_t n = r16(eocd + 10), cd_off = r32(eocd + 16);
        if (eocd >= 20 && r32(eocd - 20) == 0x07064b50u) {   // zip64 locator -> zip64 end record
            const uint64_t z = r64(eocd - 20 + 8);
            if (z <= bytes.size() && bytes.size() - z >= 56 && r32(z) == 0x06064b50u) { n = r64(z + 32); cd_off = r64(z + 48); }
        }
        if (n > bytes.size() / 46) { err = "corrupt zip end record (entry count)"; return false; }
        size_t o = cd_off;
        for (uint64_t i = 0; i < n; ++i) {
            if (o > bytes.size() || bytes.size() - o < 46 || r32(o) != 0x02014b50u) { err = "corrupt zip central directory"; return false; }
            const uint16_t method = r16(o + 10), nl = r16(o + 28), xl = r16(o + 30), cl = r16(o + 32);
            uint64_t csize = r32(o + 20), usize = r32(o + 24), lho = r32(o + 42);
            // name, extra field and comment of this record lie inside the file (all lengths come from the file)
            if (bytes.size() - o - 46 < (size_t)nl + xl + cl) { err = "corrupt zip central directory (record runs past the end of the file)"; return false; }
            const std::string name = bytes.substr(o + 46, nl);
            // zip64 extended information
            size_t x = o + 46 + nl;
            const size_t xend = x + xl;
            while (x + 4 <= xend) {
                const uint16_t id = r16(x), sz = r16(x + 2);
                if (x + 4 + sz > xend) break;   // truncated extra field
                if (id == 1) {
                    size_t q = x + 4;
                    if (usize == 0xFFFFFFFFu && q + 8 <= x + 4 + sz) { usize = r64(q); q += 8; }
                    if (csize == 0xFFFFFFFFu && q + 8 <= x + 4 + sz) { csize = r64(q); q += 8; }
                    if (lho == 0xFFFFFFFFu && q + 8 <= x + 4 + sz) { lho = r64(q); q += 8; }
                }
                x += 4 + sz;
            }
            if (lho > bytes.size() || bytes.size() - lho < 30 || r32(lho) != 0x04034b50u) { err = "corrupt zip local header"; return false; }
            ZipEntry e;
            e.data_off = lho + 30 + r16(lho + 26) + r16(lho + 28);
            e.size = usize;
            e.stored = method == 0 && csize == usize;   // libtorch deflates only the code/ entries, which are not needed here
            if (e.data_off > bytes.size() || csize > bytes.size() - e.data_off) { err = "zip entry " + name + " runs past the end of the file"; return false; }
            entries[name] = e;
            o += 46 + nl + xl + cl;
        }
        return true;
    }
};

// The pickle subset libtorch's and Python's picklers emit for tensor containers.
struct PVal;
using PRef = std::shared_ptr<PVal>;
struct PVal {
    enum Kind { None, Bool, Int, Float, Str, Global, Tuple, List, Dict, Storage, TensorV, Obj } kind = None;
    int64_t i = 0;
    double f = 0;
    std::string s;                               // Str / Global ("module name") / Storage key
    std::vector<PRef> items;                      // Tuple / List
    std::vector<std::pair<PRef, PRef>> dict;      // Dict / Obj state
    // TensorV
    std::string storage_key, storage_type;
    uint64_t offset = 0, storage_numel = 0;
    std::vector<uint64_t> sizes, strides;
};
inline PRef mk(PVal::Kind k) { auto p = std::make_shared<PVal>(); p->kind = k; return p; }

struct Unpickler {
    const uint8_t* p;
    size_t n, o = 0;
    std::string err;
    std::vector<PRef> stack;
    std::vector<size_t> marks;
    std::map<uint32_t, PRef> memo;
    Unpickler(const uint8_t* data, size_t size) : p(data), n(size) {}
    bool need(size_t k) { if (o + k > n) { err = "truncated pickle"; return false; } return true; }
    uint32_t u32() { uint32_t v = p[o] | (p[o + 1] << 8) | (p[o + 2] << 16) | ((uint32_t)p[o + 3] << 24); o += 4; return v; }
    std::string line() { std::string s; while (o < n && p[o] != '\n') s.push_back((char)p[o++]); ++o; return s; }
    PRef pop() { if (stack.empty()) { err = "pickle stack underflow"; return mk(PVal::None); } PRef v = stack.back(); stack.pop_back(); return v; }
    std::vector<PRef> pop_mark()
    {
        if (marks.empty()) { err = "pickle mark underflow"; return {}; }
        const size_t m = marks.back();
        marks.pop_back();
        std::vector<PRef> v(stack.begin() + m, stack.end());
        stack.resize(m);
        return v;
    }
    static std::vector<uint64_t> ints(const PRef& t)
    {
        std::vector<uint64_t> v;
        for (const auto& x : t->items) v.push_back((uint64_t)x->i);
        return v;
    }
    PRef reduce(const PRef& fn, const PRef& args)
    {
        if (fn->kind == PVal::Global) {
            const std::string& g = fn->s;
            if ((g == "torch._utils _rebuild_tensor_v2" || g == "torch._utils _rebuild_tensor") && args->items.size() >= 4 &&
                args->items[0]->kind == PVal::Storage) {
                PRef t = mk(PVal::TensorV);
                t->storage_key = args->items[0]->s;
                t->storage_type = args->items[0]->storage_type;
                t->storage_numel = args->items[0]->storage_numel;
                t->offset = (uint64_t)args->items[1]->i;
                t->sizes = ints(args->items[2]);
                t->strides = ints(args->items[3]);
                return t;
            }
            if (g == "torch._utils _rebuild_parameter" && !args->items.empty()) return args->items[0];
            if (g == "collections OrderedDict") return mk(PVal::Dict);
        }
        return mk(PVal::Obj);
    }
    PRef run()
    {
        while (o < n && err.empty()) {
            const uint8_t op = p[o++];
            switch (op) {
                case 0x80: if (!need(1)) break; ++o; break;                                   // PROTO
                case 0x95: if (!need(8)) break; o += 8; break;                               // FRAME
                case 'c': { PRef g = mk(PVal::Global); g->s = line(); g->s += " " + line(); stack.push_back(g); break; }
                case 0x93: { PRef name = pop(), mod = pop(); PRef g = mk(PVal::Global); g->s = mod->s + " " + name->s; stack.push_back(g); break; }   // STACK_GLOBAL
                case 'q': if (!need(1) || stack.empty()) { err = "bad BINPUT"; break; } memo[p[o++]] = stack.back(); break;
                case 'r': if (!need(4) || stack.empty()) { err = "bad LONG_BINPUT"; break; } memo[u32()] = stack.back(); break;
                case 0x94: if (stack.empty()) { err = "bad MEMOIZE"; break; } { const uint32_t id = (uint32_t)memo.size(); memo[id] = stack.back(); } break;
                case 'h': { if (!need(1)) break; auto it = memo.find(p[o++]); if (it == memo.end()) { err = "bad BINGET"; break; } stack.push_back(it->second); break; }
                case 'j': { if (!need(4)) break; auto it = memo.find(u32()); if (it == memo.end()) { err = "bad LONG_BINGET"; break; } stack.push_back(it->second); break; }
                case ')': stack.push_back(mk(PVal::Tuple)); break;
                case '}': stack.push_back(mk(PVal::Dict)); break;
                case ']': stack.push_back(mk(PVal::List)); break;
                case '(': marks.push_back(stack.size()); break;
                case 'N': stack.push_back(mk(PVal::None)); break;
                case 0x88: case 0x89: { PRef b = mk(PVal::Bool); b->i = op == 0x88; stack.push_back(b); break; }
                case 'K': { if (!need(1)) break; PRef v = mk(PVal::Int); v->i = p[o++]; stack.push_back(v); break; }
                case 'M': { if (!need(2)) break; PRef v = mk(PVal::Int); v->i = p[o] | (p[o + 1] << 8); o += 2; stack.push_back(v); break; }
                case 'J': { if (!need(4)) break; PRef v = mk(PVal::Int); v->i = (int32_t)u32(); stack.push_back(v); break; }
                case 0x8a: {                                                                   // LONG1
                    if (!need(1)) break;
                    const uint8_t k = p[o++];
                    if (!need(k) || k > 8) { err = "unsupported LONG1"; break; }
                    uint64_t v = 0;
                    for (uint8_t b = 0; b < k; ++b) v |= (uint64_t)p[o + b] << (8 * b);
                    if (k && k < 8 && (p[o + k - 1] & 0x80)) v |= ~0ull << (8 * k);
                    o += k;
                    PRef r = mk(PVal::Int); r->i = (int64_t)v; stack.push_back(r);
                    break;
                }
                case 'G': { if (!need(8)) break; uint64_t b = 0; for (int k = 0; k < 8; ++k) b = (b << 8) | p[o + k]; o += 8; PRef v = mk(PVal::Float); memcpy(&v->f, &b, 8); stack.push_back(v); break; }
                case 'X': { if (!need(4)) break; const uint32_t k = u32(); if (!need(k)) break; PRef v = mk(PVal::Str); v->s.assign((const char*)p + o, k); o += k; stack.push_back(v); break; }
                case 0x8c: { if (!need(1)) break; const uint8_t k = p[o++]; if (!need(k)) break; PRef v = mk(PVal::Str); v->s.assign((const char*)p + o, k); o += k; stack.push_back(v); break; }   // SHORT_BINUNICODE
                case 't': { PRef t = mk(PVal::Tuple); t->items = pop_mark(); stack.push_back(t); break; }
                case 0x85: { PRef t = mk(PVal::Tuple); t->items = {pop()}; stack.push_back(t); break; }
                case 0x86: { PRef b = pop(), a = pop(); PRef t = mk(PVal::Tuple); t->items = {a, b}; stack.push_back(t); break; }
                case 0x87: { PRef c = pop(), b = pop(), a = pop(); PRef t = mk(PVal::Tuple); t->items = {a, b, c}; stack.push_back(t); break; }
                case 'l': { PRef t = mk(PVal::List); t->items = pop_mark(); stack.push_back(t); break; }
                case 'a': { PRef v = pop(); if (stack.empty()) { err = "bad APPEND"; break; } stack.back()->items.push_back(v); break; }
                case 'e': { auto v = pop_mark(); if (stack.empty()) { err = "bad APPENDS"; break; } for (auto& x : v) stack.back()->items.push_back(x); break; }
                case 's': { PRef v = pop(), k = pop(); if (stack.empty()) { err = "bad SETITEM"; break; } stack.back()->dict.emplace_back(k, v); break; }
                case 'u': {
                    auto v = pop_mark();
                    if (stack.empty() || v.size() % 2) { err = "bad SETITEMS"; break; }
                    for (size_t k = 0; k + 1 < v.size(); k += 2) stack.back()->dict.emplace_back(v[k], v[k + 1]);
                    break;
                }
                case 'Q': {                                                                     // BINPERSID: ('storage', type, key, location, numel)
                    PRef id = pop();
                    PRef s = mk(PVal::Storage);
                    if (id->kind == PVal::Tuple && id->items.size() >= 5 && id->items[0]->s == "storage") {
                        s->storage_type = id->items[1]->s;
                        s->s = id->items[2]->s;
                        s->storage_numel = (uint64_t)id->items[4]->i;
                    } else err = "unknown persistent id in data.pkl";
                    stack.push_back(s);
                    break;
                }
                case 0x81: { pop(); pop(); stack.push_back(mk(PVal::Obj)); break; }            // NEWOBJ
                case 'R': { PRef args = pop(), fn = pop(); stack.push_back(reduce(fn, args)); break; }
                case 'b': {                                                                     // BUILD: the state becomes the object's dict
                    PRef state = pop();
                    if (stack.empty()) { err = "bad BUILD"; break; }
                    if (state->kind == PVal::Dict) for (auto& kv : state->dict) stack.back()->dict.push_back(kv);
                    else if (state->kind == PVal::Tuple)   // (dict_state, slots_state)
                        for (auto& part : state->items) if (part->kind == PVal::Dict) for (auto& kv : part->dict) stack.back()->dict.push_back(kv);
                    break;
                }
                case '.': return pop();
                default: err = "unsupported pickle opcode 0x" + std::string(1, "0123456789abcdef"[op >> 4]) + std::string(1, "0123456789abcdef"[op & 15]);
            }
        }
        if (err.empty()) err = "pickle ended without STOP";
        return mk(PVal::None);
    }
};

// walks nested module objects / dicts, joining names with '.', the way named_parameters() reports them
inline void collect_tensors(const PRef& v, const std::string& prefix, std::vector<std::pair<std::string, PRef>>& out, int depth = 0)
{
    if (depth > 16) return;
    for (const auto& kv : v->dict) {
        if (kv.first->kind != PVal::Str) continue;
        const std::string name = prefix.empty() ? kv.first->s : prefix + "." + kv.first->s;
        if (kv.second->kind == PVal::TensorV) out.emplace_back(name, kv.second);
        else if (kv.second->kind == PVal::Obj || kv.second->kind == PVal::Dict) collect_tensors(kv.second, name, out, depth + 1);
    }
}

// returns "" on success
inline std::string read_archive_impl(const std::string& path, std::vector<Tensor>& out)
{
    ZipReader z;
    if (!z.open(path)) return z.err;
    std::string prefix;
    const ZipEntry* pkl = nullptr;
    for (const auto& kv : z.entries) {
        const std::string& nm = kv.first;
        if (nm == "data.pkl" || (nm.size() > 9 && nm.compare(nm.size() - 9, 9, "/data.pkl") == 0)) {
            // the module's own pickle sits at the top of the archive directory (not under code/ or .data/)
            const std::string pre = nm.substr(0, nm.size() - 8);
            if (!pkl || pre.size() < prefix.size()) { pkl = &kv.second; prefix = pre; }
        }
    }
    if (!pkl) return path + " holds no data.pkl (not a libtorch archive)";
    if (!pkl->stored) return path + ": data.pkl is compressed";
    Unpickler u((const uint8_t*)z.bytes.data() + pkl->data_off, pkl->size);
    PRef root = u.run();
    if (!u.err.empty() && root->kind == PVal::None) return path + ": " + u.err;
    std::vector<std::pair<std::string, PRef>> found;
    collect_tensors(root, "", found);
    for (const auto& nt : found) {
        const PVal& t = *nt.second;
        if (t.storage_type != "torch FloatStorage") return path + ": variable '" + nt.first + "' is not float32";
        auto it = z.entries.find(prefix + "data/" + t.storage_key);
        if (it == z.entries.end() || !it->second.stored) return path + ": storage " + t.storage_key + " of '" + nt.first + "' is missing";
        const char* src = z.bytes.data() + it->second.data_off;
        const uint64_t avail = it->second.size / 4;
        Tensor r;
        r.name = nt.first;
        r.dims = t.sizes;
        uint64_t numel = 1;
        bool overflow = t.sizes.size() > 16;
        for (auto d : t.sizes) {   // sizes come from the file: no product overflow, nothing larger than a model could be
            if (d != 0 && numel > (1ull << 33) / d) overflow = true;
            if (!overflow) numel *= d;
        }
        if (overflow || numel > (1ull << 33)) return path + ": variable '" + nt.first + "' has an implausible shape";
        if (t.sizes.size() != t.strides.size()) return path + ": variable '" + nt.first + "' has inconsistent strides";
        if (numel > 0 && avail == 0) return path + ": variable '" + nt.first + "' reads past its storage";
        r.data.resize(numel);
        std::vector<uint64_t> idx(t.sizes.size(), 0);
        for (uint64_t e = 0; e < numel; ++e) {   // general strided read (libtorch writes contiguous tensors; Python exports may not be)
            uint64_t off = t.offset;
            bool past = off >= avail;
            for (size_t d = 0; d < idx.size() && !past; ++d) {
                if (idx[d] != 0 && t.strides[d] > avail / idx[d]) { past = true; break; }
                off += idx[d] * t.strides[d];
                past = off >= avail;
            }
            if (past) return path + ": variable '" + nt.first + "' reads past its storage";
            float v;
            memcpy(&v, src + off * 4, 4);
            r.data[e] = v;
            for (size_t d = idx.size(); d-- > 0;) { if (++idx[d] < t.sizes[d]) break; idx[d] = 0; }
        }
        out.push_back(std::move(r));
    }
    return "";
}

// Everything above is driven by lengths and counts read from the file; whatever still throws (bad_alloc, length_error,
// out_of_range) on a malformed archive must not cross the extern "C" boundary.
inline std::string read_archive(const std::string& path, std::vector<Tensor>& out)
{
    try {
        return read_archive_impl(path, out);
    } catch (const std::exception& e) {
        return path + ": malformed archive (" + e.what() + ")";
    } catch (...) {
        return path + ": malformed archive";
    }
}

}  // namespace tcha
}  // namespace bdr
