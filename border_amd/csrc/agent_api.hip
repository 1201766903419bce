// C-ABI entry points shared by every agent kind (include/border_amd.h): dispatch through the
// polymorphic handle of agent_base.hpp.  Agent-specific constructors live next to their kernels.
#include <cerrno>
#include <cstdlib>
#include <sys/stat.h>
#include <sys/types.h>

#include <algorithm>
#include <cmath>

#include "agent_base.hpp"
#include "tch_archive.hpp"

using namespace bdr;

namespace bdr {
int32_t dqn_cnn_create(const bdr_dqn_config* cfg, bdr_agent** out);
int32_t dqn_mlp_create(const bdr_dqn_config* cfg, bdr_agent** out);
int32_t dqn_cnn_update_on_batch(bdr_agent* a, uint64_t n, const void* obs, const int64_t* act, const void* next_obs,
                                const float* reward, const int8_t* term, const float* weight);
int32_t dqn_mlp_update_on_batch(bdr_agent* a, uint64_t n, const void* obs, const int64_t* act, const void* next_obs,
                                const float* reward, const int8_t* term, const float* weight);
int32_t dqn_cnn_qvalues(bdr_agent* a, uint64_t n, const void* obs, float* q_out);
int32_t dqn_mlp_qvalues(bdr_agent* a, uint64_t n, const void* obs, float* q_out);
int32_t dqn_cnn_probe(bdr_agent* a, int32_t what, float* out, uint64_t n);
int32_t dqn_mlp_probe(bdr_agent* a, int32_t what, float* out, uint64_t n);

// Checkpoint containers, chosen by the file name exactly as tch's VarStore::save / load choose:
//  * "*.safetensors": 8-byte little-endian header length, JSON header
//    {"name": {"dtype": "F32", "shape": [...], "data_offsets": [begin, end]}, ...}, raw little-endian data;
//  * anything else (the reference's default "*.pt.tch"): the libtorch named-tensor archive of tch_archive.hpp.
// Both hold the reference's variable names (c1.weight ... l2.bias, mlp.ln{i}.*, ...) in the reference's layouts, so files
// written here load into border-tch-agent's VarStore and vice versa.
static bool is_safetensors_path(const std::string& path)
{
    const std::string ext = ".safetensors";
    return path.size() >= ext.size() && path.compare(path.size() - ext.size(), ext.size(), ext) == 0;
}

static int32_t save_archive(const std::string& path, const std::vector<NamedTensor>& meta, const float* data, size_t n)
{
    std::vector<tcha::Tensor> ts;
    size_t o = 0;
    for (const auto& m : meta) {
        size_t k = 1;
        for (auto d : m.dims) k *= d;
        if (o + k > n) return fail(BDR_ERR_IO, "tensor metadata exceeds the parameter vector");
        tcha::Tensor t;
        t.name = m.name; t.dims = m.dims; t.data.assign(data + o, data + o + k);
        ts.push_back(std::move(t));
        o += k;
    }
    if (o != n) return fail(BDR_ERR_IO, "tensor metadata does not cover the parameter vector");
    const std::string err = tcha::write_archive(path, ts);
    return err.empty() ? BDR_OK : fail(BDR_ERR_IO, "%s", err.c_str());
}

static int32_t load_archive(const std::string& path, const std::vector<NamedTensor>& meta, float* data, size_t n)
{
    std::vector<tcha::Tensor> ts;
    const std::string err = tcha::read_archive(path, ts);
    if (!err.empty()) return fail(BDR_ERR_IO, "%s", err.c_str());
    size_t o = 0;
    for (const auto& m : meta) {   // as for safetensors: every variable of the model, with its reference shape; extras are ignored
        const tcha::Tensor* t = nullptr;
        for (const auto& c : ts) if (c.name == m.name) { t = &c; break; }
        if (!t) return fail(BDR_ERR_IO, "%s: variable '%s' is missing", path.c_str(), m.name.c_str());
        if (t->dims != m.dims || o + t->data.size() > n) return fail(BDR_ERR_IO, "%s: variable '%s' has a different shape", path.c_str(), m.name.c_str());
        std::copy(t->data.begin(), t->data.end(), data + o);
        o += t->data.size();
    }
    if (o != n) return fail(BDR_ERR_IO, "%s does not cover the model's parameters", path.c_str());
    return BDR_OK;
}

static int32_t save_safetensors(const std::string& path, const std::vector<NamedTensor>& meta, const float* data, size_t n)
{
    std::string hdr = "{";
    size_t o = 0;
    for (size_t t = 0; t < meta.size(); ++t) {
        size_t k = 1;
        std::string shape;
        for (size_t d = 0; d < meta[t].dims.size(); ++d) { k *= meta[t].dims[d]; shape += (d ? "," : "") + std::to_string(meta[t].dims[d]); }
        if (o + k > n) return fail(BDR_ERR_IO, "tensor metadata exceeds the parameter vector");
        hdr += (t ? "," : "") + std::string("\"") + meta[t].name + "\":{\"dtype\":\"F32\",\"shape\":[" + shape + "],\"data_offsets\":[" +
               std::to_string(o * 4) + "," + std::to_string((o + k) * 4) + "]}";
        o += k;
    }
    hdr += "}";
    if (o != n) return fail(BDR_ERR_IO, "tensor metadata does not cover the parameter vector");
    while (hdr.size() % 8) hdr += ' ';   // the data section stays 8-byte aligned
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) return fail(BDR_ERR_IO, "cannot open %s for writing", path.c_str());
    const uint64_t hl = hdr.size();
    bool ok = fwrite(&hl, 8, 1, f) == 1 && fwrite(hdr.data(), 1, hdr.size(), f) == hdr.size() && fwrite(data, 4, n, f) == n;
    ok = fflush(f) == 0 && ok;
    fclose(f);
    return ok ? BDR_OK : fail(BDR_ERR_IO, "write to %s failed", path.c_str());
}

namespace {
// the subset of JSON a safetensors header uses
struct StEntry { std::string dtype; std::vector<uint64_t> shape; uint64_t begin = 0, end = 0; };
struct JsonCur {
    const std::string& s; size_t i = 0; bool ok = true;
    explicit JsonCur(const std::string& str) : s(str) {}
    void ws() { while (i < s.size() && (s[i] == ' ' || s[i] == '\n' || s[i] == '\t' || s[i] == '\r')) ++i; }
    bool eat(char c) { ws(); if (i < s.size() && s[i] == c) { ++i; return true; } return false; }
    std::string str()
    {
        std::string o;
        if (!eat('"')) { ok = false; return o; }
        while (i < s.size() && s[i] != '"') { if (s[i] == '\\' && i + 1 < s.size()) ++i; o += s[i++]; }
        if (i >= s.size()) ok = false; else ++i;
        return o;
    }
    uint64_t num() { ws(); uint64_t v = 0; bool any = false; while (i < s.size() && s[i] >= '0' && s[i] <= '9') { v = v * 10 + (uint64_t)(s[i++] - '0'); any = true; } ok = ok && any; return v; }
    void skip()   // any value
    {
        ws();
        if (i >= s.size()) { ok = false; return; }
        if (s[i] == '"') { str(); return; }
        if (s[i] == '{' || s[i] == '[') {
            const char close = s[i] == '{' ? '}' : ']';
            ++i;
            if (eat(close)) return;
            do { if (close == '}') { str(); if (!eat(':')) ok = false; } skip(); } while (ok && eat(','));
            if (!eat(close)) ok = false;
            return;
        }
        while (i < s.size() && s[i] != ',' && s[i] != '}' && s[i] != ']') ++i;
    }
};
bool parse_safetensors_header(const std::string& h, std::vector<std::pair<std::string, StEntry>>& out)
{
    JsonCur c(h);
    if (!c.eat('{')) return false;
    if (c.eat('}')) return true;
    do {
        const std::string name = c.str();
        if (!c.ok || !c.eat(':')) return false;
        if (name == "__metadata__") { c.skip(); continue; }
        StEntry e;
        if (!c.eat('{')) return false;
        do {
            const std::string key = c.str();
            if (!c.ok || !c.eat(':')) return false;
            if (key == "dtype") e.dtype = c.str();
            else if (key == "shape") { if (!c.eat('[')) return false; if (!c.eat(']')) { do e.shape.push_back(c.num()); while (c.eat(',')); if (!c.eat(']')) return false; } }
            else if (key == "data_offsets") { if (!c.eat('[')) return false; e.begin = c.num(); if (!c.eat(',')) return false; e.end = c.num(); if (!c.eat(']')) return false; }
            else c.skip();
        } while (c.ok && c.eat(','));
        if (!c.ok || !c.eat('}')) return false;
        out.emplace_back(name, e);
    } while (c.ok && c.eat(','));
    return c.ok && c.eat('}');
}
}  // namespace

static int32_t load_safetensors(const std::string& path, const std::vector<NamedTensor>& meta, float* data, size_t n)
{
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return fail(BDR_ERR_IO, "cannot open %s", path.c_str());
    uint64_t hl = 0;
    std::string hdr;
    bool ok = fread(&hl, 8, 1, f) == 1 && hl >= 2 && hl < (1ull << 26);
    if (ok) { hdr.resize(hl); ok = fread(&hdr[0], 1, hl, f) == hl; }
    std::vector<std::pair<std::string, StEntry>> ents;
    ok = ok && parse_safetensors_header(hdr, ents);
    if (!ok) { fclose(f); return fail(BDR_ERR_IO, "%s is not a safetensors file", path.c_str()); }
    size_t o = 0;
    for (const auto& t : meta) {   // every variable of the model must be present with its reference shape (extra entries are ignored)
        const StEntry* e = nullptr;
        for (const auto& kv : ents) if (kv.first == t.name) { e = &kv.second; break; }
        size_t k = 1;
        for (auto d : t.dims) k *= d;
        if (!e) { fclose(f); return fail(BDR_ERR_IO, "%s: variable '%s' is missing", path.c_str(), t.name.c_str()); }
        if (e->dtype != "F32" || e->shape != t.dims || e->end - e->begin != k * 4 || o + k > n) {
            fclose(f);
            return fail(BDR_ERR_IO, "%s: variable '%s' has a different dtype or shape", path.c_str(), t.name.c_str());
        }
        if (fseek(f, (long)(8 + hl + e->begin), SEEK_SET) != 0 || fread(data + o, 4, k, f) != k) {
            fclose(f);
            return fail(BDR_ERR_IO, "%s: short read of '%s'", path.c_str(), t.name.c_str());
        }
        o += k;
    }
    fclose(f);
    if (o != n) return fail(BDR_ERR_IO, "%s does not cover the model's parameters", path.c_str());
    return BDR_OK;
}

int32_t save_named(const std::string& path, const std::vector<NamedTensor>& meta, const float* data, size_t n)
{
    return is_safetensors_path(path) ? save_safetensors(path, meta, data, n) : save_archive(path, meta, data, n);
}

int32_t load_named(const std::string& path, const std::vector<NamedTensor>& meta, float* data, size_t n)
{
    return is_safetensors_path(path) ? load_safetensors(path, meta, data, n) : load_archive(path, meta, data, n);
}

std::string ckpt_save_path(const bdr_agent* a, const char* dir, const std::string& stem)
{
    return std::string(dir) + "/" + stem + (a->ckpt_format == BDR_CKPT_SAFETENSORS ? ".safetensors" : ".pt.tch");
}

std::string ckpt_load_path(const bdr_agent* a, const char* dir, const std::string& stem)
{
    const std::string first = ckpt_save_path(a, dir, stem);
    const std::string second = std::string(dir) + "/" + stem + (a->ckpt_format == BDR_CKPT_SAFETENSORS ? ".pt.tch" : ".safetensors");
    FILE* f = fopen(first.c_str(), "rb");
    if (f) { fclose(f); return first; }
    f = fopen(second.c_str(), "rb");
    if (f) { fclose(f); return second; }
    return first;   // the open error then names the file the reference would have looked for
}
}  // namespace bdr

// ---- device-side error words (agent_base.hpp) -------------------------------------------------------------------------
int32_t bdr_agent::err_report(const unsigned* w)
{
    if (!(w[ERR_ACTION] | w[ERR_GATE] | w[ERR_NONFINITE])) return BDR_OK;
    const unsigned act = w[ERR_ACTION], gate = w[ERR_GATE], nonf = w[ERR_NONFINITE];
    if (gate) drain_queues();   // all of the agent's queues idle before the poison word goes down
    (void)hipMemsetAsync(dev_err, 0, ERR_WORDS * sizeof(unsigned), stream);
    (void)hipStreamSynchronize(stream);
    memset(host_err, 0, ERR_WORDS * sizeof(unsigned));
    if (gate) {
        on_gate_timeout();
        const int32_t st__ = fail(BDR_ERR_HIP, "cross-queue gate %u of a %s agent timed out (the producer kernel it waits for never arrived): the agent continues on its "
                                 "fallback schedule (DqnCnn: event ordering; "
                                 "Sac: one queue; in both the parameter updates behind the failed wait were skipped on the device and the host's "
                                 "counters - n_opts, the Adam step numbers - were rolled back to the last update that was applied)", gate - 1, kind());
        g_err_deferred = 1;
        return st__;
    }
    const int32_t st__ = act ? fail(BDR_ERR_INVALID, "an action index outside [0, n_actions) reached the TD step (the reference's gather raises "
                                                      "an index error); it was clamped")
                             : fail(BDR_ERR_INVALID, "non-finite value flagged on the device (%u)", nonf);
    g_err_deferred = 1;
    return st__;
}

int32_t bdr_agent::err_check()
{
    if (!dev_err) return BDR_OK;
    unsigned w[ERR_WORDS];
    BDR_HIP(hipMemcpy(w, dev_err, sizeof w, hipMemcpyDeviceToHost));
    BDR_TRY(err_report(w));
    bool alive = false;
    const int32_t st = replay_per_check(last_replay_uid, &alive);   // (lookup + check under the registry lock)
    if (!alive) last_replay_uid = 0;
    return st;
}

int32_t bdr_agent::err_poll()
{
    if (!dev_err) return BDR_OK;
    unsigned w[ERR_WORDS];
    for (int i = 0; i < ERR_WORDS; ++i) w[i] = reinterpret_cast<volatile unsigned*>(host_err)[i];
    BDR_TRY(err_report(w));
    if (++err_poll_count % ERR_POLL_INTERVAL == 0) {
        if (!host_err_dev) BDR_HIP(hipHostGetDevicePointer((void**)&host_err_dev, host_err, 0));
        hipLaunchKernelGGL(k_err_mirror, dim3(1), dim3(64), 0, stream, (const unsigned*)dev_err, host_err_dev, (int)ERR_WORDS);
        BDR_HIP(hipGetLastError());
    }
    return BDR_OK;
}

// fs::create_dir_all (dqn/base.rs:346, iqn/base.rs:304, sac/base.rs:314)
static int32_t create_dir_all(const char* dir)
{
    std::string p(dir);
    if (p.empty()) return fail(BDR_ERR_IO, "empty checkpoint directory");
    for (size_t i = 1; i <= p.size(); ++i) {
        if (i != p.size() && p[i] != '/') continue;
        const std::string sub = p.substr(0, i);
        if (mkdir(sub.c_str(), 0777) != 0 && errno != EEXIST) return fail(BDR_ERR_IO, "cannot create directory %s: %s", sub.c_str(), strerror(errno));
    }
    struct stat st;
    if (stat(p.c_str(), &st) != 0 || !S_ISDIR(st.st_mode)) return fail(BDR_ERR_IO, "%s is not a directory", p.c_str());
    return BDR_OK;
}

// the error words after a step that THIS call enqueued and synchronised: a device-side report is then about this very step, which has
// run - bdr_last_error_is_deferred() says 2, not 1, so that a caller keeps the record and does not run the step a second time
static int32_t err_check_after_step(bdr_agent* a)
{
    const int32_t st = a->err_check();
    if (st != BDR_OK && bdr::g_err_deferred) bdr::g_err_deferred = 2;
    return st;
}

static bool is_dqn(const bdr_agent* a) { return a && (!strcmp(a->kind(), "dqn_cnn") || !strcmp(a->kind(), "dqn_mlp")); }

extern "C" {

void bdr_dqn_config_default(bdr_dqn_config* c)
{
    if (!c) return;
    memset(c, 0, sizeof *c);
    // dqn/config.rs:82-102
    c->net.kind = BDR_NET_ATARI_CNN; c->net.n_stack = 4; c->net.out_dim = 0;
    c->opt_kind = BDR_OPT_ADAM; c->lr = 0.0;
    c->beta1 = 0.9; c->beta2 = 0.999; c->weight_decay = 0.0; c->eps = 1e-8;
    c->soft_update_interval = 1; c->n_updates_per_opt = 1; c->batch_size = 1;
    c->discount_factor = 0.99; c->tau = 0.005; c->train = 0; c->double_dqn = 0;
    c->critic_loss = BDR_LOSS_MSE; c->has_clip_td_err = 0; c->record_verbose_level = 0;
    c->device = -1; c->param_seed = 0;
    c->arithmetic = BDR_ARITH_BF16X3_6;
}

int32_t bdr_dqn_create(const bdr_dqn_config* cfg, bdr_agent** out)
{
    BDR_REQUIRE(cfg && out, "null argument");
    BDR_REQUIRE(cfg->device >= 0, "No device is given for DQN agent");   // dqn/base.rs:256-259
    BDR_REQUIRE(cfg->net.kind == BDR_NET_ATARI_CNN || cfg->net.kind == BDR_NET_MLP, "unknown Q-network kind");
    BDR_REQUIRE(cfg->batch_size >= 1 && cfg->batch_size <= 65536, "batch_size out of range");
    BDR_REQUIRE(cfg->soft_update_interval >= 1 && cfg->n_updates_per_opt >= 1, "intervals must be >= 1");
    BDR_REQUIRE(cfg->critic_loss == BDR_LOSS_MSE || cfg->critic_loss == BDR_LOSS_SMOOTH_L1, "unknown critic_loss");
    BDR_REQUIRE(cfg->opt_kind == BDR_OPT_ADAM || cfg->opt_kind == BDR_OPT_ADAMW, "unknown optimizer");
    BDR_REQUIRE(cfg->arithmetic == BDR_ARITH_BF16X3_6 || cfg->arithmetic == BDR_ARITH_F32_EXACT, "unknown arithmetic %d (BDR_ARITH_*)", cfg->arithmetic);
    BDR_TRY(ensure_device(cfg->device));
    return cfg->net.kind == BDR_NET_ATARI_CNN ? dqn_cnn_create(cfg, out) : dqn_mlp_create(cfg, out);
}

int32_t bdr_agent_destroy(bdr_agent* a)
{
    if (!a) return BDR_OK;
    (void)hipSetDevice(a->device);
    for (auto& s : a->slots) { (void)hipEventDestroy(s.e0); (void)hipEventDestroy(s.e1); }
    hipStream_t st = a->stream;
    if (st) (void)hipStreamSynchronize(st);
    delete a;
    if (st) { stream_retire(st); (void)hipStreamDestroy(st); }
    return BDR_OK;
}

int32_t bdr_agent_set_train(bdr_agent* a, int32_t train) { BDR_REQUIRE(a, "null agent"); a->train = train != 0; return BDR_OK; }
int32_t bdr_agent_is_train(const bdr_agent* a, int32_t* out) { BDR_REQUIRE(a && out, "null argument"); *out = a->train; return BDR_OK; }
int32_t bdr_agent_n_opts(const bdr_agent* a, uint64_t* n) { BDR_REQUIRE(a && n, "null argument"); *n = a->n_opts; return BDR_OK; }

int32_t bdr_agent_sync(bdr_agent* a)
{
    BDR_REQUIRE(a, "null agent");
    BDR_HIP(hipSetDevice(a->device));
    BDR_HIP(hipStreamSynchronize(a->stream));
    BDR_TRY(a->after_sync());
    return a->err_check();
}

int32_t bdr_agent_opt(bdr_agent* a, bdr_replay* r)
{
    BDR_REQUIRE(a && r, "null argument");
    BDR_HIP(hipSetDevice(a->device));
    BDR_TRY(a->err_poll());   // asynchronous: a device-side failure of an earlier step surfaces here without a synchronisation
    a->last_replay_uid = r->uid;
    BDR_TRY(a->opt(r));
    prof_collect(a);
    return BDR_OK;
}

int32_t bdr_agent_opt_with_record(bdr_agent* a, bdr_replay* r, bdr_dqn_record* rec)
{
    BDR_REQUIRE(a && r && rec, "null argument");
    BDR_REQUIRE(is_dqn(a), "bdr_agent_opt_with_record(bdr_dqn_record) needs a DQN agent; use bdr_agent_opt_with_scalars");
    BDR_HIP(hipSetDevice(a->device));
    BDR_TRY(a->err_poll());   // as bdr_agent_opt: a failure of an earlier step is reported before this one is enqueued
    a->last_replay_uid = r->uid;
    BDR_TRY(a->opt(r));
    prof_collect(a);
    float v[128]; int n = 0;
    a->rec_opt = true;
    const int32_t st = a->record(v, 128, &n);
    a->rec_opt = false;
    BDR_TRY(st);
    rec->loss = v[0]; rec->has_verbose = n >= 5;   // the record is complete whatever the error words say about the step that produced it
    if (n >= 5) { rec->pred_mean = v[1]; rec->reward_mean = v[2]; rec->tgt_mean = v[3]; rec->tgt_minus_pred_mean = v[4]; }
    return err_check_after_step(a);
}

int32_t bdr_agent_opt_with_scalars(bdr_agent* a, bdr_replay* r, float* out, int32_t cap, int32_t* n_out)
{
    BDR_REQUIRE(a && r && out && n_out, "null argument");
    BDR_HIP(hipSetDevice(a->device));
    BDR_TRY(a->err_poll());
    a->last_replay_uid = r->uid;
    BDR_TRY(a->opt(r));
    prof_collect(a);
    int n = 0;
    a->rec_opt = true;
    const int32_t st = a->record(out, cap, &n);
    a->rec_opt = false;
    BDR_TRY(st);
    *n_out = n;   // the record is complete whatever the error words say about the step that produced it
    return err_check_after_step(a);
}

// names of the scalars bdr_agent_opt_with_scalars returns, '\n'-separated, in order
int32_t bdr_agent_record_keys(bdr_agent* a, char* names_out, uint64_t names_cap, int32_t* n_keys)
{
    BDR_REQUIRE(a && names_out && names_cap > 0, "null argument");
    std::vector<std::string> keys;
    a->record_keys(keys);
    std::string all;
    for (const auto& k : keys) { all += k; all += '\n'; }
    BDR_REQUIRE(all.size() < names_cap, "names_cap too small (%llu needed)", (unsigned long long)all.size() + 1);
    memcpy(names_out, all.c_str(), all.size() + 1);
    if (n_keys) *n_keys = (int32_t)keys.size();
    return BDR_OK;
}

// test helper: n draws of the agent's device noise stream, copied to the host (advances the stream like an update does)
int32_t bdr_agent_draw_noise(bdr_agent* a, uint64_t n, float* out)
{
    BDR_REQUIRE(a && out && n >= 1 && n <= (1ull << 28), "bad argument");
    BDR_HIP(hipSetDevice(a->device));
    float* d = nullptr;
    BDR_HIP(hipMalloc((void**)&d, n * 4));
    int32_t st = a->noise(d, n);
    if (st == BDR_OK) {
        hipError_t e = hipMemcpyAsync(out, d, n * 4, hipMemcpyDeviceToHost, a->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(a->stream);
        if (e != hipSuccess) st = fail(BDR_ERR_HIP, "noise copy failed: %s", hipGetErrorString(e));
    } else {
        (void)hipStreamSynchronize(a->stream);
    }
    (void)hipFree(d);
    return st;
}

int32_t bdr_dqn_update_on_batch(bdr_agent* a, uint64_t n, const void* obs, const int64_t* act, const void* next_obs,
                                const float* reward, const int8_t* term, bdr_dqn_record* rec)
{
    return bdr_dqn_update_on_batch_weighted(a, n, obs, act, next_obs, reward, term, nullptr, nullptr, rec);
}

int32_t bdr_dqn_update_on_batch_weighted(bdr_agent* a, uint64_t n, const void* obs, const int64_t* act, const void* next_obs,
                                         const float* reward, const int8_t* term, const float* weight, float* td_errs_out,
                                         bdr_dqn_record* rec)
{
    BDR_REQUIRE(a && obs && act && next_obs && reward && term, "null argument");
    BDR_REQUIRE(is_dqn(a), "not a DQN agent");
    BDR_REQUIRE(n >= 1 && n <= 65536, "batch size out of range");
    BDR_HIP(hipSetDevice(a->device));
    if (!strcmp(a->kind(), "dqn_cnn")) BDR_TRY(dqn_cnn_update_on_batch(a, n, obs, act, next_obs, reward, term, weight));
    else BDR_TRY(dqn_mlp_update_on_batch(a, n, obs, act, next_obs, reward, term, weight));
    if (td_errs_out) {
        BDR_HIP(hipMemcpyAsync(td_errs_out, a->td_abs, n * 4, hipMemcpyDeviceToHost, a->stream));
        BDR_HIP(hipStreamSynchronize(a->stream));
    }
    prof_collect(a);
    BDR_TRY(a->err_check());   // (update_on_batch synchronises)
    if (rec) {
        float v[128]; int k = 0;
        BDR_TRY(a->record(v, 128, &k));
        rec->loss = v[0]; rec->has_verbose = k >= 5;
        if (k >= 5) { rec->pred_mean = v[1]; rec->reward_mean = v[2]; rec->tgt_mean = v[3]; rec->tgt_minus_pred_mean = v[4]; }
    }
    return BDR_OK;
}

// update_critic up to loss.backward() on a host minibatch (gradients -> arena 4; parameters, moments, counters untouched)
int32_t bdr_dqn_grads_on_batch(bdr_agent* a, uint64_t n, const void* obs, const int64_t* act, const void* next_obs,
                               const float* reward, const int8_t* term, bdr_dqn_record* rec)
{
    BDR_REQUIRE(a && obs && act && next_obs && reward && term, "null argument");
    BDR_REQUIRE(is_dqn(a), "not a DQN agent");
    BDR_REQUIRE(n >= 1 && n <= 65536, "batch size out of range");
    BDR_HIP(hipSetDevice(a->device));
    BDR_TRY(a->grads_on_batch(n, obs, act, next_obs, reward, term));
    prof_collect(a);
    BDR_TRY(a->err_check());
    if (rec) {
        float v[128]; int k = 0;
        BDR_TRY(a->record(v, 128, &k));
        rec->loss = v[0]; rec->has_verbose = k >= 5;
        if (k >= 5) { rec->pred_mean = v[1]; rec->reward_mean = v[2]; rec->tgt_mean = v[3]; rec->tgt_minus_pred_mean = v[4]; }
    }
    return BDR_OK;
}

// the optimizer step (opt.rs:74-83 `step`) on the gradient arena + the bookkeeping of opt_ (dqn/base.rs:190-198)
int32_t bdr_agent_apply_grads(bdr_agent* a)
{
    BDR_REQUIRE(a, "null agent");
    BDR_HIP(hipSetDevice(a->device));
    BDR_TRY(a->apply_grads());
    prof_collect(a);
    return BDR_OK;
}

// action values [n][A] of any value-based agent, on the host
static int32_t action_values(bdr_agent* a, uint64_t n, const void* obs, std::vector<float>& q, int* A_out)
{
    BDR_REQUIRE(n >= 1 && n <= 65536, "batch size out of range");
    BDR_HIP(hipSetDevice(a->device));
    const int A = (int)a->param_count(-1);   // number of actions (every value agent reports it as which = -1)
    q.resize(n * A);
    if (!strcmp(a->kind(), "dqn_cnn")) BDR_TRY(dqn_cnn_qvalues(a, n, obs, q.data()));
    else if (!strcmp(a->kind(), "dqn_mlp")) BDR_TRY(dqn_mlp_qvalues(a, n, obs, q.data()));
    else if (!strcmp(a->kind(), "iqn")) BDR_TRY(bdr_iqn_qvalues(a, n, obs, q.data(), nullptr));
    else return fail(BDR_ERR_INVALID, "agent kind '%s' has no discrete action values", a->kind());
    *A_out = A;
    return BDR_OK;
}

static int argmax_row(const float* q, int A)   // first maximum, like Tensor::argmax
{
    int best = 0;
    for (int k = 1; k < A; ++k) if (q[k] > q[best]) best = k;
    return best;
}

int32_t bdr_agent_qvalues(bdr_agent* a, uint64_t n, const void* obs, float* q_out, int64_t* argmax_out)
{
    BDR_REQUIRE(a && obs, "null argument");
    std::vector<float> q;
    int A = 0;
    BDR_TRY(action_values(a, n, obs, q, &A));
    if (q_out) memcpy(q_out, q.data(), q.size() * 4);
    if (argmax_out)
        for (uint64_t i = 0; i < n; ++i) argmax_out[i] = argmax_row(&q[i * A], A);
    return BDR_OK;
}

void bdr_explorer_config_default(bdr_explorer_config* e, int32_t kind)
{
    if (!e) return;
    e->kind = kind;
    e->eps_start = 1.0; e->eps_final = 0.02; e->final_step = 100000;   // dqn/explorer.rs:44-52
    e->n_calls = 0; e->seed = 0;
}

int32_t bdr_agent_set_explorer(bdr_agent* a, const bdr_explorer_config* e)
{
    BDR_REQUIRE(a && e, "null argument");
    BDR_REQUIRE(e->kind == BDR_EXPLORER_SOFTMAX || e->kind == BDR_EXPLORER_EPS_GREEDY, "unknown explorer kind");
    BDR_REQUIRE(e->kind != BDR_EXPLORER_EPS_GREEDY || e->final_step > 0, "final_step must be positive");
    Explorer& x = a->explorer;
    x.kind = e->kind; x.eps_start = e->eps_start; x.eps_final = e->eps_final; x.final_step = e->final_step;
    x.n_calls = e->n_calls;
    seed_from_u64(e->seed, x.key.k);
    x.word_pos = 0;
    return BDR_OK;
}

int32_t bdr_agent_get_explorer(const bdr_agent* a, bdr_explorer_config* e)
{
    BDR_REQUIRE(a && e, "null argument");
    const Explorer& x = a->explorer;
    e->kind = x.kind; e->eps_start = x.eps_start; e->eps_final = x.eps_final; e->final_step = x.final_step;
    e->n_calls = x.n_calls; e->seed = 0;   // the seed itself is not retained (only the expanded key)
    return BDR_OK;
}

// Policy::sample (dqn/base.rs:211-242; iqn/base.rs:204-228)
int32_t bdr_agent_sample(bdr_agent* a, uint64_t n, const void* obs, int64_t* act_out, bdr_sample_info* info)
{
    BDR_REQUIRE(a && obs && act_out, "null argument");
    std::vector<float> q;
    int A = 0;
    a->err_fresh = false;   // (set again only by THIS call's own pinned read-back: a flag left by an earlier qvalues call must not send this one down the poll on old words)
    BDR_TRY(action_values(a, n, obs, q, &A));
    if (a->err_fresh) {   // the forward pass brought the device's error words along (rows_wait): report from them, no second copy;
        a->err_fresh = false;   // the buffer's priority flag (a NaN handed to update_priority) is a host-side word and is looked at either way
        BDR_TRY(a->err_poll());
        bool alive = false;
        const int32_t st = replay_per_check(a->last_replay_uid, &alive);
        if (!alive) a->last_replay_uid = 0;
        BDR_TRY(st);
    } else BDR_TRY(a->err_check());
    Explorer& x = a->explorer;
    double eps = 0.0;
    bool is_random = false;
    if (a->train) {
        a->n_samples_act += 1;                                            // dqn/base.rs:215
        if (x.kind == BDR_EXPLORER_SOFTMAX) {                             // explorer.rs:29-31: softmax(-1).multinomial(1)
            bool all_best = true;
            for (uint64_t i = 0; i < n; ++i) {
                const float* qi = &q[i * A];
                float mx = qi[0];
                for (int k = 1; k < A; ++k) mx = std::max(mx, qi[k]);
                double z = 0.0;
                for (int k = 0; k < A; ++k) z += std::exp((double)(qi[k] - mx));
                const double u = x.f64() * z;
                double c = 0.0;
                int act = A - 1;
                for (int k = 0; k < A; ++k) {
                    c += std::exp((double)(qi[k] - mx));
                    if (c > u) { act = k; break; }
                }
                act_out[i] = act;
                all_best = all_best && act == argmax_row(qi, A);
            }
            if (all_best) a->n_samples_best_act += 1;
        } else {                                                          // explorer.rs:68-90
            const double d = (x.eps_start - x.eps_final) / (double)x.final_step;
            eps = std::max(x.eps_start - d * (double)x.n_calls, x.eps_final);
            is_random = x.f64() < eps;
            x.n_calls += 1;
            bool all_best = true;
            for (uint64_t i = 0; i < n; ++i) {
                const int best = argmax_row(&q[i * A], A);
                const int act = is_random ? (int)x.below((uint32_t)A) : best;
                act_out[i] = act;
                all_best = all_best && act == best;
            }
            if (all_best) a->n_samples_best_act += 1;                     // action_with_best, dqn/base.rs:219-224
        }
    } else {
        // eval: DQN takes a uniformly random action 1 % of the time (dqn/base.rs:231-234; the reference
        // returns ONE scalar action for the call, applied to every row here); IQN is purely greedy
        const bool dqn = !strncmp(a->kind(), "dqn", 3);
        if (dqn && x.f32() < 0.01f) {
            is_random = true;
            const int act = (int)x.below((uint32_t)A);
            for (uint64_t i = 0; i < n; ++i) act_out[i] = act;
        } else {
            for (uint64_t i = 0; i < n; ++i) act_out[i] = argmax_row(&q[i * A], A);
        }
    }
    if (info) {
        info->eps = eps; info->is_random = is_random ? 1 : 0;
        info->n_samples_act = a->n_samples_act; info->n_samples_best_act = a->n_samples_best_act;
    }
    return BDR_OK;
}

// Policy::sample / action values for observation rows that already live in HBM ("batched across many vectorised
// envs on device"): row i at obs_dev + i * row_stride bytes, e.g. the frame stacks of a bdr_atari_prep.  Same forward, same
// exploration stream, same counters as the host-row calls - only the host -> device copy of the rows is gone (contiguous rows are
// read in place).  The rows must be complete when the call is made (their producer synchronised, as bdr_atari_prep_step does) and
// stay untouched until it returns.
int32_t bdr_agent_sample_device(bdr_agent* a, uint64_t n, const void* obs_dev, uint64_t row_stride, int64_t* act_out, bdr_sample_info* info)
{
    BDR_REQUIRE(a && act_out, "null argument");
    BDR_HIP(hipSetDevice(a->device));
    BDR_TRY(a->check_device_rows(obs_dev, row_stride));
    bdr_agent::DeviceRowsScope rows(a, row_stride);
    return bdr_agent_sample(a, n, obs_dev, act_out, info);
}

int32_t bdr_agent_qvalues_device(bdr_agent* a, uint64_t n, const void* obs_dev, uint64_t row_stride, float* q_out, int64_t* argmax_out)
{
    BDR_REQUIRE(a, "null argument");
    BDR_HIP(hipSetDevice(a->device));
    BDR_TRY(a->check_device_rows(obs_dev, row_stride));
    bdr_agent::DeviceRowsScope rows(a, row_stride);
    return bdr_agent_qvalues(a, n, obs_dev, q_out, argmax_out);
}

int32_t bdr_agent_param_count(const bdr_agent* a, uint64_t* n)
{
    BDR_REQUIRE(a && n, "null argument");
    *n = const_cast<bdr_agent*>(a)->param_count(0);
    return BDR_OK;
}

int32_t bdr_agent_param_count_of(bdr_agent* a, int32_t which, uint64_t* n)
{
    BDR_REQUIRE(a && n, "null argument");
    *n = a->param_count(which);
    return BDR_OK;
}

int32_t bdr_agent_get_params(bdr_agent* a, int32_t which, float* out, uint64_t n)
{
    BDR_REQUIRE(a && out, "null argument");
    BDR_HIP(hipSetDevice(a->device));
    BDR_TRY(a->get_params(which, out, n));
    return a->err_check();
}

int32_t bdr_agent_set_params(bdr_agent* a, int32_t which, const float* inp, uint64_t n)
{
    BDR_REQUIRE(a && inp, "null argument");
    BDR_HIP(hipSetDevice(a->device));
    return a->set_params(which, inp, n);
}

int32_t bdr_agent_arena_device_ptr(bdr_agent* a, int32_t which, void** ptr, uint64_t* n_floats)
{
    BDR_REQUIRE(a && ptr && n_floats, "null argument");
    size_t n = 0;
    float* p = a->arena(which, &n);
    BDR_REQUIRE(p, "unknown arena %d", which);
    a->arena_escaped(which);   // the caller may write through it whenever they like: derived copies of these parameters are never trusted again
    *ptr = p; *n_floats = n;
    return BDR_OK;
}

int32_t bdr_agent_arena_release(bdr_agent* a, int32_t which)
{
    BDR_REQUIRE(a, "null argument");
    size_t n = 0;
    BDR_REQUIRE(a->arena(which, &n), "unknown arena %d", which);   // (joins a pending exchange and marks derived copies stale: the last write may be recent)
    a->arena_released(which);
    return BDR_OK;
}

int32_t bdr_agent_set_checkpoint_format(bdr_agent* a, int32_t format)
{
    BDR_REQUIRE(a, "null argument");
    BDR_REQUIRE(format == BDR_CKPT_TCH || format == BDR_CKPT_SAFETENSORS, "unknown checkpoint format");
    a->ckpt_format = format;
    return BDR_OK;
}

static int32_t named_meta(const bdr_named_tensor* meta, uint32_t n_tensors, std::vector<NamedTensor>& out)
{
    for (uint32_t t = 0; t < n_tensors; ++t) {
        BDR_REQUIRE(meta[t].name && (meta[t].dims || meta[t].ndim == 0), "null tensor name or dims");
        NamedTensor nt;
        nt.name = meta[t].name;
        nt.dims.assign(meta[t].dims, meta[t].dims + meta[t].ndim);
        out.push_back(std::move(nt));
    }
    return BDR_OK;
}

int32_t bdr_checkpoint_write(const char* path, const bdr_named_tensor* meta, uint32_t n_tensors, const float* data, uint64_t n)
{
    BDR_REQUIRE(path && meta && (data || n == 0), "null argument");
    std::vector<NamedTensor> m;
    BDR_TRY(named_meta(meta, n_tensors, m));
    return save_named(path, m, data, n);
}

int32_t bdr_checkpoint_read(const char* path, const bdr_named_tensor* meta, uint32_t n_tensors, float* data, uint64_t n)
{
    BDR_REQUIRE(path && meta && (data || n == 0), "null argument");
    std::vector<NamedTensor> m;
    BDR_TRY(named_meta(meta, n_tensors, m));
    return load_named(path, m, data, n);
}

int32_t bdr_agent_save_params(bdr_agent* a, const char* dir)
{
    BDR_REQUIRE(a && dir, "null argument");
    BDR_HIP(hipSetDevice(a->device));
    BDR_HIP(hipStreamSynchronize(a->stream));
    BDR_TRY(a->err_check());               // never write a checkpoint over a flagged state
    BDR_TRY(create_dir_all(dir));          // fs::create_dir_all(&path) (dqn/base.rs:346)
    return a->save(dir);
}

int32_t bdr_agent_load_params(bdr_agent* a, const char* dir)
{
    BDR_REQUIRE(a && dir, "null argument");
    BDR_HIP(hipSetDevice(a->device));
    return a->load(dir);
}

int32_t bdr_dqn_probe(bdr_agent* a, int32_t what, float* out, uint64_t n)
{
    BDR_REQUIRE(a && out, "null argument");
    BDR_REQUIRE(is_dqn(a), "not a DQN agent");
    BDR_HIP(hipSetDevice(a->device));
    return !strcmp(a->kind(), "dqn_cnn") ? dqn_cnn_probe(a, what, out, n) : dqn_mlp_probe(a, what, out, n);
}

int32_t bdr_agent_profile_enable(bdr_agent* a, int32_t on)
{
    BDR_REQUIRE(a, "null agent");
    BDR_HIP(hipSetDevice(a->device));
    BDR_HIP(hipStreamSynchronize(a->stream));
    a->prof = on != 0;
    a->slot_cursor = 0;
    for (auto& s : a->slots) { s.ms = 0; s.count = 0; }
    return BDR_OK;
}

int32_t bdr_agent_profile_read(bdr_agent* a, char* names_out, uint64_t names_cap, float* ms_out, uint64_t* count_inout)
{
    BDR_REQUIRE(a && count_inout, "null argument");
    std::string names;
    uint64_t k = 0;
    for (auto& s : a->slots) {
        if (k < *count_inout && ms_out) ms_out[k] = s.count ? (float)(s.ms / (double)s.count) : 0.f;
        names += s.name; names += '\n';
        ++k;
    }
    if (names_out && names_cap) { strncpy(names_out, names.c_str(), names_cap - 1); names_out[names_cap - 1] = 0; }
    *count_inout = k;
    return BDR_OK;
}

}  // extern "C"

// used by comm.hip
namespace bdr {
float* agent_arena(bdr_agent* a, int which, size_t* n_floats, hipStream_t* stream, int* device)
{
    if (stream) *stream = a->stream;
    if (device) *device = a->device;
    return a->arena(which, n_floats);
}
int32_t agent_scale(bdr_agent* a, float* p, size_t n, float s) { return launch_scale(a->stream, p, n, s); }
void agent_set_grad_comm(bdr_agent* a, void* comm, int32_t (*reduce)(bdr_agent*, void*)) { a->grad_comm = comm; a->grad_reduce = reduce; }
struct XSeg { size_t off, n; };
int agent_exchange_plan(bdr_agent* a, int which, XSeg* segs, int cap, hipStream_t* comm)
{
    bdr_agent::ExchangeSeg es[8];
    const int n = a->exchange_plan(which, es, std::min(cap, 8), comm);
    for (int k = 0; k < n; ++k) { segs[k].off = es[k].off; segs[k].n = es[k].n; }
    return n;
}
int32_t agent_exchange_begin(bdr_agent* a, int seg) { return a->exchange_begin(seg); }
int32_t agent_exchange_end(bdr_agent* a, int seg) { return a->exchange_end(seg); }
int32_t scale_on(hipStream_t st, float* p, size_t n, float s) { return launch_scale(st, p, n, s); }
}  // namespace bdr
