// HBM-resident SimpleReplayBuffer: fused record ring, on-device ChaCha12 index draw (bit-identical
// to rand 0.8.5 StdRng), coalesced row gather into batch layout.
// Reference: border-core/src/generic_replay_buffer/base.rs:86-123 (state), :295-316 (push),
// :376-402 (batch); border-tch-agent/src/tensor_batch.rs:85-120 (row storage).
#include "chacha.hpp"
#include <algorithm>
#include <chrono>
#include <atomic>
#include <mutex>
#include <unordered_map>
#include <unordered_set>

#include "common.hpp"
#include "step_graph.hpp"

namespace bdr {
thread_local char g_err[512] = "";
thread_local int g_err_deferred = 0;

int32_t ensure_device(int32_t device)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(BDR_ERR_NO_DEVICE, "no HIP device available (%s); libborder_amd has no CPU fallback",
                    e == hipSuccess ? "count=0" : hipGetErrorString(e));
    if (device < 0 || device >= n) return fail(BDR_ERR_INVALID, "device %d out of range [0,%d)", device, n);
    BDR_HIP(hipSetDevice(device));
    return BDR_OK;
}
}  // namespace bdr

using namespace bdr;

// ------------------------------------------------------------------------------------------------
// K1: ChaCha12 index kernel.  Word w of the StdRng stream = word (w % 16) of block (w / 16)
// (counter mode), so every index of a batch is independent: thread k computes its own block.
// ixs[k] = (u32 as usize) % size   -- base.rs:386, modulo bias kept.
// ------------------------------------------------------------------------------------------------
__global__ void k_sample_indices(ChaChaKey key, uint64_t word_pos, uint64_t size, uint32_t n,
                                 uint64_t* __restrict__ ixs)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    ixs[k] = (uint64_t)chacha12_word(key, word_pos + k) % size;
}

// ------------------------------------------------------------------------------------------------
// The optional device-native index generator (bdr_replay_config::index_rng = BDR_RNG_XOSHIRO256PP; north_star's wording): one
// xoshiro256++ generator (Blackman / Vigna) per batch lane, states in HBM.  k_xo_init: lane j <- outputs 4j .. 4j+3 of SplitMix64(seed)
// (the seeding its authors prescribe); k_xo_indices: lane j < n steps once, ix[j] = (result >> 32) % size.  Not the reference's StdRng
// stream - no parity claim - but pinned on its own: oracle.py restates it, tests compare index streams and the generator's known answers.
// ------------------------------------------------------------------------------------------------
constexpr uint32_t XO_LANES = 1u << 16;
__host__ __device__ inline uint64_t splitmix64_at(uint64_t seed, uint64_t n)   // output n (0-based) of SplitMix64 seeded with `seed`
{
    uint64_t z = seed + (n + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__global__ void k_xo_init(uint64_t* __restrict__ st, uint64_t seed, uint32_t lanes)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= lanes) return;
#pragma unroll
    for (int i = 0; i < 4; ++i) st[(size_t)j * 4 + i] = splitmix64_at(seed, (uint64_t)j * 4 + i);
}
__global__ void k_xo_indices(uint64_t* __restrict__ st, uint64_t size, uint32_t n, uint64_t* __restrict__ ixs)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    uint64_t s0 = st[(size_t)j * 4], s1 = st[(size_t)j * 4 + 1], s2 = st[(size_t)j * 4 + 2], s3 = st[(size_t)j * 4 + 3];
    const uint64_t sum = s0 + s3;
    const uint64_t result = ((sum << 23) | (sum >> 41)) + s0;      // rotl(s0 + s3, 23) + s0
    const uint64_t t = s1 << 17;
    s2 ^= s0; s3 ^= s1; s1 ^= s2; s0 ^= s3; s2 ^= t; s3 = (s3 << 45) | (s3 >> 19);
    st[(size_t)j * 4] = s0; st[(size_t)j * 4 + 1] = s1; st[(size_t)j * 4 + 2] = s2; st[(size_t)j * 4 + 3] = s3;
    ixs[j] = (uint64_t)(uint32_t)(result >> 32) % size;
}

// ------------------------------------------------------------------------------------------------
// K2: gather.  One workgroup copies one chunk of one sampled record's obs and next_obs with
// 16-byte lanes (a wave instruction moves 1 KiB of a contiguous row); the tail fields
// (act / reward / flags) of the record are transposed into the SoA batch arrays by chunk 0.
// Algorithmic bytes per sample: 2*obs_bytes + act_bytes + 6.
// ------------------------------------------------------------------------------------------------
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <typename V>
__global__ __launch_bounds__(256) void k_gather(GatherArgs a)
{
    const uint32_t sample = blockIdx.x / a.chunks, chunk = blockIdx.x % a.chunks;
    // ixs[k] = (StdRng::next_u32() as usize) % size   (base.rs:386): word k of this batch's key stream
    const uint64_t row = a.given ? a.ixs[sample] : (uint64_t)chacha12_word(a.key, a.word_pos + sample) % a.size;
    if (!a.given && chunk == 0 && threadIdx.x == 0) a.ixs[sample] = row;
    const uint8_t* rec = a.ring + row * a.stride;
    const uint64_t nvec = a.obs_bytes / sizeof(V);
    const V* src0 = reinterpret_cast<const V*>(rec);
    const V* src1 = reinterpret_cast<const V*>(rec + a.next_off);
    V* dst0 = reinterpret_cast<V*>(a.b_obs + (uint64_t)sample * a.obs_bytes);
    V* dst1 = reinterpret_cast<V*>(a.b_next + (uint64_t)sample * a.obs_bytes);
    const uint64_t v0 = (uint64_t)chunk * a.vec_per_chunk;
    const uint64_t v1 = min(v0 + a.vec_per_chunk, nvec);
    // the record's small fields are requested now, next to the first row loads: as load -> store pairs after the copy loop they
    // were three dependent HBM round trips at the tail of every sample's first workgroup
    uint8_t t_act = 0; float t_rew = 0.f; int8_t t_term = 0, t_trunc = 0;
    const bool tail = chunk == 0;
    if (tail) {
        if (threadIdx.x < a.act_bytes) t_act = rec[a.act_off + threadIdx.x];
        if (threadIdx.x == 0) {
            t_rew = *reinterpret_cast<const float*>(rec + a.tail_off);
            t_term = *reinterpret_cast<const int8_t*>(rec + a.tail_off + 4);
            t_trunc = *reinterpret_cast<const int8_t*>(rec + a.tail_off + 5);
        }
    }
    // 4 vectors of each section per thread and pass: all 8 loads are issued before the first store, so a
    // thread keeps 8 x 16 B in flight (the row addresses are random, every load is an HBM round trip)
    for (uint64_t v = v0 + threadIdx.x; v < v1; v += 1024) {
        V x[4], y[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint64_t w = min(v + (uint64_t)u * 256, v1 - 1);   // clamped: tail lanes re-read the last vector
            x[u] = __builtin_nontemporal_load(src0 + w);
            y[u] = __builtin_nontemporal_load(src1 + w);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint64_t w = v + (uint64_t)u * 256;
            if (w < v1) { dst0[w] = x[u]; dst1[w] = y[u]; }
        }
    }
    if (tail) {
        if (threadIdx.x < a.act_bytes) a.b_act[(uint64_t)sample * a.act_bytes + threadIdx.x] = t_act;
        for (uint32_t t = threadIdx.x + 256; t < a.act_bytes; t += 256)      // (actions longer than 256 bytes)
            a.b_act[(uint64_t)sample * a.act_bytes + t] = rec[a.act_off + t];
        if (threadIdx.x == 0) { a.b_reward[sample] = t_rew; a.b_term[sample] = t_term; a.b_trunc[sample] = t_trunc; }
    }
}

// K2 for the single-frame store: the record of a sampled transition names 2k frame slots; workgroup (sample, half) copies the
// k frames of obs (half 0) or next_obs (half 1) out of the frame store into the stacked batch row - the batch layout (and
// therefore every consumer) is the plain ring's.  Algorithmic bytes per sample are the same 2 * obs_bytes.
struct GatherFramesArgs {
    const uint8_t* recs; uint64_t rec_stride, rec_act_off, rec_tail_off, act_bytes;
    const uint8_t* frames; uint64_t frame_bytes; int k;
    uint64_t* ixs; ChaChaKey key; uint64_t word_pos, size;
    uint8_t *b_obs, *b_next, *b_act; float* b_reward; int8_t *b_term, *b_trunc;
    uint32_t given;
};
__global__ __launch_bounds__(256) void k_gather_frames(GatherFramesArgs a)
{
    const uint32_t sample = blockIdx.x >> 1, half = blockIdx.x & 1;
    const uint64_t row = a.given ? a.ixs[sample] : (uint64_t)chacha12_word(a.key, a.word_pos + sample) % a.size;
    if (!a.given && half == 0 && threadIdx.x == 0) a.ixs[sample] = row;
    const uint8_t* rec = a.recs + row * a.rec_stride;
    const uint32_t* slots = reinterpret_cast<const uint32_t*>(rec) + half * a.k;
    const uint64_t fv = a.frame_bytes / 16, nvec = fv * a.k;          // 16-byte vectors per frame / per stacked row
    u32x4* dst = reinterpret_cast<u32x4*>((half ? a.b_next : a.b_obs) + (uint64_t)sample * a.frame_bytes * a.k);
    for (uint64_t v = threadIdx.x; v < nvec; v += 1024) {
        u32x4 x[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint64_t w = min(v + (uint64_t)u * 256, nvec - 1);
            const uint64_t f = w / fv, o = w % fv;
            x[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(a.frames + (uint64_t)slots[f] * a.frame_bytes) + o);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint64_t w = v + (uint64_t)u * 256;
            if (w < nvec) dst[w] = x[u];
        }
    }
    if (half == 0) {
        for (uint32_t t = threadIdx.x; t < a.act_bytes; t += 256) a.b_act[(uint64_t)sample * a.act_bytes + t] = rec[a.rec_act_off + t];
        if (threadIdx.x == 0) {
            a.b_reward[sample] = *reinterpret_cast<const float*>(rec + a.rec_tail_off);
            a.b_term[sample] = *reinterpret_cast<const int8_t*>(rec + a.rec_tail_off + 4);
            a.b_trunc[sample] = *reinterpret_cast<const int8_t*>(rec + a.rec_tail_off + 5);
        }
    }
}

// synthetic fill of the single-frame store: one continuous "episode" - frame sequence number q holds the counter-based bytes of
// (seed, q), transition t is obs = frames (t+k-1 .. t), next_obs = (t+k .. t+1) (newest first), side fields as in k_fill_synthetic
struct FillFramesArgs {
    uint8_t* recs; uint64_t rec_stride, rec_act_off, rec_tail_off; uint8_t* frames; uint64_t frame_bytes, frame_cap;
    uint64_t n, seed, capacity; int k, n_actions;
};
__host__ __device__ __forceinline__ uint64_t synth_hash(uint64_t seed, uint64_t t, uint64_t sec, uint64_t w);
__global__ __launch_bounds__(256) void k_fill_frames(FillFramesArgs a)
{
    const uint64_t q = blockIdx.x;                       // frame sequence number, q < n + k
    uint8_t* dst = a.frames + (q % a.frame_cap) * a.frame_bytes;
    for (uint64_t w = threadIdx.x; w < a.frame_bytes / 8; w += 256) reinterpret_cast<uint64_t*>(dst)[w] = synth_hash(a.seed, q, 7, w);
    if (q >= a.n || threadIdx.x != 0) return;
    const uint64_t t = q;
    uint8_t* rec = a.recs + (t % a.capacity) * a.rec_stride;
    uint32_t* slots = reinterpret_cast<uint32_t*>(rec);
    for (int j = 0; j < a.k; ++j) {
        slots[j] = (uint32_t)((t + a.k - 1 - j) % a.frame_cap);
        slots[a.k + j] = (uint32_t)((t + a.k - j) % a.frame_cap);
    }
    const uint64_t h = synth_hash(a.seed, t, 2, 0);
    int64_t act = (int64_t)((h & 0xFFFFFFFFull) % (uint64_t)a.n_actions);
    memcpy(rec + a.rec_act_off, &act, 8);
    const uint32_t u = (uint32_t)(h >> 40);
    const float reward = u < 838861u ? -1.0f : (u < 15938355u ? 0.0f : 1.0f);
    memcpy(rec + a.rec_tail_off, &reward, 4);
    rec[a.rec_tail_off + 4] = (synth_hash(a.seed, t, 2, 1) >> 40) < 83886u ? 1 : 0;
    rec[a.rec_tail_off + 5] = 0;
}

// ------------------------------------------------------------------------------------------------
// Synthetic fill (benchmark input).  Counter-based: 64-bit word w of
// section s of transition t = splitmix64-finalised hash of (seed, t, s, w), so a host restatement
// (tests/synth.py) reproduces the ring bit for bit.
// ------------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ uint64_t synth_hash(uint64_t seed, uint64_t t, uint64_t sec, uint64_t w)
{
    uint64_t x = (seed + 1) * 0x9E3779B97F4A7C15ull;
    x ^= (t + 1) * 0xBF58476D1CE4E5B9ull;
    x ^= ((sec << 40) | w) * 0x94D049BB133111EBull;
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27; x *= 0x94D049BB133111EBull;
    x ^= x >> 31;
    return x;
}

__device__ __forceinline__ float synth_normal(uint64_t h)
{
    // Irwin-Hall(4) of 16-bit uniforms, centred and scaled to unit variance
    float u = (float)(h & 0xFFFF) + (float)((h >> 16) & 0xFFFF) + (float)((h >> 32) & 0xFFFF) + (float)(h >> 48);
    return (u * (1.0f / 65536.0f) - 2.0f) * 1.7320508f;
}

struct FillArgs {
    uint8_t* ring;
    uint64_t stride, obs_bytes, act_bytes, next_off, act_off, tail_off, capacity;
    uint64_t n, seed;
    int32_t kind, n_actions;
};

__global__ __launch_bounds__(256) void k_fill_synthetic(FillArgs a)
{
    const uint64_t t = blockIdx.x;  // transition index (t < n <= capacity handled by the host loop)
    uint8_t* rec = a.ring + (t % a.capacity) * a.stride;
    const uint64_t nw = a.obs_bytes / 8, rem = a.obs_bytes % 8;
    for (int sec = 0; sec < 2; ++sec) {
        uint8_t* dst = rec + (sec ? a.next_off : 0);
        if (a.kind == 0) {
            for (uint64_t w = threadIdx.x; w < nw; w += 256)
                reinterpret_cast<uint64_t*>(dst)[w] = synth_hash(a.seed, t, sec, w);
            if (threadIdx.x == 0 && rem) {
                uint64_t h = synth_hash(a.seed, t, sec, nw);
                for (uint64_t b = 0; b < rem; ++b) dst[nw * 8 + b] = (uint8_t)(h >> (8 * b));
            }
        } else {
            for (uint64_t w = threadIdx.x; w < a.obs_bytes / 4; w += 256)
                reinterpret_cast<float*>(dst)[w] = synth_normal(synth_hash(a.seed, t, sec, w));
        }
    }
    if (threadIdx.x == 0) {
        const uint64_t h = synth_hash(a.seed, t, 2, 0);
        if (a.n_actions > 0) {
            int64_t act = (int64_t)((h & 0xFFFFFFFFull) % (uint64_t)a.n_actions);
            memcpy(rec + a.act_off, &act, 8);
        } else {
            for (uint64_t w = 0; w < a.act_bytes / 4; ++w) {
                uint64_t g = synth_hash(a.seed, t, 3, w);
                float v = (float)(g >> 40) * (2.0f / 16777216.0f) - 1.0f;
                memcpy(rec + a.act_off + 4 * w, &v, 4);
            }
        }
        float reward;
        const int8_t term = (synth_hash(a.seed, t, 2, 1) >> 40) < 83886u ? 1 : 0;  // .005 of 2^24
        if (a.kind == 0) {
            const uint32_t u = (uint32_t)(h >> 40);  // 24 bits
            reward = u < 838861u ? -1.0f : (u < 15938355u ? 0.0f : 1.0f);  // .05 / .90 / .05
        } else {
            reward = synth_normal(synth_hash(a.seed, t, 2, 2));
        }
        memcpy(rec + a.tail_off, &reward, 4);
        rec[a.tail_off + 4] = (uint8_t)term;
        rec[a.tail_off + 5] = 0;
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
namespace bdr {
static std::mutex g_streams_mu;
static std::unordered_set<hipStream_t> g_streams;
void stream_register(hipStream_t s) { std::lock_guard<std::mutex> l(g_streams_mu); g_streams.insert(s); }
void stream_retire(hipStream_t s) { std::lock_guard<std::mutex> l(g_streams_mu); g_streams.erase(s); }
bool stream_alive(hipStream_t s) { std::lock_guard<std::mutex> l(g_streams_mu); return g_streams.count(s) != 0; }
// Live replay handles by uid: an agent remembers the BUFFER OF ITS LAST OPT by uid, never by pointer (the buffer may be
// destroyed before the agent's next synchronising call: train -> rb.close() -> agent.sample() for evaluation).
static std::mutex g_replays_mu;
static std::unordered_map<uint64_t, bdr_replay*> g_replays;
static void replay_register(bdr_replay* r) { std::lock_guard<std::mutex> l(g_replays_mu); g_replays[r->uid] = r; }
static void replay_retire(bdr_replay* r) { std::lock_guard<std::mutex> l(g_replays_mu); g_replays.erase(r->uid); }
bdr_replay* replay_lookup(uint64_t uid)
{
    if (!uid) return nullptr;
    std::lock_guard<std::mutex> l(g_replays_mu);
    auto it = g_replays.find(uid);
    return it == g_replays.end() ? nullptr : it->second;
}
// per_check of the live buffer with this uid, under the registry lock: a concurrent bdr_replay_destroy on another thread (the Rust
// handles are Send) cannot free the handle between the lookup and the dereference.  *alive = false once the buffer is gone.
int32_t replay_per_check(uint64_t uid, bool* alive)
{
    *alive = false;
    if (!uid) return BDR_OK;
    std::lock_guard<std::mutex> l(g_replays_mu);
    auto it = g_replays.find(uid);
    if (it == g_replays.end()) return BDR_OK;
    *alive = true;
    return it->second->per ? per_check(it->second->per) : BDR_OK;
}
}  // namespace bdr

// `written` was just recorded on some stream: consumers have to wait for it again
static void mark_written(bdr_replay* r) { r->written_gen += 1; }

// WAR / WAW against the last gather: the stream `s` is about to overwrite ring rows or the batch buffers
static int32_t wait_for_reader(bdr_replay* r, hipStream_t s)
{
    if (!r->read_pending) return BDR_OK;
    r->read_pending = false;
    if (r->read_stream == s) return BDR_OK;   // same queue: already ordered
    // a retired stream was synchronised by its owner before it was destroyed: nothing left to wait for
    if (r->read_stream != r->stream && !stream_alive(r->read_stream)) return BDR_OK;
    if (hipEventRecord(r->read, r->read_stream) != hipSuccess) {
        (void)hipGetLastError();
        BDR_HIP(hipDeviceSynchronize());
        return BDR_OK;
    }
    BDR_HIP(hipStreamWaitEvent(s, r->read, 0));
    return BDR_OK;
}

// RAW against pushes / fills / tree updates: once per consumer stream and generation of `written`
static int32_t wait_for_writer(bdr_replay* r, hipStream_t s)
{
    if (r->written_lazy) {   // a small device push left the record of `written` to its first consumer: on the writer's stream now, it covers that push
        r->written_lazy = false;
        BDR_HIP(hipEventRecord(r->written, r->stream));
    }
    for (auto& w : r->waited)
        if (w.first == s) {
            if (w.second == r->written_gen) return BDR_OK;
            w.second = r->written_gen;
            BDR_HIP(hipStreamWaitEvent(s, r->written, 0));
            return BDR_OK;
        }
    if (r->waited.size() >= 16) r->waited.erase(r->waited.begin());
    r->waited.emplace_back(s, r->written_gen);
    BDR_HIP(hipStreamWaitEvent(s, r->written, 0));
    return BDR_OK;
}

// push() of the single-frame store.  Sharing is FOUND, not assumed: frame j of an incoming stack reuses an already stored frame
// only when its bytes equal that frame's bytes -
//   obs_t == the previous push's next_obs (episode continues)      -> obs costs nothing
//   next_obs_t[1..k) == obs_t[0..k-1) (stack_frame, env.rs:197-209) -> next_obs costs its newest frame only
//   equal neighbouring frames inside one stack (reset: all k slots hold the first frame, env.rs:263-296) are stored once
// - so whatever rows are pushed come back bit for bit.  Frames are allocated in push order in a ring of frame_cap slots; a slot
// may only be reused once no live transition references it (first_ref), otherwise the push fails loudly.
static int32_t push_frames(bdr_replay* r, uint64_t n, const uint8_t* o, const uint8_t* a, const uint8_t* x, const float* reward,
                           const int8_t* term, const int8_t* trunc)
{
    const int k = r->frame_stack;
    const uint64_t fb = r->frame_bytes;
    std::vector<uint64_t> oseq(k), nseq(k);
    uint64_t staged = 0, stage_first_seq = r->frame_seq, rec_done = 0;
    auto flush_frames = [&]() -> int32_t {   // staged frames -> their slots (<= 2 runs: the store is a ring)
        uint64_t off = 0;
        while (off < staged) {
            const uint64_t slot = (stage_first_seq + off) % r->frame_cap;
            const uint64_t m = std::min(staged - off, r->frame_cap - slot);
            BDR_HIP(hipMemcpyAsync(r->frames + slot * fb, r->fstage + off * fb, m * fb, hipMemcpyHostToDevice, r->stream));
            off += m;
        }
        if (staged) BDR_HIP(hipStreamSynchronize(r->stream));   // staging reusable
        stage_first_seq = r->frame_seq; staged = 0;
        return BDR_OK;
    };
    auto flush_records = [&](uint64_t upto) -> int32_t {       // packed records [rec_done, upto) -> ring (<= 2 runs)
        uint64_t s0 = rec_done;
        while (s0 < upto) {
            const uint64_t pos = (r->i + s0) % r->capacity;
            const uint64_t m = std::min(std::min(upto - s0, r->capacity - pos), r->stage_records);
            BDR_HIP(hipMemcpyAsync(r->ring + pos * r->stride, r->stage + (s0 - rec_done) * r->stride, m * r->stride, hipMemcpyHostToDevice, r->stream));
            s0 += m;
        }
        BDR_HIP(hipStreamSynchronize(r->stream));
        rec_done = upto;
        return BDR_OK;
    };
    auto alloc = [&](const uint8_t* frame, uint64_t oldest_live) -> int64_t {   // -> sequence number, or -1 (store exhausted)
        if (r->frame_seq >= r->frame_cap && r->frame_seq - r->frame_cap >= oldest_live) return -1;   // would overwrite a live frame
        memcpy(r->fstage + staged * fb, frame, fb);
        staged += 1;
        return (int64_t)r->frame_seq++;
    };
    for (uint64_t s = 0; s < n; ++s) {
        if (staged + 2 * (uint64_t)k > r->fstage_frames) BDR_TRY(flush_frames());
        if (s - rec_done >= r->stage_records) BDR_TRY(flush_records(s));
        const uint64_t pos = (r->i + s) % r->capacity;
        // the oldest frame any transition that stays live references: slot `pos` itself is being replaced
        uint64_t oldest_live = UINT64_MAX;
        const uint64_t live = std::min(r->size + s, r->capacity);
        if (live > 0) {
            const bool replacing = r->size + s >= r->capacity;
            if (!replacing) oldest_live = r->first_ref[(r->i + s + r->capacity - live) % r->capacity];
            else if (r->capacity > 1) oldest_live = r->first_ref[(pos + 1) % r->capacity];
        }
        if (r->have_last) oldest_live = std::min(oldest_live, *std::min_element(r->last_next_seq.begin(), r->last_next_seq.end()));
        const uint8_t* os = o + s * r->obs_bytes;
        const uint8_t* ns = x + s * r->obs_bytes;
        bool ok = true;
        if (r->have_last && memcmp(os, r->last_next.data(), r->obs_bytes) == 0) oseq = r->last_next_seq;
        else {
            for (int j = k - 1; j >= 0 && ok; --j) {   // oldest first, so that sequence numbers grow with time
                if (j < k - 1 && memcmp(os + j * fb, os + (j + 1) * fb, fb) == 0) oseq[j] = oseq[j + 1];
                else { const int64_t q = alloc(os + j * fb, oldest_live); ok = q >= 0; oseq[j] = (uint64_t)q; }
            }
        }
        if (ok) {
            if (k > 1 && memcmp(ns + fb, os, (k - 1) * fb) == 0) {
                for (int j = 1; j < k; ++j) nseq[j] = oseq[j - 1];
                if (memcmp(ns, ns + fb, fb) == 0) nseq[0] = nseq[1];
                else { const int64_t q = alloc(ns, std::min(oldest_live, *std::min_element(oseq.begin(), oseq.end()))); ok = q >= 0; nseq[0] = (uint64_t)q; }
            } else {
                for (int j = k - 1; j >= 0 && ok; --j) {
                    if (j < k - 1 && memcmp(ns + j * fb, ns + (j + 1) * fb, fb) == 0) nseq[j] = nseq[j + 1];
                    else { const int64_t q = alloc(ns + j * fb, std::min(oldest_live, *std::min_element(oseq.begin(), oseq.end()))); ok = q >= 0; nseq[j] = (uint64_t)q; }
                }
            }
        }
        if (!ok) {
            // keep the buffer consistent: what was staged for the transitions already accepted is written, the rest is dropped
            r->frame_seq = stage_first_seq + staged;   // (frames of the failed transition may have been staged: harmless)
            BDR_TRY(flush_frames());
            BDR_TRY(flush_records(s));
            BDR_HIP(hipEventRecord(r->written, r->stream)); mark_written(r);
            if (r->per && s) BDR_TRY(per_push(r->per, r->i, s, r->stream));
            r->i = (r->i + s) % r->capacity; r->size = std::min(r->size + s, r->capacity);
            return fail(BDR_ERR_INVALID, "single-frame store exhausted: %llu frames cannot hold the frames of %llu live transitions "
                                         "(raise bdr_replay_config::frame_capacity)", (unsigned long long)r->frame_cap, (unsigned long long)r->capacity);
        }
        uint8_t* rec = r->stage + (s - rec_done) * r->stride;
        memset(rec, 0, r->stride);
        uint32_t* slots = reinterpret_cast<uint32_t*>(rec);
        uint64_t fr = UINT64_MAX;
        for (int j = 0; j < k; ++j) {
            slots[j] = (uint32_t)(oseq[j] % r->frame_cap); slots[k + j] = (uint32_t)(nseq[j] % r->frame_cap);
            fr = std::min(fr, std::min(oseq[j], nseq[j]));
        }
        memcpy(rec + r->rec_act_off, a + s * r->act_bytes, r->act_bytes);
        memcpy(rec + r->rec_tail_off, &reward[s], 4);
        rec[r->rec_tail_off + 4] = (uint8_t)term[s]; rec[r->rec_tail_off + 5] = (uint8_t)trunc[s];
        r->first_ref[pos] = fr;
        memcpy(r->last_next.data(), ns, r->obs_bytes); r->last_next_seq = nseq; r->have_last = true;
    }
    BDR_TRY(flush_frames());
    BDR_TRY(flush_records(n));
    if (r->per) {
        BDR_HIP(hipStreamWaitEvent(r->stream, r->written, 0));
        BDR_TRY(per_push(r->per, r->i, n, r->stream));
    }
    BDR_HIP(hipEventRecord(r->written, r->stream)); mark_written(r);
    BDR_HIP(hipStreamSynchronize(r->stream));
    r->i = (r->i + n) % r->capacity;
    r->size = std::min(r->size + n, r->capacity);
    return BDR_OK;
}

extern "C" {

const char* bdr_last_error(void) { return bdr::g_err; }
int32_t bdr_last_error_is_deferred(void) { return bdr::g_err_deferred; }
const char* bdr_version(void) { return "border_amd 0.1 (gfx950)"; }

int32_t bdr_device_count(int32_t* count)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) n = 0;
    if (count) *count = n;
    return BDR_OK;
}

int32_t bdr_replay_create(const bdr_replay_config* cfg, bdr_replay** out)
{
    BDR_REQUIRE(cfg && out, "null argument");
    BDR_REQUIRE(cfg->capacity > 0, "capacity must be > 0");
    BDR_REQUIRE(cfg->obs_row_bytes > 0 && cfg->obs_row_bytes % 4 == 0, "obs_row_bytes must be a positive multiple of 4");
    BDR_REQUIRE(cfg->act_row_bytes > 0 && cfg->act_row_bytes % 4 == 0, "act_row_bytes must be a positive multiple of 4");
    BDR_TRY(ensure_device(cfg->device));
    bdr_replay* r = new bdr_replay();
    { static std::atomic<uint64_t> next_uid{1}; r->uid = next_uid.fetch_add(1); }
    r->device = cfg->device;
    r->capacity = cfg->capacity;
    r->obs_bytes = cfg->obs_row_bytes;
    r->act_bytes = cfg->act_row_bytes;
    r->next_off = round_up(r->obs_bytes, 16);
    r->act_off = round_up(r->next_off + r->obs_bytes, 16);
    r->tail_off = round_up(r->act_off + r->act_bytes, 8);
    const uint64_t raw = r->tail_off + 8;
    r->stride = round_up(raw, raw >= 1024 ? 128 : 16);
    seed_from_u64(cfg->seed, r->key);
    BDR_REQUIRE(cfg->index_rng == BDR_RNG_STDRNG || cfg->index_rng == BDR_RNG_XOSHIRO256PP, "index_rng must be BDR_RNG_STDRNG or BDR_RNG_XOSHIRO256PP");
    r->index_rng = cfg->index_rng;
    if (cfg->frame_stack > 0) {   // single-frame store: small records + frame store
        const int k = cfg->frame_stack;
        if (k > 64 || r->obs_bytes % (uint64_t)k != 0 || (r->obs_bytes / k) % 16 != 0) {
            delete r;
            return fail(BDR_ERR_INVALID, "frame_stack %d: obs_row_bytes must be frame_stack frames of a multiple of 16 bytes", k);
        }
        r->frame_stack = k; r->frame_bytes = r->obs_bytes / k;
        r->frame_cap = cfg->frame_capacity ? cfg->frame_capacity : r->capacity + r->capacity / 4 + 64;
        if (r->frame_cap < (uint64_t)2 * k + 1 || r->frame_cap > 0xFFFFFFFFull) { delete r; return fail(BDR_ERR_INVALID, "frame_capacity out of range"); }
        r->rec_act_off = round_up((uint64_t)8 * k, 8);
        r->rec_tail_off = round_up(r->rec_act_off + r->act_bytes, 8);
        r->stride = round_up(r->rec_tail_off + 8, 16);
        hipError_t fe = hipMalloc((void**)&r->frames, r->frame_cap * r->frame_bytes);
        if (fe != hipSuccess) {
            delete r;
            return fail(BDR_ERR_HIP, "hipMalloc of the %.2f GB frame store failed: %s", (double)(r->frame_cap * r->frame_bytes) / 1e9, hipGetErrorString(fe));
        }
        r->first_ref.assign(r->capacity, 0);
        r->last_next.resize(r->obs_bytes); r->last_next_seq.assign(k, 0);
    }
    hipError_t e = hipMalloc((void**)&r->ring, r->capacity * r->stride);
    if (e != hipSuccess) {
        delete r;
        return fail(BDR_ERR_HIP, "hipMalloc of the %.2f GB replay ring failed: %s",
                    (double)(cfg->capacity * r->stride) / 1e9, hipGetErrorString(e));
    }
    BDR_HIP(hipStreamCreateWithFlags(&r->stream, hipStreamNonBlocking));
    BDR_HIP(hipEventCreateWithFlags(&r->written, hipEventDisableTiming | hipEventDisableSystemFence));
    BDR_HIP(hipEventCreateWithFlags(&r->read, hipEventDisableTiming | hipEventDisableSystemFence));
    // rows are zero like `Tensor::zeros` / `vec![0.; capacity]` (tensor_batch.rs:95-101, base.rs:350-352)
    BDR_HIP(hipMemsetAsync(r->ring, 0, r->capacity * r->stride, r->stream));
    if (r->frames) {   // slot 0 stays all-zero until it is allocated: never-written records (all slots 0) read as zero rows
        BDR_HIP(hipMemsetAsync(r->frames, 0, r->frame_cap * r->frame_bytes, r->stream));
        r->fstage_frames = std::max<uint64_t>(2 * r->frame_stack, std::min<uint64_t>(1024, (8ull << 20) / r->frame_bytes));
        BDR_HIP(hipHostMalloc((void**)&r->fstage, r->fstage_frames * r->frame_bytes, hipHostMallocDefault));
    }
    if (r->index_rng == BDR_RNG_XOSHIRO256PP) {
        BDR_HIP(hipMalloc((void**)&r->xo_state, (size_t)XO_LANES * 4 * sizeof(uint64_t)));
        hipLaunchKernelGGL(k_xo_init, dim3(XO_LANES / 256), dim3(256), 0, r->stream, r->xo_state, cfg->seed, XO_LANES);
        BDR_HIP(hipGetLastError());
    }
    BDR_HIP(hipEventRecord(r->written, r->stream)); mark_written(r);
    r->stage_records = std::max<uint64_t>(1, std::min<uint64_t>(256, (8ull << 20) / r->stride));
    BDR_HIP(hipHostMalloc((void**)&r->stage, r->stage_records * r->stride, hipHostMallocDefault));
    replay_register(r);
    *out = r;
    return BDR_OK;
}

int32_t bdr_replay_destroy(bdr_replay* r)
{
    if (!r) return BDR_OK;
    replay_retire(r);   // agents that last sampled from this buffer find no handle for its uid from here on
    (void)hipSetDevice(r->device);
    (void)hipDeviceSynchronize();
    (void)hipFree(r->ring);
    (void)hipFree(r->frames);
    if (r->fstage) (void)hipHostFree(r->fstage);
    (void)hipHostFree(r->stage);
    if (r->done_host) (void)hipHostFree(r->done_host);
    (void)hipFree(r->done_ticket);
    (void)hipFree(r->b_obs); (void)hipFree(r->b_next); (void)hipFree(r->b_act);
    (void)hipFree(r->b_reward); (void)hipFree(r->b_term); (void)hipFree(r->b_trunc); (void)hipFree(r->b_ixs);
    (void)hipFree(r->alt.obs); (void)hipFree(r->alt.next); (void)hipFree(r->alt.act); (void)hipFree(r->alt.reward);
    (void)hipFree(r->alt.term); (void)hipFree(r->alt.trunc); (void)hipFree(r->alt.ixs);
    (void)hipFree(r->d_tails); (void)hipFree(r->xo_state);
    per_destroy(r->per);
    (void)hipEventDestroy(r->written); (void)hipEventDestroy(r->read);
    (void)hipStreamDestroy(r->stream);
    delete r;
    return BDR_OK;
}

int32_t bdr_replay_len(const bdr_replay* r, uint64_t* len)
{
    BDR_REQUIRE(r && len, "null argument");
    *len = r->size;
    return BDR_OK;
}

int32_t bdr_replay_frames_used(const bdr_replay* r, uint64_t* allocated, uint64_t* capacity)
{
    BDR_REQUIRE(r, "null replay handle");
    BDR_REQUIRE(r->frame_stack > 0, "not a single-frame store (bdr_replay_config::frame_stack == 0)");
    if (allocated) *allocated = r->frame_seq;
    if (capacity) *capacity = r->frame_cap;
    return BDR_OK;
}

int32_t bdr_replay_head(const bdr_replay* r, uint64_t* head)
{
    BDR_REQUIRE(r && head, "null argument");
    *head = r->i;
    return BDR_OK;
}

int32_t bdr_replay_push(bdr_replay* r, uint64_t n, const void* obs, const void* act, const void* next_obs,
                        const float* reward, const int8_t* term, const int8_t* trunc)
{
    BDR_REQUIRE(r, "null replay handle");
    if (n == 0) return BDR_OK;
    BDR_REQUIRE(obs && act && next_obs && reward && term && trunc, "null transition field");
    BDR_HIP(hipSetDevice(r->device));
    BDR_TRY(wait_for_reader(r, r->stream));  // WAR: do not overwrite rows a consumer's gather may still be reading
    const uint8_t* o = (const uint8_t*)obs; const uint8_t* a = (const uint8_t*)act; const uint8_t* x = (const uint8_t*)next_obs;
    if (r->frame_stack) return push_frames(r, n, o, a, x, reward, term, trunc);
    uint64_t done = 0;
    while (done < n) {
        // records that fit the staging buffer and do not wrap the ring
        const uint64_t pos = (r->i + done) % r->capacity;
        const uint64_t m = std::min(std::min(n - done, r->stage_records), r->capacity - pos);
        BDR_HIP(hipStreamSynchronize(r->stream));  // staging buffer free again
        for (uint64_t k = 0; k < m; ++k) {
            uint8_t* rec = r->stage + k * r->stride;
            const uint64_t s = done + k;
            memcpy(rec, o + s * r->obs_bytes, r->obs_bytes);
            memcpy(rec + r->next_off, x + s * r->obs_bytes, r->obs_bytes);
            memcpy(rec + r->act_off, a + s * r->act_bytes, r->act_bytes);
            memcpy(rec + r->tail_off, &reward[s], 4);
            rec[r->tail_off + 4] = (uint8_t)term[s];
            rec[r->tail_off + 5] = (uint8_t)trunc[s];
        }
        BDR_HIP(hipMemcpyAsync(r->ring + pos * r->stride, r->stage, m * r->stride, hipMemcpyHostToDevice, r->stream));
        done += m;
    }
    if (r->per) {   // base.rs:304-306 set_priority(len); ordered behind an agent's update_priority on its own stream
        if (r->written_lazy) { r->written_lazy = false; BDR_HIP(hipEventRecord(r->written, r->stream)); }
        BDR_HIP(hipStreamWaitEvent(r->stream, r->written, 0));
        BDR_TRY(per_push(r->per, r->i, n, r->stream));
        BDR_HIP(hipEventRecord(r->written, r->stream)); mark_written(r);
    } else {
        r->written_lazy = true; mark_written(r);   // `written` is recorded when a consumer on another stream asks for it (wait_for_writer)
    }
    // The caller's rows are in the pinned staging buffer: its buffers are free.  The copy to the ring is in flight on the buffer's stream;
    // whoever touches the staging buffer next waits for it first (the synchronisation at the head of every staging loop), consumers of the
    // ring wait through `written`.  (A synchronisation here was ~10 us of every push for nothing the caller needs.)
    // base.rs:308-312
    r->i = (r->i + n) % r->capacity;
    r->size += n;
    if (r->size >= r->capacity) r->size = r->capacity;
    return BDR_OK;
}

// ExperienceBufferBase::push for transitions whose observation rows already live in HBM (the frame stacks of a bdr_atari_prep:
// obs = the stacks before the step, next_obs = after it).  One workgroup per record copies the two rows inside HBM (16-byte lanes
// where the addresses allow) and writes act / reward / flags from a small staged array - the rows never cross PCIe.
struct PushDevArgs {
    uint8_t* ring; uint64_t stride, obs_bytes, act_bytes, next_off, act_off, tail_off, pos;
    const uint8_t* obs; uint64_t obs_stride; const uint8_t* next; uint64_t next_stride;
    const uint8_t* tails;   // [m][act_bytes + 8]: act | reward f32 | is_terminated | is_truncated | 2 pad
};
__global__ __launch_bounds__(256) void k_push_device(PushDevArgs a)
{
    const uint64_t k = blockIdx.x;
    uint8_t* rec = a.ring + (a.pos + k) * a.stride;
    const uint8_t* src[2] = {a.obs + k * a.obs_stride, a.next + k * a.next_stride};
    uint8_t* dst[2] = {rec, rec + a.next_off};
    for (int w = 0; w < 2; ++w) {
        if ((((uintptr_t)src[w] | (uintptr_t)dst[w] | a.obs_bytes) & 15) == 0) {
            const uint4* s4 = reinterpret_cast<const uint4*>(src[w]); uint4* d4 = reinterpret_cast<uint4*>(dst[w]);
            for (uint64_t i = threadIdx.x; i < a.obs_bytes / 16; i += 256) d4[i] = s4[i];
        } else if ((((uintptr_t)src[w] | (uintptr_t)dst[w] | a.obs_bytes) & 3) == 0) {
            const uint32_t* s1 = reinterpret_cast<const uint32_t*>(src[w]); uint32_t* d1 = reinterpret_cast<uint32_t*>(dst[w]);
            for (uint64_t i = threadIdx.x; i < a.obs_bytes / 4; i += 256) d1[i] = s1[i];
        } else {
            for (uint64_t i = threadIdx.x; i < a.obs_bytes; i += 256) dst[w][i] = src[w][i];
        }
    }
    const uint8_t* t = a.tails + k * (a.act_bytes + 8);
    for (uint64_t i = threadIdx.x; i < a.act_bytes; i += 256) rec[a.act_off + i] = t[i];
    if (threadIdx.x < 6) rec[a.tail_off + threadIdx.x] = t[a.act_bytes + threadIdx.x];
}

// The same push for a handful of transitions (the one-environment online loop pushes ONE per step): the small fields ride in the
// kernel arguments - no staging buffer to wait for, no host -> device copy in front of the kernel.
constexpr int PUSH_SMALL_N = 8, PUSH_SMALL_TAIL = 24;   // up to 8 records whose act | reward | flags fit 24 bytes (a discrete action is 8)
struct PushDevSmallArgs { PushDevArgs a; uint64_t capacity; unsigned* ticket; unsigned* done; unsigned seq; uint8_t tails[PUSH_SMALL_N * PUSH_SMALL_TAIL]; };   // (a.pos + k wraps at capacity; a.tails unused)
__global__ __launch_bounds__(256) void k_push_device_small(PushDevSmallArgs s)
{
    const PushDevArgs& a = s.a;
    const uint64_t k = blockIdx.x;
    uint8_t* rec = a.ring + ((a.pos + k) % s.capacity) * a.stride;
    const uint8_t* src[2] = {a.obs + k * a.obs_stride, a.next + k * a.next_stride};
    uint8_t* dst[2] = {rec, rec + a.next_off};
    for (int w = 0; w < 2; ++w) {
        if ((((uintptr_t)src[w] | (uintptr_t)dst[w] | a.obs_bytes) & 15) == 0) {
            const uint4* s4 = reinterpret_cast<const uint4*>(src[w]); uint4* d4 = reinterpret_cast<uint4*>(dst[w]);
            for (uint64_t i = threadIdx.x; i < a.obs_bytes / 16; i += 256) d4[i] = s4[i];
        } else {
            for (uint64_t i = threadIdx.x; i < a.obs_bytes; i += 256) dst[w][i] = src[w][i];
        }
    }
    const uint8_t* t = s.tails + k * PUSH_SMALL_TAIL;
    for (uint64_t i = threadIdx.x; i < a.act_bytes; i += 256) rec[a.act_off + i] = t[i];
    if (threadIdx.x < 6) rec[a.tail_off + threadIdx.x] = t[a.act_bytes + threadIdx.x];
    // "the source rows have been read": the last record to finish says so in pinned host memory, where the caller waits for it - a
    // hipStreamSynchronize costs ~15 us of host latency for a 3 us kernel (every load of this workgroup has returned: its data went into the
    // stores above; what the stores themselves still owe is ordered by the stream for everything that follows on the device)
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned n_done = __hip_atomic_fetch_add(s.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
        if (n_done == gridDim.x) {
            __hip_atomic_store(s.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(s.done, s.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

int32_t bdr_replay_push_device(bdr_replay* r, uint64_t n, const void* obs_dev, uint64_t obs_stride, const void* act, const void* next_obs_dev,
                               uint64_t next_obs_stride, const float* reward, const int8_t* term, const int8_t* trunc)
{
    BDR_REQUIRE(r, "null replay handle");
    if (n == 0) return BDR_OK;
    BDR_REQUIRE(obs_dev && act && next_obs_dev && reward && term && trunc, "null transition field");
    BDR_REQUIRE(obs_stride >= r->obs_bytes && next_obs_stride >= r->obs_bytes, "row strides must be >= the observation row size");
    BDR_REQUIRE(!r->frame_stack, "the single-frame store finds shared frames by comparing host rows: push host rows (bdr_replay_push) into a "
                                 "buffer with frame_stack > 0");
    BDR_HIP(hipSetDevice(r->device));
    for (const void* p : {obs_dev, next_obs_dev}) {
        // (the check is a driver call of ~2 us: an address that passed it last time - the environment's stacks, push after push - is not asked
        //  again; every 1024th push asks anyway, so an allocation that was freed and whose address came back as host or another GPU's memory
        //  is noticed within a bounded number of pushes instead of never)
        if ((p == r->dev_rows_ok[0] || p == r->dev_rows_ok[1]) && (r->dev_rows_checks++ & 1023u) != 1023u) continue;
        hipPointerAttribute_t at{};
        BDR_REQUIRE(hipPointerGetAttributes(&at, p) == hipSuccess && at.type == hipMemoryTypeDevice && at.device == r->device,
                    "observation rows must be device memory of the buffer's GPU (host rows go through bdr_replay_push)");
        r->dev_rows_ok[p == obs_dev ? 0 : 1] = p;
    }
    BDR_TRY(wait_for_reader(r, r->stream));  // WAR: do not overwrite rows a consumer's gather may still be reading
    if (n <= (uint64_t)PUSH_SMALL_N && r->act_bytes + 8 <= (uint64_t)PUSH_SMALL_TAIL && !r->per) {
        PushDevSmallArgs sa{};
        sa.a = PushDevArgs{r->ring, r->stride, r->obs_bytes, r->act_bytes, r->next_off, r->act_off, r->tail_off, r->i,
                           (const uint8_t*)obs_dev, obs_stride, (const uint8_t*)next_obs_dev, next_obs_stride, nullptr};
        sa.capacity = r->capacity;
        const uint8_t* ab = (const uint8_t*)act;
        for (uint64_t k = 0; k < n; ++k) {
            uint8_t* t = sa.tails + k * PUSH_SMALL_TAIL;
            memcpy(t, ab + k * r->act_bytes, r->act_bytes);
            memcpy(t + r->act_bytes, &reward[k], 4);
            t[r->act_bytes + 4] = (uint8_t)term[k]; t[r->act_bytes + 5] = (uint8_t)trunc[k];
        }
        if (!r->done_host) {
            BDR_HIP(hipHostMalloc((void**)&r->done_host, 64, hipHostMallocMapped));
            *reinterpret_cast<volatile unsigned*>(r->done_host) = 0;
            BDR_HIP(hipHostGetDevicePointer((void**)&r->done_dev, r->done_host, 0));
            BDR_HIP(hipMalloc((void**)&r->done_ticket, sizeof(unsigned)));
            BDR_HIP(hipMemsetAsync(r->done_ticket, 0, sizeof(unsigned), r->stream));
        }
        sa.ticket = r->done_ticket; sa.done = r->done_dev; sa.seq = ++r->done_seq;
        hipLaunchKernelGGL(k_push_device_small, dim3((uint32_t)n), dim3(256), 0, r->stream, sa);
        BDR_HIP(hipGetLastError());
        r->written_lazy = true; mark_written(r);   // `written` is recorded when a consumer on another stream asks for it (wait_for_writer)
        // the caller's device rows may be overwritten by its next environment step: wait until the kernel has read them
        {
            const volatile unsigned* done = reinterpret_cast<const volatile unsigned*>(r->done_host);
            const auto t0 = std::chrono::steady_clock::now();
            for (unsigned spins = 0; (int)(*done - sa.seq) < 0; ++spins) {
                if ((spins & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(10)) {   // (a lost kernel: fall back to the stream's own report)
                    BDR_HIP(hipStreamSynchronize(r->stream));
                    if ((int)(*done - sa.seq) < 0) return fail(BDR_ERR_HIP, "a small device push never reported completion (sequence %u)", sa.seq);
                    break;
                }
            }
        }
        r->i = (r->i + n) % r->capacity;
        r->size = std::min(r->size + n, r->capacity);
        return BDR_OK;
    }
    const uint64_t tw = r->act_bytes + 8;
    const uint8_t* a = (const uint8_t*)act;
    uint64_t done = 0;
    while (done < n) {
        const uint64_t pos = (r->i + done) % r->capacity;
        // the pinned staging buffer (stage_records * stride bytes) holds the small fields of a run
        const uint64_t fit = std::max<uint64_t>(1, r->stage_records * r->stride / tw);
        const uint64_t m = std::min(std::min(n - done, fit), r->capacity - pos);
        BDR_HIP(hipStreamSynchronize(r->stream));  // staging buffer free again
        if (m > r->tails_cap) {
            (void)hipFree(r->d_tails); r->d_tails = nullptr; r->tails_cap = 0;
            BDR_HIP(hipMalloc((void**)&r->d_tails, std::max<uint64_t>(m, 256) * tw));
            r->tails_cap = std::max<uint64_t>(m, 256);
        }
        for (uint64_t k = 0; k < m; ++k) {
            uint8_t* t = r->stage + k * tw;
            const uint64_t s = done + k;
            memcpy(t, a + s * r->act_bytes, r->act_bytes);
            memcpy(t + r->act_bytes, &reward[s], 4);
            t[r->act_bytes + 4] = (uint8_t)term[s]; t[r->act_bytes + 5] = (uint8_t)trunc[s]; t[r->act_bytes + 6] = t[r->act_bytes + 7] = 0;
        }
        BDR_HIP(hipMemcpyAsync(r->d_tails, r->stage, m * tw, hipMemcpyHostToDevice, r->stream));
        PushDevArgs pa{r->ring, r->stride, r->obs_bytes, r->act_bytes, r->next_off, r->act_off, r->tail_off, pos,
                       (const uint8_t*)obs_dev + done * obs_stride, obs_stride, (const uint8_t*)next_obs_dev + done * next_obs_stride, next_obs_stride, r->d_tails};
        hipLaunchKernelGGL(k_push_device, dim3((uint32_t)m), dim3(256), 0, r->stream, pa);
        BDR_HIP(hipGetLastError());
        done += m;
    }
    if (r->per) {   // base.rs:304-306 set_priority(len), as bdr_replay_push
        BDR_HIP(hipStreamWaitEvent(r->stream, r->written, 0));
        BDR_TRY(per_push(r->per, r->i, n, r->stream));
    }
    BDR_HIP(hipEventRecord(r->written, r->stream)); mark_written(r);
    BDR_HIP(hipStreamSynchronize(r->stream));  // the caller's device rows may be overwritten by its next environment step
    r->i = (r->i + n) % r->capacity;
    r->size += n;
    if (r->size >= r->capacity) r->size = r->capacity;
    return BDR_OK;
}

int32_t bdr_replay_fill_synthetic(bdr_replay* r, uint64_t n, uint64_t seed, int32_t kind, int32_t n_actions)
{
    BDR_REQUIRE(r, "null replay handle");
    BDR_REQUIRE(kind == 0 || kind == 1, "kind must be 0 or 1");
    BDR_REQUIRE(n_actions >= 0, "n_actions must be >= 0");
    BDR_REQUIRE(n_actions == 0 || r->act_bytes == 8, "discrete actions are one i64 per row");
    BDR_REQUIRE(kind == 0 || r->obs_bytes % 4 == 0, "f32 rows need obs_row_bytes %% 4 == 0");
    BDR_REQUIRE(n <= r->capacity, "fill count exceeds capacity");
    BDR_REQUIRE(!r->per || r->size == 0, "synthetic fill with PER needs an empty buffer");   // before anything is overwritten
    BDR_HIP(hipSetDevice(r->device));
    BDR_TRY(wait_for_reader(r, r->stream));
    if (r->frame_stack) {   // one continuous episode of n transitions over n + k frames (see k_fill_frames)
        BDR_REQUIRE(kind == 0 && n_actions > 0, "the single-frame store is filled with Atari-like rows (kind 0, discrete actions)");
        BDR_REQUIRE(r->size == 0 && r->frame_seq == 0, "synthetic fill of the single-frame store needs an empty buffer");
        BDR_REQUIRE(n + (uint64_t)r->frame_stack <= r->frame_cap, "fill needs n + frame_stack frames");
        const int k = r->frame_stack;
        FillFramesArgs fa{r->ring, r->stride, r->rec_act_off, r->rec_tail_off, r->frames, r->frame_bytes, r->frame_cap, n, seed, r->capacity, k, n_actions};
        hipLaunchKernelGGL(k_fill_frames, dim3((uint32_t)(n + k)), dim3(256), 0, r->stream, fa);
        BDR_HIP(hipGetLastError());
        if (r->per) BDR_TRY(per_push(r->per, 0, n, r->stream));
        BDR_HIP(hipEventRecord(r->written, r->stream)); mark_written(r);
        for (uint64_t t = 0; t < n; ++t) r->first_ref[t % r->capacity] = t;
        r->frame_seq = n + k;
        // the last next_obs (frames n+k-1 .. n, newest first) as push() will compare against it
        for (int j = 0; j < k; ++j) {
            r->last_next_seq[j] = n + k - 1 - j;
            uint64_t* dst = reinterpret_cast<uint64_t*>(r->last_next.data() + (uint64_t)j * r->frame_bytes);
            for (uint64_t w = 0; w < r->frame_bytes / 8; ++w) dst[w] = synth_hash(seed, n + k - 1 - j, 7, w);
        }
        r->have_last = n > 0;
        r->i = n % r->capacity; r->size = n;
        return BDR_OK;
    }
    FillArgs a{r->ring, r->stride, r->obs_bytes, r->act_bytes, r->next_off, r->act_off, r->tail_off, r->capacity,
               n, seed, kind, n_actions};
    // grid.x is limited to 2^31-1; n <= capacity fits comfortably for the sizes used here
    hipLaunchKernelGGL(k_fill_synthetic, dim3((uint32_t)n), dim3(256), 0, r->stream, a);
    BDR_HIP(hipGetLastError());
    if (r->per) {   // the fill is one push of n rows into an empty ring: one set_priority(n)
        BDR_TRY(per_push(r->per, 0, n, r->stream));
    }
    BDR_HIP(hipEventRecord(r->written, r->stream)); mark_written(r);
    r->i = n % r->capacity;
    r->size = n;
    return BDR_OK;
}

int32_t bdr_replay_read_rows(bdr_replay* r, uint64_t first, uint64_t n, void* obs, void* act, void* next_obs,
                             float* reward, int8_t* term, int8_t* trunc)
{
    BDR_REQUIRE(r, "null replay handle");
    BDR_REQUIRE(first + n <= r->capacity, "row range out of bounds");
    BDR_HIP(hipSetDevice(r->device));
    BDR_HIP(hipStreamSynchronize(r->stream));
    std::vector<uint8_t> tmp(r->stride);
    if (r->frame_stack) {
        const int K = r->frame_stack;
        for (uint64_t k = 0; k < n; ++k) {
            BDR_HIP(hipMemcpy(tmp.data(), r->ring + (first + k) * r->stride, r->stride, hipMemcpyDeviceToHost));
            const uint32_t* slots = reinterpret_cast<const uint32_t*>(tmp.data());
            for (int j = 0; j < K; ++j) {
                if (obs) BDR_HIP(hipMemcpy((uint8_t*)obs + k * r->obs_bytes + (uint64_t)j * r->frame_bytes, r->frames + (uint64_t)slots[j] * r->frame_bytes, r->frame_bytes, hipMemcpyDeviceToHost));
                if (next_obs) BDR_HIP(hipMemcpy((uint8_t*)next_obs + k * r->obs_bytes + (uint64_t)j * r->frame_bytes, r->frames + (uint64_t)slots[K + j] * r->frame_bytes, r->frame_bytes, hipMemcpyDeviceToHost));
            }
            if (act) memcpy((uint8_t*)act + k * r->act_bytes, tmp.data() + r->rec_act_off, r->act_bytes);
            if (reward) memcpy(&reward[k], tmp.data() + r->rec_tail_off, 4);
            if (term) term[k] = (int8_t)tmp[r->rec_tail_off + 4];
            if (trunc) trunc[k] = (int8_t)tmp[r->rec_tail_off + 5];
        }
        return BDR_OK;
    }
    for (uint64_t k = 0; k < n; ++k) {
        BDR_HIP(hipMemcpy(tmp.data(), r->ring + (first + k) * r->stride, r->stride, hipMemcpyDeviceToHost));
        if (obs) memcpy((uint8_t*)obs + k * r->obs_bytes, tmp.data(), r->obs_bytes);
        if (next_obs) memcpy((uint8_t*)next_obs + k * r->obs_bytes, tmp.data() + r->next_off, r->obs_bytes);
        if (act) memcpy((uint8_t*)act + k * r->act_bytes, tmp.data() + r->act_off, r->act_bytes);
        if (reward) memcpy(&reward[k], tmp.data() + r->tail_off, 4);
        if (term) term[k] = (int8_t)tmp[r->tail_off + 4];
        if (trunc) trunc[k] = (int8_t)tmp[r->tail_off + 5];
    }
    return BDR_OK;
}

}  // extern "C"

namespace bdr {

static void free_alt(bdr_replay* r)
{
    (void)hipFree(r->alt.obs); (void)hipFree(r->alt.next); (void)hipFree(r->alt.act); (void)hipFree(r->alt.reward);
    (void)hipFree(r->alt.term); (void)hipFree(r->alt.trunc); (void)hipFree(r->alt.ixs);
    r->alt = bdr_replay::BatchSet{};
    r->alt_valid = false;
}

int32_t replay_flip_batch(bdr_replay* r, uint64_t n)
{
    BDR_TRY(replay_ensure_batch_capacity(r, n));
    if (!r->alt_valid) {
        const uint64_t c = r->batch_cap;
        BDR_HIP(hipMalloc((void**)&r->alt.obs, c * r->obs_bytes));
        BDR_HIP(hipMalloc((void**)&r->alt.next, c * r->obs_bytes));
        BDR_HIP(hipMalloc((void**)&r->alt.act, c * r->act_bytes));
        BDR_HIP(hipMalloc((void**)&r->alt.reward, c * 4));
        BDR_HIP(hipMalloc((void**)&r->alt.term, round_up(c, 16)));
        BDR_HIP(hipMalloc((void**)&r->alt.trunc, round_up(c, 16)));
        BDR_HIP(hipMalloc((void**)&r->alt.ixs, c * 8));
        r->alt_valid = true;
    }
    std::swap(r->b_obs, r->alt.obs); std::swap(r->b_next, r->alt.next); std::swap(r->b_act, r->alt.act);
    std::swap(r->b_reward, r->alt.reward); std::swap(r->b_term, r->alt.term); std::swap(r->b_trunc, r->alt.trunc);
    std::swap(r->b_ixs, r->alt.ixs);
    r->batch_gen += 1;
    return BDR_OK;
}

int32_t replay_ensure_batch_capacity(bdr_replay* r, uint64_t n)
{
    if (n <= r->batch_cap) return BDR_OK;
    BDR_HIP(hipDeviceSynchronize());
    free_alt(r);
    (void)hipFree(r->b_obs); (void)hipFree(r->b_next); (void)hipFree(r->b_act);
    (void)hipFree(r->b_reward); (void)hipFree(r->b_term); (void)hipFree(r->b_trunc); (void)hipFree(r->b_ixs);
    BDR_HIP(hipMalloc((void**)&r->b_obs, n * r->obs_bytes));
    BDR_HIP(hipMalloc((void**)&r->b_next, n * r->obs_bytes));
    BDR_HIP(hipMalloc((void**)&r->b_act, n * r->act_bytes));
    BDR_HIP(hipMalloc((void**)&r->b_reward, n * 4));
    BDR_HIP(hipMalloc((void**)&r->b_term, round_up(n, 16)));
    BDR_HIP(hipMalloc((void**)&r->b_trunc, round_up(n, 16)));
    BDR_HIP(hipMalloc((void**)&r->b_ixs, n * 8));
    r->batch_cap = n;
    r->batch_gen += 1;
    return BDR_OK;
}

static int32_t launch_indices(bdr_replay* r, uint64_t n, hipStream_t stream)
{
    if (r->index_rng == BDR_RNG_XOSHIRO256PP) {
        BDR_REQUIRE(n <= XO_LANES, "xoshiro index generator: at most %u samples per batch (one generator per batch lane)", XO_LANES);
        BDR_HIP(step_launch(stream, false, k_xo_indices, dim3((uint32_t)((n + 255) / 256)), dim3(256), r->xo_state, r->size, (uint32_t)n, r->b_ixs));
        return BDR_OK;
    }
    ChaChaKey key;
    memcpy(key.k, r->key, sizeof key.k);
    hipLaunchKernelGGL(k_sample_indices, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, stream, key, r->word_pos,
                       r->size, (uint32_t)n, r->b_ixs);
    BDR_HIP(hipGetLastError());
    r->word_pos += n;  // one next_u32() per index
    return BDR_OK;
}

// Everything a sample on `stream` needs BEFORE its kernels can be enqueued: the batch buffers, the order behind the writer and
// the handover from the previous consumer.  Idempotent; agents that capture their step into a graph call it ahead of the
// capture (allocation, event waits on work outside the graph and device syncs are not capturable).
int32_t replay_prepare_sample(bdr_replay* r, uint64_t n, hipStream_t stream)
{
    if (r->size == 0) return fail(BDR_ERR_EMPTY, "batch() on an empty replay buffer");
    BDR_REQUIRE(n > 0 && n < (1ull << 24), "batch size out of range");
    BDR_TRY(replay_ensure_batch_capacity(r, n));
    if (stream != r->stream) stream_register(stream);
    BDR_TRY(wait_for_writer(r, stream));
    // The batch buffers belong to the consumer that sampled last: kernels on any of ITS queues may still be reading them
    // (the reference's `&mut` exclusivity, one consumer at a time).  A different queue taking over is rare (two agents on
    // one buffer, bdr_replay_batch between opts, profiling toggles) and simply drains the device.
    if (r->read_pending && r->read_stream != stream) {
        BDR_HIP(hipDeviceSynchronize());
        r->read_pending = false;
    }
    return BDR_OK;
}

int32_t replay_sample_on_stream(bdr_replay* r, uint64_t n, hipStream_t stream)
{
    BDR_TRY(replay_prepare_sample(r, n, stream));
    GatherArgs a{};
    a.ring = r->ring; a.stride = r->stride; a.obs_bytes = r->obs_bytes; a.act_bytes = r->act_bytes;
    a.next_off = r->next_off; a.act_off = r->act_off; a.tail_off = r->tail_off; a.ixs = r->b_ixs;
    memcpy(a.key.k, r->key, sizeof a.key.k); a.word_pos = r->word_pos; a.size = r->size;
    a.b_obs = r->b_obs; a.b_next = r->b_next; a.b_act = r->b_act; a.b_reward = r->b_reward; a.b_term = r->b_term; a.b_trunc = r->b_trunc;
    a.chunks = 1; a.vec_per_chunk = 0; a.given = 0;
    if (r->per) {   // base.rs:377-383: sum-tree sampling + importance weights; one stream word per sample
        BDR_TRY(per_sample(r->per, r->key, r->word_pos, n, r->b_ixs, stream));
        a.given = 1;
    } else if (r->index_rng == BDR_RNG_XOSHIRO256PP) {   // the device-native generator: its own kernel steps the lanes, the gather takes the indices
        BDR_TRY(launch_indices(r, n, stream));
        a.given = 1;
    }
    r->word_pos += n;  // one next_u32() per index
    if (r->frame_stack) {
        GatherFramesArgs g{};
        g.recs = r->ring; g.rec_stride = r->stride; g.rec_act_off = r->rec_act_off; g.rec_tail_off = r->rec_tail_off; g.act_bytes = r->act_bytes;
        g.frames = r->frames; g.frame_bytes = r->frame_bytes; g.k = r->frame_stack;
        g.ixs = a.ixs; g.key = a.key; g.word_pos = a.word_pos; g.size = a.size;
        g.b_obs = a.b_obs; g.b_next = a.b_next; g.b_act = a.b_act; g.b_reward = a.b_reward; g.b_term = a.b_term; g.b_trunc = a.b_trunc;
        g.given = a.given;
        BDR_HIP(step_launch(stream, true, k_gather_frames, dim3((uint32_t)(n * 2)), dim3(256), g));
    } else if (r->obs_bytes % 16 == 0) {
        const uint64_t nvec = r->obs_bytes / 16;
        // up to 4 vectors per thread per section and pass: 2 workgroups per sample for Atari rows (1764 vectors)
        a.chunks = (uint32_t)std::max<uint64_t>(1, (nvec + 1023) / 1024);
        { static const int forced = getenv("BDR_GATHER_CHUNKS") ? atoi(getenv("BDR_GATHER_CHUNKS")) : 0; if (forced > 0) a.chunks = (uint32_t)forced; }   // diagnostics
        a.vec_per_chunk = (uint32_t)((nvec + a.chunks - 1) / a.chunks);
        BDR_HIP(step_launch(stream, true, k_gather<u32x4>, dim3((uint32_t)(n * a.chunks)), dim3(256), a));
    } else {
        const uint64_t nvec = r->obs_bytes / 4;
        a.chunks = (uint32_t)std::max<uint64_t>(1, (nvec + 1023) / 1024);
        a.vec_per_chunk = (uint32_t)((nvec + a.chunks - 1) / a.chunks);
        BDR_HIP(step_launch(stream, true, k_gather<uint32_t>, dim3((uint32_t)(n * a.chunks)), dim3(256), a));
    }
    r->read_pending = true; r->read_stream = stream;   // recorded only if somebody has to wait for it (wait_for_reader)
    r->batch_n = n;
    return BDR_OK;
}

// The gather of a uniform sample over the plain ring as a descriptor instead of a launch: the caller's own kernel draws the
// indices and copies the rows (the one-workgroup Mlp step, mlp_fused.hpp, where a separate gather launch is a fifth of the step).
// All host bookkeeping of a sample (stream position, consumer hand-over) is done here; the caller MUST run a kernel on `stream`
// that fills the batch buffers exactly as k_gather would.
int32_t replay_sample_plan(bdr_replay* r, uint64_t n, hipStream_t stream, GatherArgs* out)
{
    BDR_REQUIRE(!r->per && !r->frame_stack && r->index_rng == BDR_RNG_STDRNG, "sample plans cover the plain uniform ring with the StdRng index stream only");
    BDR_TRY(replay_prepare_sample(r, n, stream));
    GatherArgs a{};
    a.ring = r->ring; a.stride = r->stride; a.obs_bytes = r->obs_bytes; a.act_bytes = r->act_bytes;
    a.next_off = r->next_off; a.act_off = r->act_off; a.tail_off = r->tail_off; a.ixs = r->b_ixs;
    memcpy(a.key.k, r->key, sizeof a.key.k); a.word_pos = r->word_pos; a.size = r->size;
    a.b_obs = r->b_obs; a.b_next = r->b_next; a.b_act = r->b_act; a.b_reward = r->b_reward; a.b_term = r->b_term; a.b_trunc = r->b_trunc;
    a.chunks = 1; a.vec_per_chunk = 0; a.given = 0;
    r->word_pos += n;  // one next_u32() per index
    r->read_pending = true; r->read_stream = stream;
    r->batch_n = n;
    *out = a;
    return BDR_OK;
}

// dqn/base.rs:143: buffer.update_priority(&ixs, &Some(td_errs)) for the batch last drawn on `stream`
int32_t replay_update_priority_on_stream(bdr_replay* r, uint64_t n, const float* td_dev, hipStream_t stream)
{
    if (!r->per) return BDR_OK;
    BDR_REQUIRE(n == r->batch_n, "update_priority size differs from the last batch");
    BDR_TRY(per_update(r->per, n, r->b_ixs, td_dev, stream));
    BDR_HIP(hipEventRecord(r->written, stream)); mark_written(r);   // later pushes / samples order behind the tree update
    return BDR_OK;
}

}  // namespace bdr

extern "C" {

int32_t bdr_replay_sample_indices(bdr_replay* r, uint64_t n, uint64_t* ixs_out)
{
    BDR_REQUIRE(r && ixs_out, "null argument");
    if (r->size == 0) return fail(BDR_ERR_EMPTY, "batch() on an empty replay buffer");
    BDR_REQUIRE(n > 0 && n < (1ull << 24), "batch size out of range");
    BDR_HIP(hipSetDevice(r->device));
    BDR_TRY(replay_ensure_batch_capacity(r, n));
    BDR_TRY(launch_indices(r, n, r->stream));
    BDR_HIP(hipMemcpyAsync(ixs_out, r->b_ixs, n * 8, hipMemcpyDeviceToHost, r->stream));
    BDR_HIP(hipStreamSynchronize(r->stream));
    return BDR_OK;
}

int32_t bdr_replay_batch(bdr_replay* r, uint64_t n, uint64_t* ixs_out, void* obs_out, void* act_out,
                         void* next_obs_out, float* reward_out, int8_t* term_out, int8_t* trunc_out)
{
    BDR_REQUIRE(r, "null replay handle");
    BDR_HIP(hipSetDevice(r->device));
    BDR_TRY(replay_sample_on_stream(r, n, r->stream));
    hipStream_t s = r->stream;
    if (ixs_out) BDR_HIP(hipMemcpyAsync(ixs_out, r->b_ixs, n * 8, hipMemcpyDeviceToHost, s));
    if (obs_out) BDR_HIP(hipMemcpyAsync(obs_out, r->b_obs, n * r->obs_bytes, hipMemcpyDeviceToHost, s));
    if (next_obs_out) BDR_HIP(hipMemcpyAsync(next_obs_out, r->b_next, n * r->obs_bytes, hipMemcpyDeviceToHost, s));
    if (act_out) BDR_HIP(hipMemcpyAsync(act_out, r->b_act, n * r->act_bytes, hipMemcpyDeviceToHost, s));
    if (reward_out) BDR_HIP(hipMemcpyAsync(reward_out, r->b_reward, n * 4, hipMemcpyDeviceToHost, s));
    if (term_out) BDR_HIP(hipMemcpyAsync(term_out, r->b_term, n, hipMemcpyDeviceToHost, s));
    if (trunc_out) BDR_HIP(hipMemcpyAsync(trunc_out, r->b_trunc, n, hipMemcpyDeviceToHost, s));
    BDR_HIP(hipStreamSynchronize(s));
    return BDR_OK;
}

int32_t bdr_replay_last_batch(const bdr_replay* r, bdr_device_batch* out)
{
    BDR_REQUIRE(r && out, "null argument");
    BDR_REQUIRE(r->batch_n > 0, "no batch has been drawn yet");
    out->n = r->batch_n; out->obs = r->b_obs; out->next_obs = r->b_next; out->act = r->b_act;
    out->reward = r->b_reward; out->is_terminated = r->b_term; out->is_truncated = r->b_trunc; out->ixs = r->b_ixs;
    out->weight = replay_batch_weights(r);
    return BDR_OK;
}

void bdr_per_config_default(bdr_per_config* c)   // config.rs:67-83
{
    if (!c) return;
    c->alpha = 0.6f; c->beta_0 = 0.4f; c->beta_final = 1.0f; c->n_opts_final = 500000;
    c->normalize = BDR_PER_NORMALIZE_ALL; c->reserved = 0;
}

int32_t bdr_replay_enable_per(bdr_replay* r, const bdr_per_config* c)
{
    BDR_REQUIRE(r && c, "null argument");
    BDR_REQUIRE(!r->per, "PER is already enabled");
    BDR_REQUIRE(r->size == 0 && r->i == 0, "PER must be enabled on an empty buffer");
    BDR_REQUIRE(r->index_rng == BDR_RNG_STDRNG, "prioritized sampling draws from the StdRng stream (base.rs:377-383): index_rng must be BDR_RNG_STDRNG");
    BDR_HIP(hipSetDevice(r->device));
    BDR_TRY(per_create(c, r->capacity, r->stream, &r->per));
    BDR_HIP(hipEventRecord(r->written, r->stream)); mark_written(r);
    return BDR_OK;
}

int32_t bdr_replay_update_priority(bdr_replay* r, uint64_t n, const uint64_t* ixs, const float* td_errs)
{
    BDR_REQUIRE(r, "null replay handle");
    if (!r->per) return BDR_OK;   // base.rs:414: no-op without per_state
    BDR_REQUIRE(ixs && td_errs, "ixs and td_errs must be given when PER is enabled");   // the reference's expect()s
    BDR_REQUIRE(n > 0 && n <= (1ull << 20), "update size out of range");
    BDR_HIP(hipSetDevice(r->device));
    for (uint64_t k = 0; k < n; ++k) BDR_REQUIRE(ixs[k] < r->capacity, "index %llu out of range", (unsigned long long)ixs[k]);
    uint64_t* d_ix = nullptr; float* d_td = nullptr;
    BDR_HIP(hipMalloc((void**)&d_ix, n * 8));
    hipError_t e = hipMalloc((void**)&d_td, n * 4);
    if (e == hipSuccess) e = hipMemcpyAsync(d_ix, ixs, n * 8, hipMemcpyHostToDevice, r->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_td, td_errs, n * 4, hipMemcpyHostToDevice, r->stream);
    int32_t rc = e == hipSuccess ? per_update(r->per, n, d_ix, d_td, r->stream) : fail(BDR_ERR_HIP, "update_priority copy failed: %s", hipGetErrorString(e));
    if (rc == BDR_OK) { (void)hipEventRecord(r->written, r->stream); mark_written(r); }
    (void)hipStreamSynchronize(r->stream);
    (void)hipFree(d_ix); (void)hipFree(d_td);
    if (rc == BDR_OK) rc = per_check(r->per);
    return rc;
}

int32_t bdr_replay_batch_weights(bdr_replay* r, uint64_t n, float* w_out)
{
    BDR_REQUIRE(r && w_out, "null argument");
    BDR_REQUIRE(r->per, "PER is not enabled on this buffer (weight: None)");
    BDR_REQUIRE(r->batch_n > 0 && n <= r->batch_n, "no batch of that size has been drawn");
    BDR_HIP(hipSetDevice(r->device));
    BDR_HIP(hipMemcpyAsync(w_out, per_weights(r->per), n * 4, hipMemcpyDeviceToHost, r->stream));
    BDR_HIP(hipStreamSynchronize(r->stream));
    return BDR_OK;
}

int32_t bdr_replay_per_info(bdr_replay* r, bdr_per_info* out)
{
    BDR_REQUIRE(r && out, "null argument");
    BDR_REQUIRE(r->per, "PER is not enabled on this buffer");
    BDR_HIP(hipSetDevice(r->device));
    per_info(r->per, out);
    float total = 0.f, mm[2];
    BDR_TRY(per_read(r->per, 0, &total, 1, r->stream));
    BDR_TRY(per_read(r->per, 1, mm, 2, r->stream));
    out->total = total; out->min_p = mm[0]; out->max_p = mm[1];   // transformed (p+eps)^alpha domain
    return per_check(r->per);
}

int32_t bdr_replay_per_read(bdr_replay* r, int32_t what, float* out, uint64_t n)
{
    BDR_REQUIRE(r && out, "null argument");
    BDR_REQUIRE(r->per, "PER is not enabled on this buffer");
    BDR_HIP(hipSetDevice(r->device));
    return per_read(r->per, what, out, n, r->stream);
}

int32_t bdr_replay_per_get(bdr_replay* r, float s, uint64_t* ix)
{
    BDR_REQUIRE(r && ix, "null argument");
    BDR_REQUIRE(r->per, "PER is not enabled on this buffer");
    BDR_HIP(hipSetDevice(r->device));
    return per_get(r->per, s, ix, r->stream);
}

}  // extern "C"
