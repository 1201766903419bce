// Shared host-side plumbing of libborder_amd.so: error reporting, HIP call checking,
// the opaque handle layouts.  gfx950 only; no CPU fallback anywhere in this library.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/border_amd.h"
#include "chacha.hpp"

namespace bdr {

extern thread_local char g_err[512];
extern thread_local int g_err_deferred;   // the last failure on this thread reports a device-side condition of an EARLIER, asynchronous step (bdr_last_error_is_deferred)

inline int32_t fail(int32_t code, const char* fmt, ...)
{
    g_err_deferred = 0;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

#define BDR_HIP(expr)                                                                              \
    do {                                                                                           \
        hipError_t e__ = (expr);                                                                   \
        if (e__ != hipSuccess)                                                                     \
            return ::bdr::fail(BDR_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), \
                               __FILE__, __LINE__);                                                \
    } while (0)

#define BDR_TRY(expr)                                                                              \
    do {                                                                                           \
        int32_t s__ = (expr);                                                                      \
        if (s__ != BDR_OK) return s__;                                                             \
    } while (0)

#define BDR_REQUIRE(cond, ...)                                                                     \
    do {                                                                                           \
        if (!(cond)) return ::bdr::fail(BDR_ERR_INVALID, __VA_ARGS__);                             \
    } while (0)

inline uint64_t round_up(uint64_t x, uint64_t a) { return (x + a - 1) / a * a; }

int32_t ensure_device(int32_t device);

// bdr_{dqn,iqn}_config::arithmetic -> "the large layers take the split-operand bf16 kernels".  `override_var` (=1 exact, =0 split;
// set but empty counts as 1, as in round 5) is read for A/B runs only: the boundary's field is what a caller states.
inline bool arith_is_split(int32_t arithmetic, const char* override_var)
{
    bool split = arithmetic != BDR_ARITH_F32_EXACT;
    if (const char* e = getenv(override_var)) split = e[0] == '0';
    return split;
}

}  // namespace bdr

// ---------------------------------------------------------------------------------------------
// Replay ring: one fused record per transition, record_stride bytes apart (multiple of 128):
//   [0, obs_bytes)                       obs
//   [obs_off2, obs_off2 + obs_bytes)     next_obs      (obs_off2 = round_up(obs_bytes, 16))
//   [act_off, act_off + act_bytes)       act
//   [tail_off + 0]  f32 reward, [+4] i8 is_terminated, [+5] i8 is_truncated
// ---------------------------------------------------------------------------------------------
struct bdr_per;   // per.hip

struct bdr_replay {
    int32_t device = 0;
    bdr_per* per = nullptr;    // prioritized replay state (nullptr: uniform sampling)
    uint64_t capacity = 0, i = 0, size = 0;
    uint64_t obs_bytes = 0, act_bytes = 0;
    uint64_t next_off = 0, act_off = 0, tail_off = 0, stride = 0;
    int32_t index_rng = 0;     // BDR_RNG_STDRNG / BDR_RNG_XOSHIRO256PP
    uint64_t* xo_state = nullptr;   // xoshiro256++: [XO_LANES][4] generator states in HBM (lane j draws sample j of every batch)
    uint32_t key[8] = {0};     // ChaCha12 key = seed_from_u64(seed)
    uint64_t word_pos = 0;     // next u32 word of the key stream (host-tracked, passed by value)
    uint8_t* ring = nullptr;   // capacity * stride bytes in HBM
    hipStream_t stream = nullptr;
    hipEvent_t written = nullptr;  // recorded after the last push/fill on `stream`
    hipEvent_t read = nullptr;     // recorded by the consumer after the last gather
    // Cross-stream ordering is lazy: a barrier packet (event record or wait) costs its queue a 5-7 us bubble
    // (tools/rocprof_timeline.py), so the opt loop's gather records / waits nothing unless a push, fill or tree update
    // actually happened in between.
    bool read_pending = false;          // a gather on read_stream may still be reading ring rows / owns the batch buffers
    hipStream_t read_stream = nullptr;
    uint64_t written_gen = 1;           // bumped with every record of `written`
    bool written_lazy = false;          // the newest write has not recorded `written` yet (wait_for_writer does it on the writer's stream)
    std::vector<std::pair<hipStream_t, uint64_t>> waited;   // consumer stream -> generation of `written` it has waited for
    // pinned staging for push
    uint8_t* stage = nullptr;
    uint64_t stage_records = 0;
    unsigned* done_host = nullptr; unsigned* done_dev = nullptr; unsigned done_seq = 0;   // pinned "small push finished" word (host view / device view) and its sequence number
    unsigned* done_ticket = nullptr;                        // device: records of the current small push that are finished
    unsigned dev_rows_checks = 0;                            // counts cache hits of dev_rows_ok (every 1024th one re-validates)
    const void* dev_rows_ok[2] = {nullptr, nullptr};       // bdr_replay_push_device: the obs / next_obs base addresses that passed the device-pointer check last
    uint8_t* d_tails = nullptr; uint64_t tails_cap = 0;   // bdr_replay_push_device: device copy of a run's act / reward / flags
    // device batch buffers (lazily sized)
    uint64_t batch_cap = 0, batch_n = 0;
    uint64_t uid = 0, batch_gen = 0;   // process-unique handle id; bumped whenever the batch buffers are re-allocated or flipped
    uint8_t *b_obs = nullptr, *b_next = nullptr, *b_act = nullptr;
    float* b_reward = nullptr;
    int8_t *b_term = nullptr, *b_trunc = nullptr;
    uint64_t* b_ixs = nullptr;
    // second set of batch buffers (replay_flip_batch): a consumer that gathers the next batch on another queue while kernels
    // of the previous update still read the current one alternates between the two sets
    struct BatchSet { uint8_t *obs = nullptr, *next = nullptr, *act = nullptr; float* reward = nullptr; int8_t *term = nullptr, *trunc = nullptr; uint64_t* ixs = nullptr; } alt;
    bool alt_valid = false;
    // ---- single-frame store (bdr_replay_config::frame_stack > 0): `ring` then holds only the small records
    //   [u32 frame slot x 2k (obs newest..oldest, next_obs newest..oldest) | act | reward f32 | is_terminated i8 | is_truncated i8]
    // and `frames` the frame store (frame_cap slots of frame_bytes, allocated in push order, slot = seq % frame_cap).
    int32_t frame_stack = 0;
    uint64_t frame_bytes = 0, frame_cap = 0, frame_seq = 0;   // frame_seq: frames allocated so far
    uint8_t* frames = nullptr;
    uint64_t rec_act_off = 0, rec_tail_off = 0;               // record layout (offsets of act / tail inside a record)
    std::vector<uint64_t> first_ref;       // per transition slot: smallest frame sequence number it references
    std::vector<uint8_t> last_next;        // host copy of the last pushed next_obs (episode continuation test)
    std::vector<uint64_t> last_next_seq;   // ... and the sequence numbers of its k frames
    bool have_last = false;
    uint8_t* fstage = nullptr;             // pinned staging for new frames
    uint64_t fstage_frames = 0;
};

namespace bdr {
// Enqueue "draw n indices + gather" on `stream` (the consumer's stream).  Handles the
// cross-stream ordering against pushes.  Advances the RNG like one batch(n).
int32_t replay_sample_on_stream(bdr_replay* r, uint64_t n, hipStream_t stream);
int32_t replay_prepare_sample(bdr_replay* r, uint64_t n, hipStream_t stream);

// arguments of the gather kernels (replay.hip k_gather) - also handed to agents whose own kernel does the gather (replay_sample_plan)
struct GatherArgs {
    const uint8_t* ring;
    uint64_t stride, obs_bytes, act_bytes, next_off, act_off, tail_off;
    uint64_t* ixs;           // [n] sampled indices (written by chunk 0 of every sample)
    ChaChaKey key;           // K1 fused: every workgroup draws its own index (wave-uniform -> scalar unit)
    uint64_t word_pos, size;
    uint8_t *b_obs, *b_next, *b_act;
    float* b_reward;
    int8_t *b_term, *b_trunc;
    uint32_t chunks;      // workgroups per sample
    uint32_t vec_per_chunk;  // 16-byte vectors per chunk
    uint32_t given;       // 1: ixs[] was produced by the PER sampler, gather those rows
};
int32_t replay_sample_plan(bdr_replay* r, uint64_t n, hipStream_t stream, GatherArgs* out);
int32_t replay_ensure_batch_capacity(bdr_replay* r, uint64_t n);
// Consumer streams a replay buffer may have to record an event on later (lazy ordering): registered when they first sample,
// retired by their owner after synchronising and before hipStreamDestroy, so that a buffer never touches a dead handle.
void stream_register(hipStream_t s);
void stream_retire(hipStream_t s);
bool stream_alive(hipStream_t s);
bdr_replay* replay_lookup(uint64_t uid);   // live handle with this uid, or nullptr once it was destroyed
int32_t replay_per_check(uint64_t uid, bool* alive);   // per_check of that handle while the registry lock is held
// the host state a sample advances (uniform ring): snapshot / restore around a step-graph pass that may have to be re-enqueued
struct ReplaySnap {
    uint64_t word_pos, batch_n; bool read_pending; hipStream_t read_stream;
    explicit ReplaySnap(const bdr_replay* r) : word_pos(r->word_pos), batch_n(r->batch_n), read_pending(r->read_pending), read_stream(r->read_stream) {}
    void restore(bdr_replay* r) const { r->word_pos = word_pos; r->batch_n = batch_n; r->read_pending = read_pending; r->read_stream = read_stream; }
};
int32_t replay_flip_batch(bdr_replay* r, uint64_t n);   // makes the other buffer set current (allocated on first use)

// per.hip
int32_t per_create(const bdr_per_config* c, uint64_t capacity, hipStream_t stream, bdr_per** out);
void per_destroy(bdr_per* p);
int32_t per_push(bdr_per* p, uint64_t i0, uint64_t len, hipStream_t st);
int32_t per_sample(bdr_per* p, const uint32_t key[8], uint64_t word_pos, uint64_t n, uint64_t* ixs_dev, hipStream_t st);
int32_t per_update(bdr_per* p, uint64_t n, const uint64_t* ixs_dev, const float* td_dev, hipStream_t st);
const float* per_weights(const bdr_per* p);
int32_t per_read(const bdr_per* p, int32_t what, float* out, uint64_t n, hipStream_t st);
int32_t per_get(const bdr_per* p, float s, uint64_t* ix, hipStream_t st);
void per_info(const bdr_per* p, bdr_per_info* o);
int32_t per_check(bdr_per* p);   // non-finite priorities flagged by the update kernels (the reference panics); call after a sync
// weights of the batch last drawn on the consumer's stream (nullptr without PER), and the priority update
// an agent enqueues after its backward pass (device arrays, the agent's stream)
inline const float* replay_batch_weights(const bdr_replay* r) { return r->per ? per_weights(r->per) : nullptr; }
int32_t replay_update_priority_on_stream(bdr_replay* r, uint64_t n, const float* td_dev, hipStream_t stream);
}  // namespace bdr
