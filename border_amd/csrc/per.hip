// Prioritized experience replay on the device: the `per_config: Some(..)` branch of SimpleReplayBuffer
// (border-core/src/generic_replay_buffer/base.rs:227-235 set_priority, :376-383 batch, :413-426
// update_priority) over the SumTree of base/sum_tree.rs and the IwScheduler of base/iw_scheduler.rs.
//
// Parity constraints that shape the kernels:
//  * The sum tree is INCREMENTAL in f32 (`tree[parent] += change`, sum_tree.rs:46-52), so a node's value is
//    the result of one particular sequence of float additions.  Updates of a batch are applied by the
//    reference one after the other (base.rs:421-423); k_per_propagate keeps that order PER NODE (different
//    nodes are independent): the first update that touches a node owns it and folds every later change to
//    the same node in batch order.  Duplicate indices inside a batch (sampling is with replacement) chain
//    through the leaf exactly like the sequential loop does.
//  * The tree keeps the reference's array layout for ANY capacity (2*capacity-1 nodes, leaves at
//    capacity-1+ix, parent (i-1)/2), because the shape decides which partial sums exist.
//  * min / max (segment-tree SegmentPoint with Min/MaxIgnoreNaN) are exact operations: they live in two
//    ordinary power-of-two tournament trees and are order-independent.
//  * powf: `(p+eps).powf(alpha)` etc. are evaluated as (float)pow((double)x,(double)y).
//  * The batch's uniforms come from the buffer's StdRng stream, one u32 per sample, f32 = (w >> 9) * 2^-23
//    (the reference calls the unseeded fastrand::f32()).
#include <algorithm>
#include <cfloat>

#include "chacha.hpp"
#include "common.hpp"

using namespace bdr;

struct bdr_per {
    float alpha = 0.6f, eps = 1e-8f;
    float beta_0 = 0.4f, beta_final = 1.0f;
    uint64_t n_opts_final = 500000, n_opts = 0;
    int32_t normalize = BDR_PER_NORMALIZE_ALL;
    uint64_t capacity = 0, n_samples = 0, p2 = 1;
    int maxdepth = 0;          // depth of the deepest leaf of the sum tree
    float* tree = nullptr;     // [2*capacity-1]
    float* mint = nullptr;     // [2*p2], root at 1, leaf ix at p2+ix
    float* maxt = nullptr;
    // scratch of one update batch (<= PER_CHUNK entries)
    uint64_t* u_ix = nullptr;
    float* u_p = nullptr;
    float* u_change = nullptr;
    float* u_praw = nullptr;   // SumTree::max() broadcast for a push
    float* w = nullptr;        // importance weights of the last batch [batch_cap]
    uint64_t w_cap = 0;
};

namespace {

constexpr int PER_CHUNK = 1024;

__device__ __forceinline__ float powf_ref(float x, float y) { return (float)pow((double)x, (double)y); }
__device__ __forceinline__ int node_depth(uint64_t i) { return 63 - __clzll((long long)(i + 1)); }

__global__ void k_per_init(float* mint, float* maxt, uint64_t p2, uint64_t capacity)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * p2) return;
    // leaves: min f32::MAX (sum_tree.rs:40), max 1e-8 (:41); padding leaves are neutral; inner nodes follow
    const bool leaf = i >= p2;
    const bool real = leaf && (i - p2) < capacity;
    mint[i] = FLT_MAX;
    maxt[i] = leaf ? (real ? 1e-8f : -FLT_MAX) : 1e-8f;
}

// One batch of updates, part 1 (one workgroup): transformed priorities, per-update change of the leaf,
// the leaves themselves, and the min / max tournament trees.
struct PrepArgs {
    float* tree; float* mint; float* maxt;
    uint64_t capacity, p2;
    const uint64_t* ixs;      // [n] (device) or nullptr: consecutive rows ix0, ix0+1, ... (mod capacity)
    uint64_t ix0;
    const float* p_raw;       // [n] raw priorities (|td| of update_priority, or SumTree::max() of a push)
    float alpha, eps;
    int n;
    uint64_t* u_ix; float* u_p; float* u_change;
};
__global__ __launch_bounds__(256) void k_per_prepare(PrepArgs a)
{
    __shared__ uint64_t s_ix[PER_CHUNK];
    __shared__ float s_p[PER_CHUNK];
    for (int k = threadIdx.x; k < a.n; k += 256) {
        const uint64_t ix = a.ixs ? a.ixs[k] : (a.ix0 + (uint64_t)k) % a.capacity;
        const float p = a.p_raw[k];
        s_ix[k] = ix;
        s_p[k] = powf_ref(p + a.eps, a.alpha);            // update(): (p + eps).powf(alpha)  (:96)
    }
    __syncthreads();
    for (int k = threadIdx.x; k < a.n; k += 256) {
        const uint64_t ix = s_ix[k];
        int prev = -1;
        bool last = true;
        for (int j = k - 1; j >= 0; --j) if (s_ix[j] == ix) { prev = j; break; }
        for (int j = k + 1; j < a.n; ++j) if (s_ix[j] == ix) { last = false; break; }
        const uint64_t leaf = ix + a.capacity - 1;
        const float old = prev >= 0 ? s_p[prev] : a.tree[leaf];
        a.u_ix[k] = ix; a.u_p[k] = s_p[k];
        a.u_change[k] = s_p[k] - old;                     // change = p - tree[ix]  (:100)
        if (last) { a.mint[a.p2 + ix] = s_p[k]; a.maxt[a.p2 + ix] = s_p[k]; }
    }
    __syncthreads();
    // leaves are assigned (not accumulated): tree[ix] = p (:105).  Written after every `old` has been read.
    for (int k = threadIdx.x; k < a.n; k += 256) {
        bool last = true;
        for (int j = k + 1; j < a.n; ++j) if (s_ix[j] == s_ix[k]) { last = false; break; }
        if (last) a.tree[s_ix[k] + a.capacity - 1] = s_p[k];
    }
    // tournament trees, level by level (one workgroup: __syncthreads orders the levels)
    for (uint64_t width = a.p2 >> 1, shift = 1; width >= 1; width >>= 1, ++shift) {
        __threadfence_block();
        __syncthreads();
        for (int k = threadIdx.x; k < a.n; k += 256) {
            const uint64_t node = (a.p2 + s_ix[k]) >> shift;
            a.mint[node] = fminf(a.mint[2 * node], a.mint[2 * node + 1]);
            a.maxt[node] = fmaxf(a.maxt[2 * node], a.maxt[2 * node + 1]);
        }
    }
}

// part 2: workgroup d folds the changes into the ancestors at depth d, in batch order per node.
struct PropArgs { float* tree; uint64_t capacity; const uint64_t* u_ix; const float* u_change; int n; };
__global__ __launch_bounds__(256) void k_per_propagate(PropArgs a)
{
    __shared__ long long s_node[PER_CHUNK];
    __shared__ float s_change[PER_CHUNK];
    const int d = blockIdx.x;
    for (int k = threadIdx.x; k < a.n; k += 256) {
        const uint64_t leaf = a.u_ix[k] + a.capacity - 1;
        const int dl = node_depth(leaf);
        s_node[k] = d < dl ? (long long)(((leaf + 1) >> (dl - d)) - 1) : -1;   // ancestor of `leaf` at depth d
        s_change[k] = a.u_change[k];
    }
    __syncthreads();
    for (int k = threadIdx.x; k < a.n; k += 256) {
        const long long node = s_node[k];
        if (node < 0) continue;
        bool first = true;
        for (int j = 0; j < k; ++j) if (s_node[j] == node) { first = false; break; }
        if (!first) continue;
        float v = a.tree[node];
        for (int j = k; j < a.n; ++j) if (s_node[j] == node) v += s_change[j];   // tree[parent] += change (:48)
        a.tree[node] = v;
    }
}

// SumTree::sample (sum_tree.rs:120-157): one thread per sample; one workgroup (n <= 1024).
struct SampleArgs {
    const float* tree; const float* mint;
    uint64_t capacity, n_samples;
    ChaChaKey key; uint64_t word_pos;
    float beta; int normalize; int n;
    uint64_t* ixs; float* w;
};
__global__ __launch_bounds__(1024) void k_per_sample(SampleArgs a)
{
    __shared__ float s_red[1024];
    const int k = threadIdx.x;
    const uint64_t len = 2 * a.capacity - 1;
    const float p_sum = a.tree[0];
    const float nn = (float)a.n_samples / p_sum;                       // :131
    float wk = -FLT_MAX;
    if (k < a.n) {
        const float u = (float)(chacha12_word(a.key, a.word_pos + k) >> 9) * (1.0f / 8388608.0f);
        float s = p_sum * u;                                           // :122-124
        uint64_t ix = 0;
        for (;;) {                                                     // retrieve (:54-66)
            const uint64_t left = 2 * ix + 1, right = left + 1;
            if (left >= len) break;
            const float tl = a.tree[left];
            if (s <= tl || a.tree[right] == 0.f) ix = left;
            else { s -= tl; ix = right; }
        }
        a.ixs[k] = ix + 1 - a.capacity;                                // get (:110-114)
        wk = powf_ref(nn * a.tree[ix], -a.beta);                       // :132-136
    }
    float w_max_inv;
    if (a.normalize == BDR_PER_NORMALIZE_ALL) {
        w_max_inv = powf_ref(nn * a.mint[1], a.beta);                  // :140 (min over [0, n_samples))
    } else {
        s_red[k] = wk;
        __syncthreads();
        for (int off = 512; off > 0; off >>= 1) {
            if (k < off) s_red[k] = fmaxf(s_red[k], s_red[k + off]);
            __syncthreads();
        }
        w_max_inv = 1.0f / s_red[0];                                   // :141
    }
    if (k < a.n) a.w[k] = wk * w_max_inv;
}

// SumTree::max() (sum_tree.rs:72-76) = root of the max tree ^ (1/alpha), broadcast to a chunk of raw priorities
__global__ void k_per_fill_max(float* out, const float* maxt, float alpha, int n)
{
    const float v = powf_ref(maxt[1], 1.0f / alpha);
    for (int i = threadIdx.x; i < n; i += blockDim.x) out[i] = v;
}

__global__ void k_per_get(const float* tree, uint64_t capacity, float s, uint64_t* out)
{
    const uint64_t len = 2 * capacity - 1;
    uint64_t ix = 0;
    for (;;) {
        const uint64_t left = 2 * ix + 1, right = left + 1;
        if (left >= len) break;
        if (s <= tree[left] || tree[right] == 0.f) ix = left;
        else { s -= tree[left]; ix = right; }
    }
    *out = ix + 1 - capacity;
}

}  // namespace

namespace bdr {

void per_destroy(bdr_per* p)
{
    if (!p) return;
    (void)hipFree(p->tree); (void)hipFree(p->mint); (void)hipFree(p->maxt);
    (void)hipFree(p->u_ix); (void)hipFree(p->u_p); (void)hipFree(p->u_change); (void)hipFree(p->u_praw); (void)hipFree(p->w);
    delete p;
}

int32_t per_create(const bdr_per_config* c, uint64_t capacity, hipStream_t stream, bdr_per** out)
{
    BDR_REQUIRE(c->alpha > 0.f, "alpha must be positive");
    BDR_REQUIRE(c->normalize == BDR_PER_NORMALIZE_ALL || c->normalize == BDR_PER_NORMALIZE_BATCH, "unknown weight normalizer");
    BDR_REQUIRE(capacity >= 2, "PER needs capacity >= 2");
    bdr_per* p = new bdr_per();
    p->alpha = c->alpha; p->beta_0 = c->beta_0; p->beta_final = c->beta_final; p->n_opts_final = c->n_opts_final;
    p->normalize = c->normalize; p->capacity = capacity;
    while (p->p2 < capacity) p->p2 <<= 1;
    p->maxdepth = 63 - __builtin_clzll(2 * capacity - 1);   // depth of the last leaf (array index 2C-2)
    auto fail_free = [&](hipError_t e) { per_destroy(p); return fail(BDR_ERR_HIP, "PER allocation failed: %s", hipGetErrorString(e)); };
    hipError_t e;
    if ((e = hipMalloc((void**)&p->tree, (2 * capacity - 1) * 4)) != hipSuccess) return fail_free(e);
    if ((e = hipMalloc((void**)&p->mint, 2 * p->p2 * 4)) != hipSuccess) return fail_free(e);
    if ((e = hipMalloc((void**)&p->maxt, 2 * p->p2 * 4)) != hipSuccess) return fail_free(e);
    if ((e = hipMalloc((void**)&p->u_ix, PER_CHUNK * 8)) != hipSuccess) return fail_free(e);
    if ((e = hipMalloc((void**)&p->u_p, PER_CHUNK * 4)) != hipSuccess) return fail_free(e);
    if ((e = hipMalloc((void**)&p->u_change, PER_CHUNK * 4)) != hipSuccess) return fail_free(e);
    if ((e = hipMalloc((void**)&p->u_praw, PER_CHUNK * 4)) != hipSuccess) return fail_free(e);
    if ((e = hipMemsetAsync(p->tree, 0, (2 * capacity - 1) * 4, stream)) != hipSuccess) return fail_free(e);
    hipLaunchKernelGGL(k_per_init, dim3((unsigned)((2 * p->p2 + 255) / 256)), dim3(256), 0, stream, p->mint, p->maxt, p->p2, capacity);
    if ((e = hipGetLastError()) != hipSuccess) return fail_free(e);
    *out = p;
    return BDR_OK;
}

// n <= PER_CHUNK updates in batch order.  ixs == nullptr: consecutive rows ix0.. (push)
static int32_t per_apply(bdr_per* p, int n, const uint64_t* ixs, uint64_t ix0, const float* p_raw, hipStream_t st)
{
    PrepArgs a{p->tree, p->mint, p->maxt, p->capacity, p->p2, ixs, ix0, p_raw, p->alpha, p->eps, n, p->u_ix, p->u_p, p->u_change};
    hipLaunchKernelGGL(k_per_prepare, dim3(1), dim3(256), 0, st, a);
    BDR_HIP(hipGetLastError());
    PropArgs b{p->tree, p->capacity, p->u_ix, p->u_change, n};
    hipLaunchKernelGGL(k_per_propagate, dim3((unsigned)p->maxdepth), dim3(256), 0, st, b);
    BDR_HIP(hipGetLastError());
    return BDR_OK;
}

// set_priority (base.rs:227-235): `len` rows starting at ring position i0 get the current maximum priority.
// SumTree::max() is taken ONCE before the loop (:229), so it is materialised first; the rows are then added
// in order (chunks of PER_CHUNK keep that order).
int32_t per_push(bdr_per* p, uint64_t i0, uint64_t len, hipStream_t st)
{
    hipLaunchKernelGGL(k_per_fill_max, dim3(1), dim3(256), 0, st, p->u_praw, p->maxt, p->alpha, PER_CHUNK);
    BDR_HIP(hipGetLastError());
    for (uint64_t done = 0; done < len; done += PER_CHUNK) {
        const int m = (int)std::min<uint64_t>(PER_CHUNK, len - done);
        BDR_TRY(per_apply(p, m, nullptr, (i0 + done) % p->capacity, p->u_praw, st));
    }
    p->n_samples = std::min(p->capacity, p->n_samples + len);       // add(): n_samples += 1 up to capacity (:86-88)
    return BDR_OK;
}

float per_beta(const bdr_per* p)                                    // iw_scheduler.rs:35-43
{
    if (p->n_opts >= p->n_opts_final) return p->beta_final;
    const float d = p->beta_final - p->beta_0;
    return p->beta_0 + d * ((float)p->n_opts / (float)p->n_opts_final);
}

int32_t per_sample(bdr_per* p, const uint32_t key[8], uint64_t word_pos, uint64_t n, uint64_t* ixs_dev, hipStream_t st)
{
    BDR_REQUIRE(n <= 1024, "PER batches are limited to 1024 samples");
    if (p->w_cap < n) {
        if (p->w) { BDR_HIP(hipStreamSynchronize(st)); BDR_HIP(hipFree(p->w)); p->w = nullptr; }
        BDR_HIP(hipMalloc((void**)&p->w, std::max<uint64_t>(n, 256) * 4));
        p->w_cap = std::max<uint64_t>(n, 256);
    }
    SampleArgs a{};
    a.tree = p->tree; a.mint = p->mint; a.capacity = p->capacity; a.n_samples = p->n_samples;
    memcpy(a.key.k, key, sizeof a.key.k); a.word_pos = word_pos;
    a.beta = per_beta(p); a.normalize = p->normalize; a.n = (int)n; a.ixs = ixs_dev; a.w = p->w;
    hipLaunchKernelGGL(k_per_sample, dim3(1), dim3(1024), 0, st, a);
    BDR_HIP(hipGetLastError());
    return BDR_OK;
}

// update_priority (base.rs:413-426): sum_tree.update(ix, td_err) in batch order, then add_n_opts()
int32_t per_update(bdr_per* p, uint64_t n, const uint64_t* ixs_dev, const float* td_dev, hipStream_t st)
{
    for (uint64_t done = 0; done < n; done += PER_CHUNK) {
        const int m = (int)std::min<uint64_t>(PER_CHUNK, n - done);
        BDR_TRY(per_apply(p, m, ixs_dev + done, 0, td_dev + done, st));
    }
    p->n_opts += 1;
    return BDR_OK;
}

const float* per_weights(const bdr_per* p) { return p->w; }

int32_t per_read(const bdr_per* p, int32_t what, float* out, uint64_t n, hipStream_t st)
{
    const float* src = nullptr; uint64_t have = 0;
    switch (what) {
        case 0: src = p->tree; have = 2 * p->capacity - 1; break;
        case 1: src = p->mint; have = 2 * p->p2; break;
        case 2: src = p->maxt; have = 2 * p->p2; break;
        default: return fail(BDR_ERR_INVALID, "unknown PER array %d", what);
    }
    BDR_REQUIRE(n <= have, "PER array has %llu elements", (unsigned long long)have);
    BDR_HIP(hipMemcpyAsync(out, src, n * 4, hipMemcpyDeviceToHost, st));
    BDR_HIP(hipStreamSynchronize(st));
    return BDR_OK;
}

int32_t per_get(const bdr_per* p, float s, uint64_t* ix, hipStream_t st)
{
    uint64_t* d = nullptr;
    BDR_HIP(hipMalloc((void**)&d, 8));
    hipLaunchKernelGGL(k_per_get, dim3(1), dim3(1), 0, st, p->tree, p->capacity, s, d);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(ix, d, 8, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(d);
    BDR_HIP(e);
    return BDR_OK;
}

void per_info(const bdr_per* p, bdr_per_info* o)
{
    o->n_samples = p->n_samples; o->n_opts = p->n_opts; o->beta = per_beta(p);
}

}  // namespace bdr
