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
//  * min / max (segment-tree SegmentPoint with Min/MaxIgnoreNaN) are exact, order-independent operations: they are
//    recomputed when needed by one coalesced pass over the leaves (k_per_minmax: 4 MB for 1 M transitions, a few
//    microseconds) instead of being maintained by dependent read-modify-writes up a tree.
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
    uint64_t capacity = 0, n_samples = 0;
    int maxdepth = 0;          // depth of the deepest leaf of the sum tree
    float* tree = nullptr;     // [2*capacity-1]
    unsigned* mm = nullptr;    // {min of leaves [0, n_samples), max of all leaves} as bit patterns (k_per_minmax); mm[2]: NaN flag
    bool mm_valid = false;     // mm holds the min/max of the CURRENT leaves (computed off the critical path after an update)
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

// min over the leaves [0, n_samples) (sum_tree.rs:140 `min_tree.query(0, n_samples)`) and max over every leaf
// (:72-76 `max_tree.query(0, len)`; never-set leaves hold the tree's 0, below the reference's initial 1e-8).
// mm[0] / mm[1] are the bit patterns of the running min / max (positive floats order like unsigned integers);
// whoever consumes them re-arms them (mm_take), so no memset sits between the kernels.
constexpr unsigned MM_MIN_INIT = 0x7f7fffffu;   // f32::MAX: MinIgnoreNaN identity (:40)
constexpr unsigned MM_MAX_INIT = 0x322bcc77u;   // 1e-8f: initial value of every max-tree leaf (:41)
__global__ __launch_bounds__(256) void k_per_minmax(const float* __restrict__ leaves, uint64_t n_samples, uint64_t capacity,
                                                    unsigned* __restrict__ mm)
{
    __shared__ float smin[256], smax[256];
    float v[8];
    const uint64_t base = (uint64_t)blockIdx.x * 2048 + threadIdx.x;
#pragma unroll
    for (int u = 0; u < 8; ++u) { const uint64_t i = base + (uint64_t)u * 256; v[u] = i < capacity ? leaves[i] : 0.f; }
    float mn = FLT_MAX, mx = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        mx = fmaxf(mx, v[u]);
        if (base + (uint64_t)u * 256 < n_samples) mn = fminf(mn, v[u]);
    }
    smin[threadIdx.x] = mn; smax[threadIdx.x] = mx;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            smin[threadIdx.x] = fminf(smin[threadIdx.x], smin[threadIdx.x + off]);
            smax[threadIdx.x] = fmaxf(smax[threadIdx.x], smax[threadIdx.x + off]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        atomicMin(&mm[0], __float_as_uint(smin[0]));
        atomicMax(&mm[1], __float_as_uint(smax[0]));
    }
}
// read {min, max} (every thread of the single consuming workgroup), then re-arm them for the next k_per_minmax
__device__ __forceinline__ void mm_take(unsigned* mm, float& mn, float& mx)
{
    mn = __uint_as_float(mm[0]); mx = __uint_as_float(mm[1]);
    __syncthreads();
    if (threadIdx.x == 0) { mm[0] = MM_MIN_INIT; mm[1] = MM_MAX_INIT; }
}

// One batch of updates, part 1 (one workgroup): transformed priorities, per-update change of the leaf, and the
// leaves themselves.
struct PrepArgs {
    float* tree;
    uint64_t capacity;
    const uint64_t* ixs;      // [n] (device) or nullptr: consecutive rows ix0, ix0+1, ... (mod capacity)
    uint64_t ix0;
    const float* p_raw;       // [n] raw priorities (|td| of update_priority, or SumTree::max() of a push)
    float alpha, eps;
    int n;
    uint64_t* u_ix; float* u_p; float* u_change;
    unsigned* mm;             // re-armed here: the leaves are about to change; mm[2] is raised when a change is NaN
};
__global__ __launch_bounds__(256) void k_per_prepare(PrepArgs a)
{
    if (threadIdx.x == 0) { a.mm[0] = MM_MIN_INIT; a.mm[1] = MM_MAX_INIT; }
    __shared__ uint32_t s_ix[PER_CHUNK + 8];   // capacity < 2^31 (per_create)
    __shared__ float s_p[PER_CHUNK];
    for (int k = threadIdx.x; k < PER_CHUNK + 8; k += 256) s_ix[k] = 0xffffffffu;   // padding never matches
    __syncthreads();
    for (int k = threadIdx.x; k < a.n; k += 256) {
        const uint64_t ix = a.ixs ? a.ixs[k] : (a.ix0 + (uint64_t)k) % a.capacity;
        s_ix[k] = (uint32_t)ix;
        s_p[k] = powf_ref(a.p_raw[k] + a.eps, a.alpha);   // update(): (p + eps).powf(alpha)  (:96)
    }
    __syncthreads();
    float old[PER_CHUNK / 256]; bool last[PER_CHUNK / 256];
#pragma unroll
    for (int it = 0; it < PER_CHUNK / 256; ++it) {
        const int k = threadIdx.x + it * 256;
        if (k >= a.n) continue;
        const uint32_t ix = s_ix[k];
        // duplicates of ix in the batch: the latest one before k, and whether one follows (break-free scan, 8 keys
        // per iteration, so the LDS reads pipeline instead of paying their latency one by one)
        int prev = -1;
        bool lst = true;
        for (int j0 = 0; j0 < a.n; j0 += 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int j = j0 + u;
                const bool m = s_ix[j] == ix;
                prev = (m && j < k) ? j : prev;
                lst = lst && !(m && j > k);
            }
        }
        old[it] = prev >= 0 ? s_p[prev] : a.tree[(uint64_t)ix + a.capacity - 1];
        last[it] = lst;
    }
    __syncthreads();   // every `old` leaf has been read before any leaf is written
#pragma unroll
    for (int it = 0; it < PER_CHUNK / 256; ++it) {
        const int k = threadIdx.x + it * 256;
        if (k >= a.n) continue;
        a.u_ix[k] = s_ix[k]; a.u_p[k] = s_p[k];
        float change = s_p[k] - old[it];                                       // change = p - tree[ix]  (:100)
        // `if change.is_nan() { panic!() }` (:101-104): the library never aborts - the update is dropped (leaf and sums keep
        // their values, so the tree stays usable) and a flag is raised that the next host synchronisation turns into an error
        const bool bad = change != change;
        if (bad) { atomicOr(a.mm + 2, 1u); change = 0.f; }
        a.u_change[k] = change;
        if (last[it] && !bad) a.tree[(uint64_t)s_ix[k] + a.capacity - 1] = s_p[k];   // leaves are assigned, not accumulated (:105)
    }
}

// part 2: workgroup d folds the changes into the ancestors at depth d, in batch order per node.
struct PropArgs { float* tree; uint64_t capacity; const uint64_t* u_ix; const float* u_change; int n; };
__global__ __launch_bounds__(256) void k_per_propagate(PropArgs a)
{
    __shared__ uint32_t s_node[PER_CHUNK + 8];   // node ids < 2*capacity < 2^32; 0xffffffff = none
    __shared__ float s_change[PER_CHUNK + 8];
    const int d = blockIdx.x;
    for (int k = threadIdx.x; k < PER_CHUNK + 8; k += 256) { s_node[k] = 0xffffffffu; s_change[k] = 0.f; }
    __syncthreads();
    for (int k = threadIdx.x; k < a.n; k += 256) {
        const uint64_t leaf = a.u_ix[k] + a.capacity - 1;
        const int dl = node_depth(leaf);
        s_node[k] = d < dl ? (uint32_t)(((leaf + 1) >> (dl - d)) - 1) : 0xffffffffu;   // ancestor of `leaf` at depth d
        s_change[k] = a.u_change[k];
    }
    __syncthreads();
    for (int k = threadIdx.x; k < a.n; k += 256) {
        const uint32_t node = s_node[k];
        if (node == 0xffffffffu) continue;
        bool first = true;                       // break-free: 8 keys per iteration
        for (int j0 = 0; j0 < k; j0 += 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) first = first && !(j0 + u < k && s_node[j0 + u] == node);
        }
        if (!first) continue;
        float v = a.tree[node];
        for (int j0 = k; j0 < a.n; j0 += 8) {   // tree[parent] += change (:48), in batch order
#pragma unroll
            for (int u = 0; u < 8; ++u) v = s_node[j0 + u] == node ? v + s_change[j0 + u] : v;
        }
        a.tree[node] = v;
    }
}

// SumTree::sample (sum_tree.rs:120-157): one thread per sample; one workgroup (n <= 1024).
// The descent is a chain of dependent loads; the top PER_TOP levels are staged in LDS in one round trip, below that
// every round trip fetches children, grandchildren and great-grandchildren (contiguous in the heap layout).
constexpr int PER_TOP = 13;                      // nodes 0 .. 2^13 - 2 (32 KB)
struct SampleArgs {
    const float* tree; unsigned* mm;
    uint64_t capacity, n_samples;
    ChaChaKey key; uint64_t word_pos;
    float beta; int normalize; int n;
    uint64_t* ixs; float* w;
};
__global__ __launch_bounds__(1024) void k_per_sample(SampleArgs a)
{
    __shared__ float s_top[(1 << PER_TOP) - 1];
    __shared__ float s_red[1024];
    const int k = threadIdx.x;
    const uint64_t len = 2 * a.capacity - 1;
    const uint64_t ntop = len < (uint64_t)((1 << PER_TOP) - 1) ? len : (uint64_t)((1 << PER_TOP) - 1);
    for (uint64_t i = k; i < ntop; i += 1024) s_top[i] = a.tree[i];
    float mn = FLT_MAX, mx;
    if (a.normalize == BDR_PER_NORMALIZE_ALL) mm_take(a.mm, mn, mx);
    __syncthreads();
    const float p_sum = s_top[0];
    const float nn = (float)a.n_samples / p_sum;                       // :131
    float wk = -FLT_MAX;
    if (k < a.n) {
        const float u = (float)(chacha12_word(a.key, a.word_pos + k) >> 9) * (1.0f / 8388608.0f);
        float s = p_sum * u;                                           // :122-124
        uint64_t ix = 0;
        // retrieve (:54-66): `s <= tree[left] || tree[right] == 0` -> left, else s -= tree[left], right
        auto step = [&](float tl, float tr) {
            if (s <= tl || tr == 0.f) ix = 2 * ix + 1;
            else { s -= tl; ix = 2 * ix + 2; }
        };
        while (2 * ix + 2 < ntop) step(s_top[2 * ix + 1], s_top[2 * ix + 2]);   // both children in LDS
        for (;;) {
            if (2 * ix + 1 >= len) break;
            // the 2 children, 4 grandchildren and 8 great-grandchildren of ix are three contiguous runs of the heap:
            // one round trip resolves up to three levels
            const uint64_t c0 = 2 * ix + 1, g0 = 4 * ix + 3, h0 = 8 * ix + 7;
            const bool lv2 = g0 + 3 < len, lv3 = h0 + 7 < len;
            float c[2], g[4], h[8];
            c[0] = a.tree[c0]; c[1] = a.tree[c0 + 1];
#pragma unroll
            for (int q = 0; q < 4; ++q) g[q] = lv2 ? a.tree[g0 + q] : 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) h[q] = lv3 ? a.tree[h0 + q] : 0.f;
            const uint64_t base = ix;
            step(c[0], c[1]);
            if (!lv2) continue;
            const int s1 = (int)(ix - (2 * base + 1));                 // 0 left, 1 right
            step(g[2 * s1], g[2 * s1 + 1]);
            if (!lv3) continue;
            const int s2 = (int)(ix - (4 * base + 3));                 // 0..3
            step(h[2 * s2], h[2 * s2 + 1]);
        }
        a.ixs[k] = ix + 1 - a.capacity;                                // get (:110-114)
        wk = powf_ref(nn * a.tree[ix], -a.beta);                       // :132-136
    }
    float w_max_inv;
    if (a.normalize == BDR_PER_NORMALIZE_ALL) {
        w_max_inv = powf_ref(nn * mn, a.beta);                         // :140 (min over [0, n_samples))
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

// SumTree::max() (sum_tree.rs:72-76) = root of the max tree ^ (1/alpha), broadcast to a chunk of raw priorities
__global__ void k_per_fill_max(float* out, unsigned* mm, float alpha, int n)
{
    float mn, mx;
    mm_take(mm, mn, mx);
    const float v = powf_ref(mx, 1.0f / alpha);
    for (int i = threadIdx.x; i < n; i += blockDim.x) out[i] = v;
}

}  // namespace

namespace bdr {

void per_destroy(bdr_per* p)
{
    if (!p) return;
    (void)hipFree(p->tree); (void)hipFree(p->mm);
    (void)hipFree(p->u_ix); (void)hipFree(p->u_p); (void)hipFree(p->u_change); (void)hipFree(p->u_praw); (void)hipFree(p->w);
    delete p;
}

int32_t per_create(const bdr_per_config* c, uint64_t capacity, hipStream_t stream, bdr_per** out)
{
    BDR_REQUIRE(c->alpha > 0.f, "alpha must be positive");
    BDR_REQUIRE(c->normalize == BDR_PER_NORMALIZE_ALL || c->normalize == BDR_PER_NORMALIZE_BATCH, "unknown weight normalizer");
    BDR_REQUIRE(capacity >= 2 && capacity < (1ull << 31), "PER needs 2 <= capacity < 2^31");
    bdr_per* p = new bdr_per();
    p->alpha = c->alpha; p->beta_0 = c->beta_0; p->beta_final = c->beta_final; p->n_opts_final = c->n_opts_final;
    p->normalize = c->normalize; p->capacity = capacity;
    p->maxdepth = 63 - __builtin_clzll(2 * capacity - 1);   // depth of the last leaf (array index 2C-2)
    auto fail_free = [&](hipError_t e) { per_destroy(p); return fail(BDR_ERR_HIP, "PER allocation failed: %s", hipGetErrorString(e)); };
    hipError_t e;
    if ((e = hipMalloc((void**)&p->tree, (2 * capacity - 1) * 4)) != hipSuccess) return fail_free(e);
    if ((e = hipMalloc((void**)&p->mm, 4 * 4)) != hipSuccess) return fail_free(e);
    { const unsigned init[4] = {MM_MIN_INIT, MM_MAX_INIT, 0u, 0u}; if ((e = hipMemcpyAsync(p->mm, init, 16, hipMemcpyHostToDevice, stream)) != hipSuccess) return fail_free(e); if ((e = hipStreamSynchronize(stream)) != hipSuccess) return fail_free(e); }
    if ((e = hipMalloc((void**)&p->u_ix, PER_CHUNK * 8)) != hipSuccess) return fail_free(e);
    if ((e = hipMalloc((void**)&p->u_p, PER_CHUNK * 4)) != hipSuccess) return fail_free(e);
    if ((e = hipMalloc((void**)&p->u_change, PER_CHUNK * 4)) != hipSuccess) return fail_free(e);
    if ((e = hipMalloc((void**)&p->u_praw, PER_CHUNK * 4)) != hipSuccess) return fail_free(e);
    if ((e = hipMemsetAsync(p->tree, 0, (2 * capacity - 1) * 4, stream)) != hipSuccess) return fail_free(e);
    *out = p;
    return BDR_OK;
}

// n <= PER_CHUNK updates in batch order.  ixs == nullptr: consecutive rows ix0.. (push)
static int32_t per_apply(bdr_per* p, int n, const uint64_t* ixs, uint64_t ix0, const float* p_raw, hipStream_t st)
{
    PrepArgs a{p->tree, p->capacity, ixs, ix0, p_raw, p->alpha, p->eps, n, p->u_ix, p->u_p, p->u_change, p->mm};
    p->mm_valid = false;
    hipLaunchKernelGGL(k_per_prepare, dim3(1), dim3(256), 0, st, a);
    BDR_HIP(hipGetLastError());
    PropArgs b{p->tree, p->capacity, p->u_ix, p->u_change, n};
    hipLaunchKernelGGL(k_per_propagate, dim3((unsigned)p->maxdepth), dim3(256), 0, st, b);
    BDR_HIP(hipGetLastError());
    return BDR_OK;
}

static int32_t per_minmax(bdr_per* p, hipStream_t st)
{
    hipLaunchKernelGGL(k_per_minmax, dim3((unsigned)((p->capacity + 2047) / 2048)), dim3(256), 0, st, p->tree + (p->capacity - 1), p->n_samples, p->capacity, p->mm);
    BDR_HIP(hipGetLastError());
    return BDR_OK;
}

// set_priority (base.rs:227-235): `len` rows starting at ring position i0 get the current maximum priority.
// SumTree::max() is taken ONCE before the loop (:229), so it is materialised first; the rows are then added
// in order (chunks of PER_CHUNK keep that order).
int32_t per_push(bdr_per* p, uint64_t i0, uint64_t len, hipStream_t st)
{
    if (!p->mm_valid) BDR_TRY(per_minmax(p, st));
    hipLaunchKernelGGL(k_per_fill_max, dim3(1), dim3(256), 0, st, p->u_praw, p->mm, p->alpha, PER_CHUNK);
    p->mm_valid = false;
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
    if (p->normalize == BDR_PER_NORMALIZE_ALL && !p->mm_valid) BDR_TRY(per_minmax(p, st));   // `All` needs the current minimum
    p->mm_valid = false;   // k_per_sample consumes and re-arms it
    a.tree = p->tree; a.mm = p->mm; a.capacity = p->capacity; a.n_samples = p->n_samples;
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
    if (p->normalize == BDR_PER_NORMALIZE_ALL) {   // the next batch()'s minimum, computed here: off the sampling path
        BDR_TRY(per_minmax(p, st));
        p->mm_valid = true;
    }
    p->n_opts += 1;
    return BDR_OK;
}

const float* per_weights(const bdr_per* p) { return p->w; }

int32_t per_read(const bdr_per* p, int32_t what, float* out, uint64_t n, hipStream_t st)
{
    if (what == 0) {           // the sum tree in the reference's layout
        BDR_REQUIRE(n <= 2 * p->capacity - 1, "the sum tree has %llu nodes", (unsigned long long)(2 * p->capacity - 1));
        BDR_HIP(hipMemcpyAsync(out, p->tree, n * 4, hipMemcpyDeviceToHost, st));
        BDR_HIP(hipStreamSynchronize(st));
        return BDR_OK;
    }
    if (what == 1) {           // {min over [0, n_samples), max over all leaves} in the (p+eps)^alpha domain
        BDR_REQUIRE(n == 2, "min/max is two floats");
        if (!p->mm_valid) BDR_TRY(per_minmax(const_cast<bdr_per*>(p), st));
        const_cast<bdr_per*>(p)->mm_valid = false;
        unsigned h[2];
        BDR_HIP(hipMemcpyAsync(h, p->mm, sizeof h, hipMemcpyDeviceToHost, st));
        const unsigned init[2] = {MM_MIN_INIT, MM_MAX_INIT};   // consumed: re-arm
        BDR_HIP(hipMemcpyAsync(p->mm, init, sizeof init, hipMemcpyHostToDevice, st));
        BDR_HIP(hipStreamSynchronize(st));
        memcpy(&out[0], &h[0], 4); memcpy(&out[1], &h[1], 4);
        return BDR_OK;
    }
    return fail(BDR_ERR_INVALID, "unknown PER array %d", what);
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

int32_t per_check(bdr_per* p)
{
    unsigned flag = 0;
    BDR_HIP(hipMemcpy(&flag, p->mm + 2, 4, hipMemcpyDeviceToHost));
    if (!flag) return BDR_OK;
    BDR_HIP(hipMemset(p->mm + 2, 0, 4));
    const int32_t st = fail(BDR_ERR_INVALID, "SumTree::update: change is NaN (a NaN priority / TD error reached update_priority; the reference panics, "
                                             "sum_tree.rs:101-104); those updates were dropped");
    g_err_deferred = 1;   // a condition of an earlier asynchronous tree update, not of the call that reports it
    return st;
}

void per_info(const bdr_per* p, bdr_per_info* o)
{
    o->n_samples = p->n_samples; o->n_opts = p->n_opts; o->beta = per_beta(p);
}

}  // namespace bdr
