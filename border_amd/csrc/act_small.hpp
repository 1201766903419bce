// Policy::sample on a handful of observations (dqn/base.rs:211-242 is called once per environment step; border-core/src/trainer/
// sampler.rs:99-144): the Nature-CNN's conv2 / conv3 / l1 / l2 for n <= ACT_SMALL_MAX rows.
//
// The training kernels tile for a batch of 256: at n = 1 conv2 is two 64 x 64 tiles walking 16 k-tiles one after the other (12 us),
// l1 streams its 6.4 MB of weights through 64 workgroups (10 us) - 53 us of kernels for 19 MFLOP.  Here a layer is cut the other way:
// a workgroup owns a 32 x 32 output tile AND one of KS slices of the reduction, its four waves a quarter of the slice each, operands
// straight from memory into the FP32 MFMAs (dense.hpp dense_small_tile: no LDS staging, no barrier in the loop); the KS partial tiles
// meet in the LAST workgroup of the tile to finish (ticket; agent-scope stores / loads: the XCDs' L2s are not coherent), which adds
// them in slice order, then bias and ReLU.  conv2 at n = 1: 24 workgroups of 16 MFMAs per wave instead of 2 x 256; l1: 16 column tiles
// x 7 slices = 112 workgroups (49 slices of one round trip each were slower: the last arriver then adds 49 partials).  The workgroup that completes l1's last tile goes on to l2 and stores the Q rows, the device's error
// words and a sequence number into pinned host memory, where the caller waits: five launches, no copy command, no synchronisation.
// Products and the per-slice sums are exact-f32 MFMA chains as in training; the ORDER of the additions differs from the training
// forward (k-tiles there, slices here), so Q agrees with it to ~1e-6 relative, not bit for bit - acting and training never compare bits.
#pragma once
#include "dense.hpp"

namespace bdr {

constexpr int ACT_SMALL_MAX = 8;    // rows (environments) per call served by this path (at 16 the training kernels are as fast: 55 vs 57 us)

struct ActLayerArgs {
    const float* x;          // input activations, NHWC f32: conv2 a1 [n][20][20][32], conv3 a2 [n][9][9][64], l1 a3 [n][3136]
    const float* w;          // [K][N] (k ordered (kh, kw, c) as the input rows are gathered)
    const float* bias;
    float* out;              // [M][N]
    float* part;             // [KS][Mpad][N] partial tiles (KS > 1)
    unsigned* tickets;       // [m-tiles * n-tiles] (+ 1 for the head ticket), zero between launches
    int M, Mpad, N, K, KS, relu;
    const float* had; int had_group;   // GEO 1: the input row m is x[m][k] * had[m / had_group][k] (IQN's merge relu(phi) * psi(x)[b]); both rows K long
    // l1 only (head.n_rows > 0): the workgroup that completes the LAST tile of the layer goes on to l2 and the hand-over to the host
    struct Head {
        const float* w5; const float* b5; float* q; int n_rows, A;
        float* rows_host; unsigned* seq_host; unsigned seq; const unsigned* dev_err; int n_err;
    } head;
};

// GEO: 0 = dense rows, 1 = dense rows times a per-group row (Hadamard), 2 = conv2 patches (4 x 4 x 32, stride 2 on a 20 x 20 map), 3 = conv3 patches (3 x 3 x 64, stride 1 on 9 x 9)
template <int GEO>
static __global__ __launch_bounds__(256) void k_act_layer(ActLayerArgs a)
{
    __shared__ float red[4][32][33];
    __shared__ unsigned s_last;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NT = a.N / 32, tile = blockIdx.x, mt = tile / NT, nt = tile - mt * NT, ks = blockIdx.y;
    const int m0 = mt * 32, n0 = nt * 32, kslice = a.K / a.KS, k0 = ks * kslice;
    const int row = min(m0 + (lane & 31), a.M - 1);   // rows >= M alias the last row (never stored)
    const float* base;
    if constexpr (GEO == 2) { const int b = row / 81, r = row - b * 81, oh = r / 9, ow = r - oh * 9; base = a.x + ((size_t)(b * 20 + 2 * oh) * 20 + 2 * ow) * 32; }
    else if constexpr (GEO == 3) { const int b = row / 49, r = row - b * 49, oh = r / 7, ow = r - oh * 7; base = a.x + ((size_t)(b * 9 + oh) * 9 + ow) * 64; }
    else base = a.x + (size_t)row * a.K;
    const float* hrow = GEO == 1 ? a.had + (size_t)(row / a.had_group) * a.K : nullptr;
    auto loadA = [&](int k) -> f32x4 {   // the lane's four input values k0 + k ... + 3 of its row (k % 4 == 0: inside one channel run)
        const int kk = k0 + k;
        if constexpr (GEO == 2) { const int seg = kk >> 5, kh = seg >> 2, kw = seg & 3; return *reinterpret_cast<const f32x4*>(base + (kh * 20 + kw) * 32 + (kk & 31)); }
        else if constexpr (GEO == 3) { const int seg = kk >> 6, kh = seg / 3, kw = seg - kh * 3; return *reinterpret_cast<const f32x4*>(base + (kh * 9 + kw) * 64 + (kk & 63)); }
        else if constexpr (GEO == 1) return *reinterpret_cast<const f32x4*>(base + kk) * *reinterpret_cast<const f32x4*>(hrow + kk);
        else return *reinterpret_cast<const f32x4*>(base + kk);
    };
    // the epilogue's operand is requested beside the tile operands
    const int r = tid >> 3, c4 = (tid & 7) * 4, m = m0 + r, n = n0 + c4;
    const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bias + n);
    dense_small_tile<false>(loadA, a.w + (size_t)k0 * a.N, a.N, n0, kslice, wave, lane, red);
    __syncthreads();
    f32x4 v;
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = dense_small_sum(red, r, c4 + q);
    if (a.KS > 1) {
        if (m < a.M) {   // (rows beyond M are nobody's)
            float* p = a.part + ((size_t)ks * a.Mpad + m) * a.N + n;
#pragma unroll
            for (int q = 0; q < 4; ++q) st_agent(p + q, v[q]);
        }
        if (!last_workgroup(a.tickets + tile, (unsigned)a.KS, &s_last)) return;
        // the tile's KS partials, in slice order whoever arrives last
        if (m < a.M) {
            // (relaxed agent-scope atomic loads are issued one round trip at a time: 4 KS of them in a row were most of the kernel.  Eight
            // slices' 16-byte loads go out together - sc1: served by the coherent level - and are waited for once)
            const float* p0 = a.part + (size_t)m * a.N + n;
            const size_t sstride = (size_t)a.Mpad * a.N;
            v = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int s0 = 0; s0 < a.KS; s0 += 8) {
                f32x4 t[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float* ps = p0 + (size_t)min(s0 + e, a.KS - 1) * sstride;
                    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(t[e]) : "v"(ps) : "memory");
                }
                // (the wait carries the eight destinations as in/out operands: without that the compiler schedules their first uses in
                // front of it - to the compiler an asm's output is there when the statement is)
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[4]), "+v"(t[5]), "+v"(t[6]), "+v"(t[7]) :: "memory");
#pragma unroll
                for (int e = 0; e < 8; ++e) if (s0 + e < a.KS) v += t[e];
            }
        }
    }
    if (m < a.M) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { v[q] += bv[q]; if (a.relu) v[q] = v[q] > 0.f ? v[q] : 0.f; }
        if (GEO == 0 && a.head.n_rows > 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) st_agent(a.out + (size_t)m * a.N + n + q, v[q]);   // read by the head workgroup below, possibly on another XCD
        } else {
            *reinterpret_cast<f32x4*>(a.out + (size_t)m * a.N + n) = v;
        }
    }
    if constexpr (GEO == 0) {
        if (a.head.n_rows <= 0) return;
        // ---- l2 + hand-over, by the workgroup that completes the layer's last tile (second ticket, behind this tile's output stores):
        // Q[row][act] = h1[row] . W5[act] + b5[act], a wave per row, 8 consecutive values of the row per lane held while the actions go by
        const int n_tiles = (a.Mpad / 32) * NT;
        if (!last_workgroup(a.tickets + n_tiles, (unsigned)n_tiles, &s_last)) return;
        const ActLayerArgs::Head& hd = a.head;
        for (int rw = wave; rw < hd.n_rows; rw += 4) {
            // the row's 512 values, 8 consecutive ones per lane: two 16-byte loads served by the coherent level (the tiles' last arrivers
            // stored them with agent scope, possibly on other XCDs), both in flight before the wait
            f32x4 h0, h1v;
            {
                const float* ph = a.out + (size_t)rw * 512 + 8 * lane;
                asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(h0) : "v"(ph) : "memory");
                asm volatile("global_load_dwordx4 %0, %1, off offset:16 sc1" : "=&v"(h1v) : "v"(ph) : "memory");
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(h0), "+v"(h1v) :: "memory");
            }
            for (int a0 = 0; a0 < hd.A; a0 += 8) {   // eight actions at a time: their weight loads are in flight together
                float sacc[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float* w = hd.w5 + (size_t)min(a0 + e, hd.A - 1) * 512 + 8 * lane;
                    const f32x4 w0 = *reinterpret_cast<const f32x4*>(w), w1 = *reinterpret_cast<const f32x4*>(w + 4);
                    float t = 0.f;
#pragma unroll
                    for (int j = 0; j < 4; ++j) t = fmaf(h0[j], w0[j], t);
#pragma unroll
                    for (int j = 0; j < 4; ++j) t = fmaf(h1v[j], w1[j], t);
                    sacc[e] = t;
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) {
#pragma unroll
                    for (int d = 32; d > 0; d >>= 1) sacc[e] += __shfl_xor(sacc[e], d, 64);
                }
                if (lane < 8 && a0 + lane < hd.A) {
                    float mine = sacc[0];
#pragma unroll
                    for (int e = 1; e < 8; ++e) mine = lane == e ? sacc[e] : mine;
                    const float qv = mine + hd.b5[a0 + lane];
                    hd.q[rw * hd.A + a0 + lane] = qv; hd.rows_host[rw * hd.A + a0 + lane] = qv;
                }
            }
        }
        if (hd.dev_err && tid < hd.n_err) hd.seq_host[4 + tid] = hd.dev_err[tid];
        __threadfence_system();
        __syncthreads();
        if (tid == 0) __hip_atomic_store(hd.seq_host, hd.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

template <int GEO>
inline hipError_t launch_act_layer(hipStream_t st, const ActLayerArgs& a)
{
    hipLaunchKernelGGL(k_act_layer<GEO>, dim3((a.Mpad / 32) * (a.N / 32), a.KS), dim3(256), 0, st, a);
    return hipGetLastError();
}

// scratch of the path: partial tiles of the widest layer at ACT_SMALL_MAX rows, one ticket per output tile
constexpr size_t act_small_part_floats()
{
    const size_t c2 = (size_t)4 * (((size_t)81 * ACT_SMALL_MAX + 31) / 32 * 32) * 64, c3 = (size_t)3 * (((size_t)49 * ACT_SMALL_MAX + 31) / 32 * 32) * 64,
                 l1 = (size_t)7 * 32 * 512;
    return c2 > c3 ? (c2 > l1 ? c2 : l1) : (c3 > l1 ? c3 : l1);
}
constexpr size_t ACT_SMALL_TICKETS = 128;

}  // namespace bdr
