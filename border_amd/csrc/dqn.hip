// DQN agent on MI355X: Dqn::update_critic / opt_ (border-tch-agent/src/dqn/base.rs:60-200) for the
// AtariCnn Q-network (border-tch-agent/src/cnn/base.rs:23-36), as hand-written gfx950 kernels.
//
// One opt step enqueues, on the agent's stream (no host sync; dqn/base.rs:154-157's per-step loss
// read-back happens only in opt_with_record):
//   replay: ChaCha12 indices + row gather                       (replay.hip)
//   fwd   : conv1..conv3, l1 (split-K) for {online(obs), target(next_obs)[, online(next_obs)]}
//           in ONE launch per layer (blockIdx.z = network instance), FP32 MFMA implicit GEMM
//           (conv1 on the bf16 matrix cores with exact operands)
//   head  : l1 finish (+bias, relu) + l2 (wave dot products) -> Q-values, and in the same launch the TD step:
//           gather Q(s,a), max/argmax target, r + (1-term)*gamma*q', Huber/MSE (PER: importance-weighted), dL/dQ, dL/dh1
//   bwd   : l2 grads, l1 dW/dX, conv3 dW/dX, conv2 dW/dX (FP32 MFMA), conv1 dW (bf16, exact operands); PER: tree update
//   adam  : fused p,g,m,v pass over the flat parameter arena (libtorch Adam::step formula)
//   track : tau*src + (1-tau)*dst every soft_update_interval opts (util.rs:31-45)
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <string>

#include "common.hpp"
#include "agent_base.hpp"
#include "igemm.hpp"
#include "conv1_bf16_img.hpp"
#include "cnn_layers.hpp"
#include "igemm_b3.hpp"
#include "act_small.hpp"

using namespace bdr;

namespace {

// 4-wave teams per workgroup of each implicit-GEMM launch (see k_igemm): 2 where the grid gives ~1 workgroup per CU
#ifndef BDR_TEAMS
#define BDR_TEAMS 2
#endif
// measured on MI355X when introduced: conv3 fwd 27.1->25.4us, l1 dX 22.7->17.1us, conv3 dX 32.2->31.3us with 2 teams; the
// other launches already have >= 2.5 workgroups per CU and lose (conv2 dX 37.9->49.6us).  With the current pipeline the
// per-kernel difference is within noise, but the 8-wave form still co-runs better with the weight-gradient stream
// (A/B in the pipeline: 3 760 vs 3 695 opt-steps/s for conv3 dX).
#ifndef BDR_TEAMS_FWD_C3
#define BDR_TEAMS_FWD_C3 BDR_TEAMS
#endif
#ifndef BDR_TEAMS_DX_L1
#define BDR_TEAMS_DX_L1 BDR_TEAMS
#endif
#ifndef BDR_TEAMS_DX_C3
#define BDR_TEAMS_DX_C3 BDR_TEAMS   // round 6, with conv2's dX on position-class tiles: one team is +1.1 % IN THE STEP (5 218-5 222 against 5 158-5 163 opt-steps/s, same box,
                                   // interleaved: fewer waves leave the other queue more room) but 23.4 us instead of 18.1 on its own (18 k-tiles in sequence at the
                                   // interior positions); two teams stay: the kernel's own time is what the roofline is read on (LAB.md "Round 6")
#endif
#ifndef BDR_TEAMS_FWD_L1
#define BDR_TEAMS_FWD_L1 1
#endif
#ifndef BDR_TEAMS_FWD_C2
#define BDR_TEAMS_FWD_C2 1
#endif
#ifndef BDR_TEAMS_DX_C2
#define BDR_TEAMS_DX_C2 1
#endif
#ifndef BDR_DXC2_MERGED
#define BDR_DXC2_MERGED 2   // conv2 input gradient: 2 = the four parity classes as one GEMM over position-class tiles (DxC2MPos, valid taps only); 1 = the same GEMM over flat row tiles (DxC2M); 0 = one launch slice per class (DxC2)
#endif
// conv3 forward of the DQN step: 32x64 tiles, two k-split teams (392 workgroups per instance instead of 196: with the split forward every
// launch is ONE instance, and 196 tiles left a quarter of the CUs idle; round 4, same box: 4 700 vs 4 647 opt-steps/s)
#ifndef BDR_FWDC3_SHAPE
#define BDR_FWDC3_SHAPE 1, 2, 1, 1
#endif
using FwdC3D = FwdConvP<GeomC3, BDR_FWDC3_SHAPE>;
constexpr size_t PL2_U16 = (size_t)3 * 64 * 512, PL3_U16 = (size_t)3 * 64 * 576;   // W2's / W3's three planes
// One parameter set's planes: W2 then W3, each [3 planes][K / 32][64 cout][32 k]: the B tile of k-tile kt (64 columns x 32 k of one plane) is
// 4 KB contiguous, so a wave of staging threads (16 columns x four 16-byte chunks) fetches 1 KB in one piece.  Measured against [cout][K] rows
// (+0.8 % on the step) and [K / 8][cout][8] (-1.8 %); k_reduce_adam's 32 consecutive cout of one k land 64 bytes apart.
constexpr size_t CPL_W2 = 0, CPL_W3 = PL2_U16, CPL_U16 = PL2_U16 + PL3_U16;
__device__ __forceinline__ size_t cpl_index(int k, int n) { return ((size_t)(k >> 5) * 64 + n) * 32 + (k & 31); }

// element e = k * 64 + n of W2 (layer 0) or W3 (layer 1) -> its three bf16 terms
__device__ __forceinline__ void conv_plane_store(uint16_t* __restrict__ pl, int layer, int e, float x)
{
    uint16_t v[3];
    split3_rn(x, v);                                            // (igemm_b3.hpp: round-to-nearest terms, exact sum)
    const size_t n = (size_t)(layer ? 576 : 512) * 64, o = cpl_index(e >> 6, e & 63);
    uint16_t* d = pl + (layer ? CPL_W3 : CPL_W2);
    d[o] = v[0]; d[n + o] = v[1]; d[2 * n + o] = v[2];
}

// the planes of one parameter set from its f32 weights (every writer of conv parameters other than k_reduce_adam leaves them stale:
// DqnCnn::cpl_fresh; the forward re-splits before it reads them)
__global__ __launch_bounds__(256) void k_conv_planes(const float* __restrict__ w2, const float* __restrict__ w3, uint16_t* __restrict__ pl)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < 512 * 64) conv_plane_store(pl, 0, i, w2[i]);
    else if (i < 512 * 64 + 576 * 64) conv_plane_store(pl, 1, i - 512 * 64, w3[i - 512 * 64]);
}
// conv2 / conv3 forward on the bf16 matrix cores with split operands (igemm_b3.hpp, six of the nine partial products): the A rows are the
// f32 activations, split on their way into LDS; the weights are read from bf16 planes kept beside each parameter set (cpl_index)
#ifndef BDR_FWDC2_B3_SHAPE
#define BDR_FWDC2_B3_SHAPE 2, 2
#endif
#ifndef BDR_FWDC3_B3_SHAPE
#define BDR_FWDC3_B3_SHAPE 2, 2
#endif
struct FwdB3Args : FwdArgs { const uint16_t* wpl[MAXZ]; };
template <class G, int WM_, int WN_, int TM_ = 1, int TN_ = 1>
struct FwdB3 : FwdP<G, AFwd<G>, WM_, WN_, false, 0, TM_, TN_> {
    using Args = FwdB3Args;
    __device__ static const uint4* b_chunk(const Args& a, int z, int, int pl, int kt, int n, int kq)
    {
        return reinterpret_cast<const uint4*>(a.wpl[z] + (size_t)pl * G::COUT * G::K + cpl_index(kt * 32 + kq * 8, n));
    }
};
using FwdC2B3 = FwdB3<GeomC2, BDR_FWDC2_B3_SHAPE>;
using FwdC3B3 = FwdB3<GeomC3, BDR_FWDC3_B3_SHAPE>;
constexpr int TEAMS_FWD_C2 = BDR_TEAMS_FWD_C2, TEAMS_FWD_C3 = BDR_TEAMS_FWD_C3, TEAMS_FWD_L1 = BDR_TEAMS_FWD_L1, TEAMS_DX_L1 = BDR_TEAMS_DX_L1,
              TEAMS_DX_C3 = BDR_TEAMS_DX_C3, TEAMS_DX_C2 = BDR_TEAMS_DX_C2;

// ================================================================================================
// head: l1 finish + l2, one wave per (row, instance)
// ================================================================================================
struct HeadArgs {
    const float* p1[MAXZ];   // l1 split-K partials [S][B][512]
    const float* b4[MAXZ];
    const float* w5[MAXZ];   // [A][512]
    const float* b5[MAXZ];
    float* h1[MAXZ];         // [B][512] post-relu
    float* q[MAXZ];          // [B][A]
    int B, A, S;
};

// TD step of a row, executed by k_head right after the row's Q-values: instance 0 = qnet(obs), 1 = qnet_tgt(next_obs),
// 2 = qnet(next_obs) when double_dqn (dqn/base.rs:91-103); the Q rows and h1 never leave the workgroup.
struct TdArgs {
    int double_dqn;
    const uint8_t* act;    // [B] rows of act_bytes; first 8 bytes = i64 action
    int act_bytes;
    const float* reward;
    const int8_t* term;
    float* dh1;            // [B][512]
    float* dq;             // [B]   dL/dQ(s,a) (already divided by B)
    float* pred;           // [B]
    float* tgt;            // [B]
    float* loss_row;       // [B]
    int B, A;
    float gamma;
    int loss_kind;         // 0 mse, 1 smooth-l1
    const float* weight;   // [B] importance weights (PER) or nullptr
    float* td_abs;         // [B] |pred - tgt| (clipped) for update_priority, or nullptr
    int has_clip; float clip_min, clip_max;
    unsigned* err;         // bdr_agent::dev_err (ERR_ACTION is raised for an action outside [0, A))
    // Split forward (schedule 3): the target network's Q rows were produced by a separate launch on the other queue.
    //   q_tgt_mem   [B][A] target Q rows in memory (nullptr: the target instance is z_tgt of THIS launch, rows in LDS)
    //   z_sel       LDS instance whose argmax selects the action (double DQN: online(next_obs)); -1: the target rows themselves
    //   wait_sig    flag that must reach wait_epoch before q_tgt_mem is read (published by the other queue once the target
    //               head kernel is complete); polled inside this kernel, so the dX queue carries no extra gate packet
    const float* q_tgt_mem;
    int z_tgt, z_sel;
    unsigned* wait_sig; unsigned wait_epoch; unsigned long long wait_limit; unsigned* sig_err;
};

// One workgroup = HEAD_ROWS batch rows x nz network instances, one wave per (row, instance):
//   l1 finish (split-K partials summed in split order) + bias + ReLU -> h1, then l2 -> Q(row, .).
// With td != 0 the TD step of the row follows in the same launch (dqn/base.rs:71-74 pred, :91-105 target,
// :123-152 loss): the instance-0 wave of the row picks up the other instances' Q rows from LDS and still
// holds its own h1 row in registers for dL/dh1, so neither Q nor h1 is read back from memory.
// Latency shape: EVERYTHING the wave will need (the 2*S partial vectors, its 8 x A block of W5, the row's action /
// reward / flags) is requested up front, so the kernel pays one memory round trip; AMAX is the compile-time bound
// of the register-resident W5 block (A <= AMAX).
constexpr int HEAD_ROWS = 2;
template <int S, int AMAX>
__global__ __launch_bounds__(64 * HEAD_ROWS * MAXZ) void k_head(HeadArgs a, TdArgs t, int nz, int td)
{
    __shared__ float sq[HEAD_ROWS][MAXZ][64];
    // (scalar wave index: the instance z selects kernel-argument pointers - a.p1[z], a.h1[z], a.qv[z] ... - which must be scalar loads;
    //  with a per-lane z the compiler fetched them with vector loads, one memory round trip in front of the data loads and one
    //  `vmcnt(0)` in front of each store block)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = wave / nz, z = wave % nz;
    const int row = blockIdx.x * HEAD_ROWS + r;
    const bool valid = row < a.B;
    const int rw = valid ? row : 0;
    const int A = a.A;
    // lane owns columns [lane*8, lane*8+8)
    f32x4 part[S][2];
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const float* p = a.p1[z] + ((size_t)s * a.B + rw) * 512 + lane * 8;
        part[s][0] = *reinterpret_cast<const f32x4*>(p);
        part[s][1] = *reinterpret_cast<const f32x4*>(p + 4);
    }
    const f32x4 b4lo = *reinterpret_cast<const f32x4*>(a.b4[z] + lane * 8), b4hi = *reinterpret_cast<const f32x4*>(a.b4[z] + lane * 8 + 4);
    // W5 is [A][512] (the reference's own [out][in]): action k's weights for this lane's 8 columns are two f32x4
    const float* w5 = a.w5[z] + lane * 8;
    float wreg[8][AMAX];
#pragma unroll
    for (int k = 0; k < AMAX; ++k) {
        const float* wk = w5 + (size_t)min(k, A - 1) * 512;
        const f32x4 lo = *reinterpret_cast<const f32x4*>(wk), hi = *reinterpret_cast<const f32x4*>(wk + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { wreg[j][k] = lo[j]; wreg[4 + j][k] = hi[j]; }
    }
    const float b5 = lane < A ? a.b5[z][lane] : 0.f;
    long long act = 0; float reward = 0.f, wgt = 1.f; int term = 0;
    if (td && z == 0) {
        act = *reinterpret_cast<const long long*>(t.act + (size_t)rw * t.act_bytes);
        if (act < 0 || act >= A) {   // the reference's gather raises; here: flag for the host, clamp to stay in bounds
            if (lane == 0 && t.err) atomicOr(t.err + bdr_agent::ERR_ACTION, 1u);
            act = act < 0 ? 0 : A - 1;
        }
        reward = t.reward[rw]; term = (int)t.term[rw];
        if (t.weight) wgt = t.weight[rw];
    }

    f32x4 h[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
#pragma unroll
    for (int s = 0; s < S; ++s) { h[0] += part[s][0]; h[1] += part[s][1]; }
    float hv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float v = h[j >> 2][j & 3] + (j < 4 ? b4lo[j & 3] : b4hi[j & 3]);
        hv[j] = v > 0.f ? v : 0.f;   // cnn/base.rs:34 relu
    }
    // the TD step's row scalars land in their registers here, before the kernel's first store: stores share vmcnt with loads on gfx9 and may
    // be acknowledged out of order, so a load result first used behind a store costs an s_waitcnt vmcnt(0) - the store's round trip too
    asm volatile("" ::"v"(act), "v"(reward), "v"(term), "v"(wgt), "v"(b5));
    if (valid) {
        float* ho = a.h1[z] + (size_t)row * 512 + lane * 8;
        *reinterpret_cast<f32x4*>(ho) = f32x4{hv[0], hv[1], hv[2], hv[3]};
        *reinterpret_cast<f32x4*>(ho + 4) = f32x4{hv[4], hv[5], hv[6], hv[7]};
    }
    // l2: one dot product per action over the 512 columns; the AMAX butterflies interleave
    float sk[AMAX];
#pragma unroll
    for (int k = 0; k < AMAX; ++k) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) s = fmaf(hv[j], wreg[j][k], s);
        sk[k] = s;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
        for (int k = 0; k < AMAX; ++k) sk[k] += __shfl_xor(sk[k], off);
    // every lane holds all sums: lane k keeps action k
    float qv = 0.f;
#pragma unroll
    for (int k = 0; k < AMAX; ++k) qv = lane == k ? sk[k] : qv;
    qv += b5;
    if (lane < A) {
        if (valid) a.q[z][(size_t)row * A + lane] = qv;
        sq[r][z][lane] = qv;
    }
    if (!td) return;
    if (t.q_tgt_mem && t.wait_sig) {
        // the target rows come from the other queue: wait for its "target head complete" flag (one poller per workgroup; the
        // wait overlaps nothing critical - this workgroup's own l1 finish / l2 are done), then acquire at agent scope so the
        // rows are re-read from memory (another XCD's L2 wrote them)
        if (threadIdx.x == 0) {
            const unsigned long long t0 = wall_clock64();
            while ((int)(__hip_atomic_load(t.wait_sig, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - t.wait_epoch) < 0) {
                __builtin_amdgcn_s_sleep(2);
                if (__hip_atomic_load(t.sig_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;   // poisoned
                if (wall_clock64() - t0 > t.wait_limit) {
                    __hip_atomic_store(t.sig_err, 100u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (t.err) __hip_atomic_store(t.err + bdr_agent::ERR_GATE, 100u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
        }
    }
    __syncthreads();
    if (z != 0 || !valid) return;
    // ---- TD of this row (instance 0 = qnet(obs); target rows = qnet_tgt(next_obs); selection rows = qnet(next_obs) for double DQN)
    if (t.q_tgt_mem) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (lane < A) sq[r][MAXZ - 1][lane] = __builtin_nontemporal_load(t.q_tgt_mem + (size_t)row * A + lane);   // (instance slot MAXZ-1 is free: nz <= 2 here)
    }
    const float* q_tg = t.q_tgt_mem ? sq[r][MAXZ - 1] : sq[r][t.z_tgt];
    const float* sel = t.z_sel >= 0 ? sq[r][t.z_sel] : q_tg;
    // argmax over actions: first maximal index (at::argmax), lanes >= A hold -inf
    float v = lane < A ? sel[lane] : -INFINITY;
    int idx = lane;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(v, off);
        const int oi = __shfl_xor(idx, off);
        if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
    }
    const float qn = q_tg[idx];
    const float pred = sq[r][0][act];
    // reward + (1 - is_terminated) * discount_factor * q   (dqn/base.rs:104)
    const float tgt = td_target(reward, (float)(1 - term), t.gamma, qn);
    const TdLossIn li{t.loss_kind, t.weight != nullptr, wgt, t.has_clip, t.clip_min, t.clip_max};
    float lossb, tdv;
    const float dl = td_loss_row(pred, tgt, li, lossb, tdv);
    const float dq = dl / (float)a.B;   // Reduction::Mean
    if (lane == 0) {
        t.dq[row] = dq; t.pred[row] = pred; t.tgt[row] = tgt; t.loss_row[row] = lossb;
        if (t.td_abs) t.td_abs[row] = tdv;
    }
    // dL/dh1[row][j] = relu'(h1) * dq * W5[j][act]  (column `act` of the register block)
    float out[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float wj = 0.f;
#pragma unroll
        for (int k = 0; k < AMAX; ++k) wj = (int)act == k ? wreg[j][k] : wj;
        out[j] = hv[j] > 0.f ? dq * wj : 0.f;
    }
    float* o = t.dh1 + (size_t)row * 512 + lane * 8;
    *reinterpret_cast<f32x4*>(o) = f32x4{out[0], out[1], out[2], out[3]};
    *reinterpret_cast<f32x4*>(o + 4) = f32x4{out[4], out[5], out[6], out[7]};
}

// l2 gradients + loss mean.  Workgroup (a, half): gW5[j][a] = sum_{b: act[b]==a} h1[b][j] * dq[b] for 256 j.
// Matching rows are compacted (in batch order) into LDS first so the row loop carries no branch and
// its loads pipeline; the extra workgroup (blockIdx.x == 2*A) reduces the per-row losses to the mean.
struct HeadBwdArgs {
    const float* h1; const float* dq; const uint8_t* act; int act_bytes;
    const float* loss_row;
    float* gw5; float* gb5; float* loss;   // loss: [1]
    int B, A;
};
__global__ __launch_bounds__(256) void k_head_bwd(HeadBwdArgs a)
{
    __shared__ int s_rows[256];
    __shared__ float s_dq[256];
    __shared__ int s_wcnt[4];
    __shared__ float s_red[256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if ((int)blockIdx.x == 2 * a.A) {   // loss = mean over rows, fixed-order tree
        float s = 0.f;
        for (int b = tid; b < a.B; b += 256) s += a.loss_row[b];
        s_red[tid] = s;
        __syncthreads();
        for (int w = 128; w > 0; w >>= 1) {
            if (tid < w) s_red[tid] += s_red[tid + w];
            __syncthreads();
        }
        if (tid == 0) a.loss[0] = s_red[0] / (float)a.B;
        return;
    }
    const int ac = blockIdx.x >> 1, j = (blockIdx.x & 1) * 256 + tid;
    float acc = 0.f, bacc = 0.f;
    for (int b0 = 0; b0 < a.B; b0 += 256) {
        const int b = b0 + tid;
        bool match = false;
        float dqv = 0.f;
        if (b < a.B) {   // (dq beside the action: behind the match test it was one more dependent round trip)
            match = *reinterpret_cast<const long long*>(a.act + (size_t)b * a.act_bytes) == ac;
            dqv = a.dq[b];
        }
        const unsigned long long m = __ballot(match);
        if (lane == 0) s_wcnt[wave] = __popcll(m);
        __syncthreads();
        int base = 0;
        for (int w = 0; w < wave; ++w) base += s_wcnt[w];
        const int total = s_wcnt[0] + s_wcnt[1] + s_wcnt[2] + s_wcnt[3], padded = (total + 15) & ~15;
        if (match) {
            const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
            s_rows[pos] = b; s_dq[pos] = dqv;
        }
        if (tid >= total && tid < padded) { s_rows[tid] = b0; s_dq[tid] = 0.f; }   // pad to whole batches of 16: a valid address, weight 0, value replaced by 0 below
        __syncthreads();
        // 16 row loads in flight per thread (each is an L2 round trip); the padded tail replaces runs of 4 and single loads - up to five
        // more dependent round trips at 43 rows per action.  The bias sum (one thread, LDS only) rides under the first batch.
        for (int k = 0; k < padded; k += 16) {
            float hvv[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) hvv[u] = a.h1[(size_t)s_rows[k + u] * 512 + j];
            if (k == 0 && tid == 0) for (int q = 0; q < total; ++q) bacc += s_dq[q];
#pragma unroll
            for (int u = 0; u < 16; ++u) acc = fmaf(k + u < total ? hvv[u] : 0.f, s_dq[k + u], acc);   // (a padded slot is 0 x 0, not 0 x h1[b0]: an Inf / NaN activation of row b0 must not reach other actions' gradients)
        }
        __syncthreads();
    }
    a.gw5[(size_t)ac * 512 + j] = acc;
    if (tid == 0 && (blockIdx.x & 1) == 0) a.gb5[ac] = bacc;
}

}  // namespace

// ================================================================================================
// agent handle: Dqn<E, AtariCnn, R>
// ================================================================================================
struct DqnCnn : bdr_agent {
    bdr_dqn_config cfg;
    hipStream_t side = nullptr;          // weight-gradient kernels run here, concurrently with the dX chain
    hipStream_t aux = nullptr;           // prioritized-replay tree updates
    hipEvent_t ev_fork[4] = {nullptr}, ev_join = nullptr;
    bool kev = true;
    unsigned* sig = nullptr;   // [8] device progress flags of schedule 3
    unsigned sig_epoch = 0;
    unsigned test_epoch = 0;
    bool aux_gated = true;     // the PER queue may wait through gates (queues_independent at its creation)
    bool head_gate_enqueued = false;
    bool side_gather = true;   // BDR_NO_SIDE_GATHER=1: opt() gathers on the dX queue
    unsigned long long* gate_trace = nullptr;   // BDR_GATE_TRACE=1: [5][2] (100 MHz ticks waited, count), [20] join lag, [22 + site] longest wait; printed at destruction
    int sched = 3;   // backward schedule, see update_critic (BDR_SCHED=0|1|2; BDR_NO_OVERLAP=1 == 0)
    unsigned long long gate_limit = 1000000000ull;   // gate time limit in 100 MHz ticks (10 s; BDR_GATE_LIMIT_MS for tests)
    bool holds_gate_token = false;                   // see claim_gates()
    bool defer_adam = false;                         // update_critic stops after backward (synchronous-DP mode, grads_on_batch)
    bool split_fwd = true;                           // schedule 3: target network forward on the other queue (BDR_NO_SPLIT_FWD=1: off)
    bool noted_event_fallback = false;               // the one-line note of effective_sched() has been printed
    bool three_queues = true;                        // gather + target forward on the third stream when it is free (no prioritized replay); BDR_TQ=0: on the weight-gradient queue
    bool tgt_enqueued = false;                       // opt_inner has put this update's target forward on the other queue
    unsigned track_epoch = 0;                        // epoch of the last update that was followed by a soft update (SIG_TRACK)
    bool track_wait_pending = false;                 // the other queue must see SIG_TRACK == track_epoch before it reads q_tgt again
    // overlapped parameter exchange (exchange_plan): communication queue, and what the next update has to wait for
    hipStream_t comm_st = nullptr;
    int comm_state = 0;                              // 0 untested, 1 usable (own hardware queue), -1 not usable: exchange in-stream
    unsigned xchg_epoch = 0;                         // epoch (sig_epoch of the exchanged step) of the pending exchange
    bool xchg_pending_conv = false, xchg_pending_fc = false, xchg_tracked = false;
    Arena ar;
    int B = 0;          // activation buffers are sized for this batch
    // parameter arenas
    float *q = nullptr, *q_tgt = nullptr, *grad = nullptr, *m = nullptr, *v = nullptr;
    float* vmax = nullptr; bool amsgrad = false;   // AdamW{amsgrad: true}: max_exp_avg_sq arena (5); the step then always runs backward -> adam_all (the split path of synchronous DP)
    // activations [instance]
    float* a1[MAXZ] = {nullptr}; float* a2[MAXZ] = {nullptr}; float* a3[MAXZ] = {nullptr};
    float* p1[MAXZ] = {nullptr}; float* h1[MAXZ] = {nullptr}; float* qv[MAXZ] = {nullptr};
    // backward
    float *dh1 = nullptr, *dy3 = nullptr, *dy2 = nullptr, *dy1 = nullptr;
    float *dq = nullptr, *pred = nullptr, *tgt = nullptr, *loss_row = nullptr, *loss = nullptr;
    float *part = nullptr; size_t part_floats = 0;
    // update_on_batch staging
    uint8_t *u_obs = nullptr, *u_next = nullptr, *u_act = nullptr; float* u_rew = nullptr; int8_t* u_term = nullptr;
    uint64_t u_cap = 0;
    const float* last_reward = nullptr; int last_B = 0;
    // bookkeeping (dqn/base.rs:26-48)
    uint64_t adam_step = 0, soft_update_counter = 0;
    int64_t conv_step_lag = 0;   // the conv segment's optimizer step number is adam_step - conv_step_lag: 0 unless a gate time-out fell between the step's two optimizer passes (on_gate_timeout)
    // conv2 / conv3 on the bf16 matrix cores with split f32 operands (igemm_b3.hpp): bf16 planes of W2 / W3 beside each parameter set
    // ([0] online, [1] target).  k_reduce_adam writes the online set's planes with the parameters; every other writer of conv
    // parameters marks the set stale (planes_stale) and the next forward re-splits it on its own queue, behind whatever orders
    // it after that writer.  A raw arena pointer handed out (bdr_agent_arena_device_ptr) can be written at any time: that set is
    // then re-split before every forward.  BDR_DQN_F32_EXACT=1: every layer on the FP32 MFMA (exact products), no planes.
    uint16_t* cpl[2] = {nullptr, nullptr};
    bool cpl_fresh[2] = {false, false}, cpl_escaped[2] = {false, false};
    bool conv_b3 = true;
    void planes_stale(int set) { if (set == 0 || set == 1) cpl_fresh[set] = false; }
    void arena_escaped(int which) override { if (which == 0 || which == 1) cpl_escaped[which] = true; }
    void arena_released(int which) override { if (which == 0 || which == 1) { cpl_escaped[which] = false; cpl_fresh[which] = false; } }
    float* act_part = nullptr; unsigned* act_tickets = nullptr;   // scratch of the acting kernels (act_small.hpp)
    unsigned long long* applied_step = nullptr;   // device words: the Adam step number of the last pass that was not skipped - [0] l1 / l2 (k_adam, or the whole arena), [1] the conv segment (k_reduce_adam)

    ~DqnCnn() override;
    const char* kind() const override { return "dqn_cnn"; }
    int32_t opt(bdr_replay* r) override;
    void on_gate_timeout() override;
    void drain_queues() override
    {
        (void)hipStreamSynchronize(stream);
        if (side) (void)hipStreamSynchronize(side);
        if (aux) (void)hipStreamSynchronize(aux);
        if (comm_st) (void)hipStreamSynchronize(comm_st);
    }
    int32_t after_sync() override
    {
        if (!(xchg_pending_conv || xchg_pending_fc)) return BDR_OK;
        BDR_TRY(join_exchange(true, true));
        BDR_HIP(hipStreamSynchronize(stream));
        return BDR_OK;
    }
    int32_t grads_on_batch(uint64_t n, const void* obs, const int64_t* act, const void* next_obs, const float* reward, const int8_t* term) override;
    int32_t apply_grads() override;
    int exchange_plan(int which, ExchangeSeg* segs, int cap, hipStream_t* comm) override;
    int32_t exchange_begin(int seg) override;
    int32_t exchange_end(int seg) override;
    int32_t join_exchange(bool conv, bool fc);       // make the dX queue wait for pending exchanged segments
    int32_t record(float* out, int cap, int* n) override;
    void record_keys(std::vector<std::string>& keys) override;
    uint64_t param_count(int which) override;
    int32_t get_params(int which, float* out, uint64_t n) override;
    int32_t set_params(int which, const float* in, uint64_t n) override;
    float* arena(int which, size_t* n) override;
    int32_t save(const char* dir) override;
    int32_t load(const char* dir) override;
};

namespace {


void free_batch_buffers(DqnCnn* a)
{
    for (int z = 0; z < MAXZ; ++z) {
        (void)hipFree(a->a1[z]); (void)hipFree(a->a2[z]); (void)hipFree(a->a3[z]);
        (void)hipFree(a->p1[z]); (void)hipFree(a->h1[z]); (void)hipFree(a->qv[z]);
        a->a1[z] = a->a2[z] = a->a3[z] = a->p1[z] = a->h1[z] = a->qv[z] = nullptr;
    }
    (void)hipFree(a->dh1); (void)hipFree(a->dy3); (void)hipFree(a->dy2); (void)hipFree(a->dy1);
    (void)hipFree(a->dq); (void)hipFree(a->pred); (void)hipFree(a->tgt); (void)hipFree(a->loss_row);
    (void)hipFree(a->part);
    a->dh1 = a->dy3 = a->dy2 = a->dy1 = a->dq = a->pred = a->tgt = a->loss_row = a->part = nullptr;
}

int32_t ensure_batch(DqnCnn* a, int B)
{
    if (B <= a->B) return BDR_OK;
    BDR_HIP(hipStreamSynchronize(a->stream));
    free_batch_buffers(a);
    const int A = a->ar.A;
    for (int z = 0; z < MAXZ; ++z) {
        BDR_TRY(alloc_f(&a->a1[z], (size_t)B * 400 * 32));
        BDR_TRY(alloc_f(&a->a2[z], (size_t)B * 81 * 64));
        BDR_TRY(alloc_f(&a->a3[z], (size_t)B * 49 * 64));
        BDR_TRY(alloc_f(&a->p1[z], (size_t)L1_SPLIT * B * 512));
        BDR_TRY(alloc_f(&a->h1[z], (size_t)B * 512));
        BDR_TRY(alloc_f(&a->qv[z], (size_t)B * A));
    }
    BDR_TRY(alloc_f(&a->dh1, (size_t)B * 512));
    BDR_TRY(alloc_f(&a->dy3, (size_t)B * 49 * 64));
    BDR_TRY(alloc_f(&a->dy2, (size_t)B * 81 * 64));
    BDR_TRY(alloc_f(&a->dy1, (size_t)B * 400 * 32));
    BDR_TRY(alloc_f(&a->dq, B)); BDR_TRY(alloc_f(&a->pred, B)); BDR_TRY(alloc_f(&a->tgt, B));
    BDR_TRY(alloc_f(&a->loss_row, B));
    const DwPlan p = dw_plan(B, a->ar.ns);
    BDR_TRY(alloc_f(&a->part, p.total));
    a->part_floats = p.total;
    a->B = B;
    return BDR_OK;
}

// Conv partial reduction and the Adam step in ONE launch (one launch boundary less on the critical path):
//   blocks [0, reduce_blocks)  : k_reduce_partials3's job for 32 conv-gradient elements each, immediately followed by
//                                the Adam update of those same elements (the conv layers: 78 k of the 1.69 M parameters);
//   remaining blocks           : Adam over the rest of the arena (l1, l2), four elements per thread.
// Element formulas are k_reduce_partials3's and k_adam's, unchanged.
struct ConvReduceAdamArgs {
    Reduce3Args r;
    float* p; const float* g; float* m; float* v;
    float* gbase;            // start of the gradient arena (segment gradients live at seg.g = gbase + offset)
    size_t rest0_4, n4;      // Adam's f32x4 range [rest0_4, n4) = everything behind the conv segments
    AdamScalars s;
    int reduce_blocks;
    const unsigned* poison;  // sig + SIG_ERR: a gate timed out, leave the parameters alone
    int reduce_only;         // gradients only (the optimizer step follows an all-reduce: synchronous data-parallel mode)
    unsigned long long* applied; unsigned long long step;   // *applied = step by a launch that was not skipped (on_gate_timeout rolls the conv segment's step number back to it)
    uint16_t* cpl;           // the parameter set's bf16 planes of W2 / W3 (segments 1, 2), written with the parameters; null: none
};
// Schedule 3 cross-queue ordering without barrier packets (igemm.hpp start_signal).
// k_gate: one wave; returns once *flag has reached `epoch` (wrap-safe compare).  It holds one wave slot while it waits, so
// it cannot starve the producer; a producer that never arrives trips the time limit instead of hanging the queue
// (sig[SIG_ERR] is checked at the next synchronisation).
constexpr int SIG_HEAD = 0, SIG_DXL1 = 1, SIG_DXC3 = 2, SIG_SIDE = 3, SIG_GATHER = 4, SIG_TEST = 5, SIG_TEST_ERR = 6, SIG_ERR = 7, SIG_TGT = 8, SIG_TRACK = 9, SIG_MAIN_END = 10, SIG_XCONV = 11, SIG_XFC = 14;   // (12, 13: trace timestamp)
// A gate that times out POISONS the agent: sig[SIG_ERR] (and dev_err[ERR_GATE], the word the host polls) is set, every later
// gate returns at once instead of waiting another 10 s, and the kernels that write parameters (k_reduce_adam, the l1 / l2
// k_adam) skip their update while the flag is up - kernels behind a failed gate run unordered, so their gradients may be
// incomplete, but the parameters, Adam moments and checkpoints stay those of the last good step.  The host sees the flag at
// its next synchronisation or poll (bdr_agent::err_check / err_poll), reports BDR_ERR_HIP, clears it and continues with
// event ordering (schedule 1).
__global__ __launch_bounds__(64) void k_gate(unsigned* sig, int which, unsigned epoch, unsigned long long* waited, int publish,
                                             unsigned long long limit = 1000000000ull /* 10 s of the 100 MHz clock */, int err_slot = SIG_ERR,
                                             unsigned* dev_err = nullptr)
{
    if (threadIdx.x != 0) return;
    // like start_signal: this kernel has started, so everything queued before it on its stream is complete
    if (publish >= 0) __hip_atomic_store(sig + publish, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (err_slot == SIG_ERR && __hip_atomic_load(sig + SIG_ERR, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;   // poisoned
    const unsigned long long t0 = wall_clock64();   // 100 MHz
    while ((int)(__hip_atomic_load(sig + which, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - epoch) < 0) {
        __builtin_amdgcn_s_sleep(4);
        if (wall_clock64() - t0 > limit || (err_slot == SIG_ERR && __hip_atomic_load(sig + SIG_ERR, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
            __hip_atomic_store(sig + err_slot, 1u + (unsigned)which, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (dev_err) __hip_atomic_store(dev_err + bdr_agent::ERR_GATE, 1u + (unsigned)which, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
    }
    if (waited) {   // BDR_GATE_TRACE: time spent waiting, per gate site; for the join also how long the flag had been set
        const unsigned long long dt = wall_clock64() - t0;
        waited[0] += dt; waited[1] += 1;
        unsigned long long* mx = waited + 22 - (which == SIG_SIDE ? 4 : which);   // (waited = trace + 2 site: the site's longest wait at trace[22 + site])
        if (dt > *mx) *mx = dt;
        if (which == SIG_SIDE) {
            const unsigned long long ts = *reinterpret_cast<volatile unsigned long long*>(sig + 12);
            if (ts && t0 > ts) waited[12] += t0 - ts;
        }
    }
}
// k_signal: "everything queued before me on this stream is complete" (same argument as start_signal)
__global__ __launch_bounds__(64) void k_signal(unsigned* sig, int which, unsigned epoch)
{
    if (threadIdx.x == 0 && which == SIG_SIDE) *reinterpret_cast<volatile unsigned long long*>(sig + 12) = wall_clock64();   // (trace only)
    if (threadIdx.x == 0) __hip_atomic_store(sig + which, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void k_copy_word_unless(unsigned long long* dst, const unsigned long long* src, const unsigned* poison) { if (!(poison && *poison)) *dst = *src; }

__global__ __launch_bounds__(256) void k_reduce_adam(ConvReduceAdamArgs a)
{
    const bool poisoned = a.reduce_only || (a.poison && *a.poison != 0);
    if (a.applied && !poisoned && blockIdx.x == 0 && threadIdx.x == 0) *a.applied = a.step;
    if ((int)blockIdx.x >= a.reduce_blocks) {
        if (poisoned) return;
        const size_t i = a.rest0_4 + (size_t)(blockIdx.x - a.reduce_blocks) * 256 + threadIdx.x;
        if (i >= a.n4) return;
        f32x4 pp = reinterpret_cast<f32x4*>(a.p)[i], gg = reinterpret_cast<const f32x4*>(a.g)[i];
        f32x4 mm = reinterpret_cast<f32x4*>(a.m)[i], vv = reinterpret_cast<f32x4*>(a.v)[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float pe = pp[j], me = mm[j], ve = vv[j];
            adam_element(pe, gg[j], me, ve, a.s);
            pp[j] = pe; mm[j] = me; vv[j] = ve;
        }
        reinterpret_cast<f32x4*>(a.p)[i] = pp;
        reinterpret_cast<f32x4*>(a.m)[i] = mm;
        reinterpret_cast<f32x4*>(a.v)[i] = vv;
        return;
    }
    __shared__ float red[8][32];
    const int s_id = (int)blockIdx.x >= a.r.seg[2].wg0 ? 2 : ((int)blockIdx.x >= a.r.seg[1].wg0 ? 1 : 0);
    const ReduceSeg& sg = a.r.seg[s_id];
    const int o = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const int i = ((int)blockIdx.x - sg.wg0) * 32 + o;
    float s = 0.f;
    if (i < sg.n) {
        int c = grp;
        for (; c + 24 < sg.chunks; c += 32) {
            const float v0 = sg.part[(size_t)c * sg.stride + i], v1 = sg.part[(size_t)(c + 8) * sg.stride + i];
            const float v2 = sg.part[(size_t)(c + 16) * sg.stride + i], v3 = sg.part[(size_t)(c + 24) * sg.stride + i];
            s += v0; s += v1; s += v2; s += v3;
        }
        for (; c < sg.chunks; c += 8) s += sg.part[(size_t)c * sg.stride + i];
    }
    red[grp][o] = s;
    __syncthreads();
    if (grp == 0 && i < sg.n) {
        float t = red[0][o];
#pragma unroll
        for (int k = 1; k < 8; ++k) t += red[k][o];
        const float g = i < sg.n_weights ? t * sg.wscale : t;
        sg.g[i] = g;
        if (poisoned) return;
        const size_t e = (size_t)(sg.g - a.gbase) + i;      // the element's index in every arena
        float pe = a.p[e], me = a.m[e], ve = a.v[e];
        adam_element(pe, g, me, ve, a.s);
        a.p[e] = pe; a.m[e] = me; a.v[e] = ve;
        if (a.cpl && s_id >= 1 && i < sg.n_weights) conv_plane_store(a.cpl, s_id - 1, i, pe);
    }
}

int effective_sched(DqnCnn* a);   // (defined with update_critic)

// ---- the forward pass of nz network instances ------------------------------------------------------
struct NetInst { const uint8_t* x; const float* params; int slot; };

// st: the queue the launches go to (the agent's dX queue, or the other queue for the split target forward)
int32_t forward(DqnCnn* a, const NetInst* inst, int nz, int B, const TdArgs* td = nullptr, hipStream_t st = nullptr)
{
    if (!st) st = a->stream;
    const Arena& ar = a->ar;
    FwdArgs f{};
    bool uses_q = false;   // an overlapped parameter exchange of the online network may still be in flight (join_exchange)
    for (int z = 0; z < nz; ++z) uses_q = uses_q || inst[z].params == a->q;
    if (uses_q) BDR_TRY(a->join_exchange(true, false));
    {
        // conv1 on the bf16 matrix cores with exact operands (conv1_bf16.hpp)
        Conv1Args c{};
        c.M = B * 400; c.nz = nz;
        for (int z = 0; z < nz; ++z) {
            c.x[z] = inst[z].x; c.w1[z] = inst[z].params + ar.w1; c.bias[z] = inst[z].params + ar.b1; c.out[z] = a->a1[inst[z].slot];
        }
        Bracket br(a, "fwd_conv1");
        BDR_HIP(conv1_forward(ar.ns, B, st, c));
    }
    f.M = B * 81;
    for (int z = 0; z < nz; ++z) { f.x[z] = a->a1[inst[z].slot]; f.w[z] = inst[z].params + ar.w2; f.bias[z] = inst[z].params + ar.b2; f.out[z] = a->a2[inst[z].slot]; }
    bool b3 = a->conv_b3;
    FwdB3Args fb{};
    for (int z = 0; z < nz && b3; ++z) b3 = inst[z].params == a->q || inst[z].params == a->q_tgt;
    for (int z = 0; z < nz && b3; ++z) {
        const int set = inst[z].params == a->q ? 0 : 1;
        if (a->cpl_fresh[set] && !a->cpl_escaped[set]) continue;
        hipLaunchKernelGGL(k_conv_planes, dim3((512 * 64 + 576 * 64 + 255) / 256), dim3(256), 0, st, inst[z].params + ar.w2, inst[z].params + ar.w3, a->cpl[set]);
        BDR_HIP(hipGetLastError());
        a->cpl_fresh[set] = true;
    }
    if (b3) {
        static_cast<FwdArgs&>(fb) = f;
        for (int z = 0; z < nz; ++z) fb.wpl[z] = a->cpl[inst[z].params == a->q ? 0 : 1] + CPL_W2;
        Bracket br(a, "fwd_conv2");
        BDR_HIP((launch_igemm_b3<FwdC2B3, 6>(st, dim3(m_tiles<FwdC2B3>(f.M) * n_tiles<FwdC2B3>(), 1, nz), fb)));
    } else { Bracket br(a, "fwd_conv2"); BDR_HIP((launch_igemm<FwdC2, TEAMS_FWD_C2>(st, dim3(m_tiles<FwdC2>(f.M) * n_tiles<FwdC2>(), 1, nz), f))); }
    f.M = B * 49;
    for (int z = 0; z < nz; ++z) { f.x[z] = a->a2[inst[z].slot]; f.w[z] = inst[z].params + ar.w3; f.bias[z] = inst[z].params + ar.b3; f.out[z] = a->a3[inst[z].slot]; }
    if (b3) {
        static_cast<FwdArgs&>(fb) = f;
        for (int z = 0; z < nz; ++z) fb.wpl[z] = a->cpl[inst[z].params == a->q ? 0 : 1] + CPL_W3;
        Bracket br(a, "fwd_conv3");
        BDR_HIP((launch_igemm_b3<FwdC3B3, 6>(st, dim3(m_tiles<FwdC3B3>(f.M) * n_tiles<FwdC3B3>(), 1, nz), fb)));
    } else { Bracket br(a, "fwd_conv3"); BDR_HIP((launch_igemm<FwdC3D, TEAMS_FWD_C3>(st, dim3(m_tiles<FwdC3D>(f.M) * n_tiles<FwdC3D>(), 1, nz), f))); }
    f.M = B; f.nkt_per_split = (98 + L1_SPLIT - 1) / L1_SPLIT;
    for (int z = 0; z < nz; ++z) { f.x[z] = a->a3[inst[z].slot]; f.w[z] = inst[z].params + ar.w4; f.bias[z] = nullptr; f.out[z] = a->p1[inst[z].slot]; }
    if (uses_q) BDR_TRY(a->join_exchange(false, true));
    {
        Bracket br(a, "fwd_l1");
#if BDR_L1_XSPLIT
        static_assert(L1_SPLIT == 8, "FwdL1X: one k-slice per XCD");
        BDR_HIP((launch_igemm<FwdL1X, TEAMS_FWD_L1>(st, dim3(m_tiles<FwdL1X>(B) * n_tiles<FwdL1X>() * 8, 1, nz), f)));
#else
        if (nz % 2 == 0) BDR_HIP((launch_igemm<FwdL1Z2, TEAMS_FWD_L1>(st, dim3(((B + 63) / 64) * 16, L1_SPLIT, nz / 2), f)));
        else BDR_HIP((launch_igemm<FwdL1, TEAMS_FWD_L1>(st, dim3(((B + 63) / 64) * 8, L1_SPLIT, nz), f)));
#endif
    }
    HeadArgs h{};
    h.B = B; h.A = ar.A; h.S = L1_SPLIT;
    for (int z = 0; z < nz; ++z) {
        h.p1[z] = a->p1[inst[z].slot]; h.b4[z] = inst[z].params + ar.b4; h.w5[z] = inst[z].params + ar.w5;
        h.b5[z] = inst[z].params + ar.b5; h.h1[z] = a->h1[inst[z].slot]; h.q[z] = a->qv[inst[z].slot];
    }
    {
        Bracket br(a, td ? "head_fwd_td" : "head_fwd");
        const dim3 grid((B + HEAD_ROWS - 1) / HEAD_ROWS), block(64 * HEAD_ROWS * nz);
        const TdArgs tv = td ? *td : TdArgs{};
        // with the two-stream backward schedule the head kernel's own packet completes the first fork event
        hipEvent_t stop = td && effective_sched(a) == 1 && a->kev ? a->ev_fork[0] : nullptr;
        const int tdi = td ? 1 : 0;
        if (ar.A <= 8) hipExtLaunchKernelGGL((k_head<L1_SPLIT, 8>), grid, block, 0, st, nullptr, stop, 0, h, tv, nz, tdi);
        else if (ar.A <= 24) hipExtLaunchKernelGGL((k_head<L1_SPLIT, 24>), grid, block, 0, st, nullptr, stop, 0, h, tv, nz, tdi);
        else hipExtLaunchKernelGGL((k_head<L1_SPLIT, 64>), grid, block, 0, st, nullptr, stop, 0, h, tv, nz, tdi);
        BDR_HIP(hipGetLastError());
    }
    return BDR_OK;
}

AdamScalars adam_scalars(const bdr_dqn_config& c, uint64_t step)
{
    return adam_scalars_for(c.opt_kind == BDR_OPT_ADAMW, c.lr, c.beta1, c.beta2, c.eps, c.weight_decay, step);
}

// Dqn::update_critic on a device-resident batch (dqn/base.rs:60-160)
// one-wave wait for flag `which` to reach `epoch` on stream st; publish >= 0: also publishes that flag at its start
int32_t launch_gate(DqnCnn* a, hipStream_t st, int which, unsigned epoch, int publish)
{
    // trace slot: 2 counters per site; sites 0..3 = flags on the weight-gradient queue, 4 = the gates on the dX queue
    unsigned long long* tr = a->gate_trace && which <= SIG_SIDE ? a->gate_trace + 2 * (st == a->stream ? 4 : which) : nullptr;
    hipLaunchKernelGGL(k_gate, dim3(1), dim3(64), 0, st, a->sig, which, epoch, tr, publish, a->gate_limit, SIG_ERR, a->dev_err);
    BDR_HIP(hipGetLastError());
    return BDR_OK;
}

// Gate kernels only work if the waiting stream and the producing stream sit on different hardware queues: HIP multiplexes
// its streams onto a bounded pool of HSA queues (GPU_MAX_HW_QUEUES), and a gate sharing an in-order queue with its producer
// would wait for a kernel queued behind it.  Checked once per pair when the streams exist: a gate with a 100 ms limit on
// `waiter`, then the signal on `producer`; if the gate times out the queues alias.
int32_t queues_independent(DqnCnn* a, hipStream_t waiter, hipStream_t producer, bool* ok)
{
    a->test_epoch += 1;
    BDR_HIP(hipMemsetAsync(a->sig + SIG_TEST_ERR, 0, sizeof(unsigned), waiter));
    hipLaunchKernelGGL(k_gate, dim3(1), dim3(64), 0, waiter, a->sig, SIG_TEST, a->test_epoch, (unsigned long long*)nullptr, -1, 10000000ull, SIG_TEST_ERR,
                       (unsigned*)nullptr);
    BDR_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_signal, dim3(1), dim3(64), 0, producer, a->sig, SIG_TEST, a->test_epoch);
    BDR_HIP(hipGetLastError());
    BDR_HIP(hipStreamSynchronize(waiter));
    BDR_HIP(hipStreamSynchronize(producer));
    unsigned err = 0;
    BDR_HIP(hipMemcpy(&err, a->sig + SIG_TEST_ERR, sizeof err, hipMemcpyDeviceToHost));
    *ok = err == 0;
    return BDR_OK;
}

// Gates are safe only while ONE agent of the process uses them: streams of different agents can alias the same hardware queue
// (GPU_MAX_HW_QUEUES), and two agents whose gates sit in front of each other's producers could wait in a cycle.  The first
// agent that runs a gated update owns the process-wide token until it is destroyed; every other agent orders its two queues
// with events (schedule 1) for as long as the token is taken.
std::atomic<DqnCnn*> g_gate_owner{nullptr};
bool claim_gates(DqnCnn* a)
{
    if (a->holds_gate_token) return true;
    DqnCnn* expected = nullptr;
    if (g_gate_owner.compare_exchange_strong(expected, a)) { a->holds_gate_token = true; return true; }
    return false;
}
int effective_sched(DqnCnn* a)
{
    if (a->prof) return 0;
    if (a->sched == 3 && !claim_gates(a)) {
        // (round-3 review: not silently) another agent of this process owns the flag-ordered schedule
        if (!a->noted_event_fallback) {
            a->noted_event_fallback = true;
            fprintf(stderr, "border_amd: another DQN agent of this process already orders its two queues with device flags; this agent orders its "
                            "queues with events (same results, ~10 %% slower steps) for as long as that agent lives\n");
        }
        return 1;
    }
    return a->sched;
}

int32_t update_critic(DqnCnn* a, int B, const uint8_t* obs, const uint8_t* next_obs, const uint8_t* act, int act_bytes,
                      const float* reward, const int8_t* term, const float* weight = nullptr, bdr_replay* per_buffer = nullptr)
{
    BDR_TRY(ensure_batch(a, B));
    BDR_TRY(a->td_buffer(B));
    a->last_reward = reward; a->last_B = B;
    const Arena& ar = a->ar;
    const bdr_dqn_config& c = a->cfg;
    // :71-74 + :91-103  slot 0 = qnet(obs), slot 1 = qnet_tgt(next_obs), slot 2 = qnet(next_obs) for double DQN
    NetInst inst[3] = {{obs, a->q, 0}, {next_obs, a->q_tgt, 1}, {next_obs, a->q, 2}};

    TdArgs t{};
    t.double_dqn = c.double_dqn;
    t.act = act; t.act_bytes = act_bytes; t.reward = reward; t.term = term;
    t.dh1 = a->dh1; t.dq = a->dq; t.pred = a->pred; t.tgt = a->tgt;
    t.loss_row = a->loss_row; t.B = B; t.A = ar.A; t.gamma = (float)c.discount_factor; t.loss_kind = c.critic_loss;
    t.weight = weight; t.td_abs = a->td_abs;
    t.has_clip = c.has_clip_td_err; t.clip_min = (float)c.clip_td_err_min; t.clip_max = (float)c.clip_td_err_max;
    t.err = a->dev_err;
    t.z_tgt = 1; t.z_sel = c.double_dqn ? 2 : -1;
    if (a->tgt_enqueued) {
        // Split forward: qnet_tgt(next_obs) is already running (or done) on the other queue (opt_inner); this queue runs the
        // online instance(s) only and the head kernel picks the target rows up from memory once their flag is up.
        a->tgt_enqueued = false;
        NetInst on[2] = {{obs, a->q, 0}, {next_obs, a->q, 2}};
        t.q_tgt_mem = a->qv[1]; t.z_tgt = -1; t.z_sel = c.double_dqn ? 1 : -1;
        t.wait_sig = a->sig + SIG_TGT; t.wait_epoch = a->sig_epoch + 1; t.wait_limit = a->gate_limit; t.sig_err = a->sig + SIG_ERR;
        BDR_TRY(forward(a, on, c.double_dqn ? 2 : 1, B, &t));
    } else {
        BDR_TRY(forward(a, inst, c.double_dqn ? 3 : 2, B, &t));   // the TD step rides on the head kernel
    }

    // Backward.  The input-gradient chain (dX of l1 -> conv3 -> conv2) is the critical path; every weight-gradient kernel
    // only needs the dY produced one link earlier and fills only part of the chip, so it runs beside the next dX kernel.
    // Three schedules (a->sched; per-kernel profiling forces 0):
    //   0  serial, one stream;
    //   1  weight gradients on a second stream, forked / joined with events (each cross-stream dependency costs the
    //      recording queue and the waiting queue a 5-7 us bubble: tools/rocprof_timeline.py);
    //   2  one stream, no events: a dX kernel is launched as a normal (barrier) dispatch and the weight-gradient kernels that
    //      may run beside it follow as hipExtAnyOrderLaunch dispatches (no barrier bit; ignored by this runtime on gfx9);
    //   3  two streams ordered through device flags instead of events: the next dX kernel's first workgroup publishes
    //      "my predecessor is complete" (start_signal), one-wave k_gate kernels on the consuming queue wait for it.  No
    //      packet is added to the dX queue except the join gate before the reduction.
    const int sched = effective_sched(a);
    const bool ov = sched == 1;
    const bool gated = sched == 3;
    hipStream_t sd = ov || gated ? a->side : a->stream;
    const unsigned epoch = gated ? ++a->sig_epoch : 0;
    auto sigf = [&](int which) -> unsigned* { return gated ? a->sig + which : nullptr; };
    auto gate = [&](hipStream_t st, int which) -> int32_t { return launch_gate(a, st, which, epoch, -1); };
    const unsigned any = sched == 2 ? hipExtAnyOrderLaunch : 0;
    // a->kev: the fork / join events are completed by the producing kernel's own dispatch packet (launch `stop` event)
    // instead of a marker packet behind it
    auto kev = [&](int k) -> hipEvent_t { return ov && a->kev ? a->ev_fork[k] : nullptr; };
    auto fork = [&](int k) -> int32_t {
        if (!ov) return BDR_OK;
        if (!a->kev) BDR_HIP(hipEventRecord(a->ev_fork[k], a->stream));
        BDR_HIP(hipStreamWaitEvent(a->side, a->ev_fork[k], 0));
        return BDR_OK;
    };
    const DwPlan pl = dw_plan(a->B, ar.ns);   // buffer layout follows the allocated batch capacity
    const bool defer = a->defer_adam;  // backward only: the optimizer step is apply_grads()
    if (!defer) a->adam_step += 1;
    const AdamScalars adam_s = adam_scalars(c, std::max<uint64_t>(a->adam_step, 1));

    auto head_bwd = [&]() -> int32_t {
        HeadBwdArgs hb{a->h1[0], a->dq, act, act_bytes, a->loss_row, a->grad + ar.w5, a->grad + ar.b5, a->loss, B, ar.A};
        Bracket br(a, "head_bwd");
        LAUNCH_FL(sd, any, nullptr, k_head_bwd, dim3(2 * ar.A + 1), dim3(256), hb);
        return BDR_OK;
    };
    auto l1_dw = [&]() -> int32_t {
        DwArgs d{a->a3[0], a->dh1, a->grad + ar.w4, 0, B};
        Bracket br(a, "bwd_l1_dw");
        LAUNCH_FL(sd, any, nullptr, k_igemm_red<DwL1>, dim3(49 * 8), dim3(256), d);
        return BDR_OK;
    };
    auto l1_dx = [&]() -> int32_t {
        DxArgs d{a->dh1, a->q + ar.w4, a->a3[0], a->dy3, B, sigf(SIG_HEAD), epoch};   // its start publishes "head done"
        Bracket br(a, "bwd_l1_dx");
        BDR_HIP((launch_igemm<DxL1, TEAMS_DX_L1>(a->stream, dim3((m_tiles<DxL1>(B) * n_tiles<DxL1>() + 7) / 8 * 8, 1, 1), d, 0, kev(1))));
        return BDR_OK;
    };
    // :150 backward_step -> Adam.  The l1 / l2 parameters (95 % of the arena) have their gradients (k_head_bwd, DwL1) and
    // their last readers of this step (k_head, DxL1) behind them once DxL1 is done, so their Adam pass - pure HBM
    // streaming - runs under the conv dX GEMMs instead of at the end of the critical path.
    auto adam_l1_l2 = [&]() -> int32_t {
        if (defer) return BDR_OK;
        Bracket br(a, "adam_l1_l2");
        const size_t r4 = (ar.total - ar.w4) / 4;
        LAUNCH_FL(sd, any, nullptr, k_adam, dim3((unsigned)((r4 + 255) / 256)), dim3(256), a->q + ar.w4, (const float*)(a->grad + ar.w4), a->m + ar.w4,
                  a->v + ar.w4, r4, adam_s, (const unsigned*)(a->sig + SIG_ERR), a->applied_step, (unsigned long long)a->adam_step);
        return BDR_OK;
    };
    auto c3_dw = [&]() -> int32_t {
        const int M = B * 49, chunks = std::min(pl.chunks_c3, (M + 31) / 32);
        DwArgs d{a->a2[0], a->dy3, a->part + pl.off_c3, pl.stride_c3, M};
        Bracket br(a, "bwd_conv3_dw");
        LAUNCH_FL(sd, any, nullptr, k_igemm_red<DwC3>, dim3(9 * chunks), dim3(256), d);
        return BDR_OK;
    };
    auto c3_dx = [&]() -> int32_t {
        DxArgs d{a->dy3, a->q + ar.w3, a->a2[0], a->dy2, B * 81, sigf(SIG_DXL1), epoch};
        Bracket br(a, "bwd_conv3_dx");
        BDR_HIP((launch_igemm<DxC3Pos, TEAMS_DX_C3>(a->stream, dim3(((B + DxC3Pos::WM * DxC3Pos::TM * 32 - 1) / (DxC3Pos::WM * DxC3Pos::TM * 32)) * n_tiles<DxC3Pos>(), 81, 1), d, 0, kev(2))));
        return BDR_OK;
    };
    auto c2_dw = [&]() -> int32_t {
        const int M = B * 81, chunks = std::min(pl.chunks_c2, (M + 31) / 32);
        DwArgs d{a->a1[0], a->dy2, a->part + pl.off_c2, pl.stride_c2, M, nullptr, 0};
        Bracket br(a, "bwd_conv2_dw");
        LAUNCH_FL(sd, any, ov && a->kev ? a->ev_join : nullptr, k_igemm_red<DwC2>, dim3(8 * chunks), dim3(256), d);
        return BDR_OK;
    };
    auto c2_dx = [&]() -> int32_t {
        DxArgs d{a->dy2, a->q + ar.w2, a->a1[0], a->dy1, B * 100, sigf(SIG_DXC3), epoch};
        Bracket br(a, "bwd_conv2_dx");
#if BDR_DXC2_MERGED == 2
        BDR_HIP((launch_igemm<DxC2MPos, TEAMS_DX_C2>(a->stream, dxc2_pos_grid<DxC2MPos>(B), d)));
#elif BDR_DXC2_MERGED
        BDR_HIP((launch_igemm<DxC2M, TEAMS_DX_C2>(a->stream, dim3(m_tiles<DxC2M>(d.M) * n_tiles<DxC2M>(), 1, 1), d)));
#else
        BDR_HIP((launch_igemm<DxC2, TEAMS_DX_C2>(a->stream, dim3((d.M + 127) / 128, 4, 1), d)));
#endif
        return BDR_OK;
    };
    auto c1_dw = [&]() -> int32_t {   // conv1 has no input gradient
        const int chunks = std::min(pl.chunks_c1, B);
        Conv1DwArgs d{obs, a->dy1, a->part + pl.off_c1, pl.stride_c1, B};
        Bracket br(a, "bwd_conv1_dw");
        BDR_HIP(launch_conv1_dw_bf16(ar.ns, dim3(chunks), a->stream, d));
        return BDR_OK;
    };

    // dqn/base.rs:143: buffer.update_priority(&ixs, &Some(td_errs)).  Nothing in this update depends on it, so with
    // concurrent schedules it runs on its own stream behind the TD step; the next batch() waits for it through the
    // buffer's `written` event.
    const bool per = per_buffer && weight;
    if (per && sched == 2) BDR_HIP(hipEventRecord(a->ev_fork[0], a->stream));
    BDR_TRY(fork(0));                  // dh1, dq, td_abs ready
    if (per) {
        hipStream_t ps = a->stream;
        if (sched != 0) {
            if (gated && a->aux_gated) BDR_TRY(gate(a->aux, SIG_HEAD));
            else {
                if (gated) BDR_HIP(hipEventRecord(a->ev_fork[0], a->stream));   // aliased queue: plain event ordering
                BDR_HIP(hipStreamWaitEvent(a->aux, a->ev_fork[0], 0));
            }
            ps = a->aux;
        }
        Bracket br(a, "per_update");
        BDR_TRY(replay_update_priority_on_stream(per_buffer, B, a->td_abs, ps));
    }
    if (gated) {
        if (!a->head_gate_enqueued) BDR_TRY(gate(sd, SIG_HEAD));   // (opt_inner enqueues it right behind a side-queue gather)
        a->head_gate_enqueued = false;
        // dX queue: DxL1, DxC3, DxC2 (each start publishing its predecessor), then conv2's and conv1's dW.
        // other queue: gate -> head_bwd, DwL1; gate -> DwC3; then the l1 / l2 Adam pass (a streaming kernel: beside a dX
        // GEMM it starves for workgroup slots - 27 us instead of 8 - so it goes last).  With the position-class conv3 dX
        // (40 % fewer products) the dX queue has room for conv2's dW; left on the other queue it made that one the longer
        // by 17 us (4 230 vs 4 515 opt-steps/s on the same box).
        BDR_TRY(head_bwd()); BDR_TRY(l1_dw());
        BDR_TRY(l1_dx());
        BDR_TRY(gate(sd, SIG_DXL1)); BDR_TRY(c3_dw()); BDR_TRY(adam_l1_l2());
        hipLaunchKernelGGL(k_signal, dim3(1), dim3(64), 0, sd, a->sig, SIG_SIDE, epoch);
        BDR_HIP(hipGetLastError());
        BDR_TRY(c3_dx());
        BDR_TRY(c2_dx());
        { hipStream_t side = sd; sd = a->stream; BDR_TRY(c2_dw()); sd = side; }
        BDR_TRY(c1_dw());
        BDR_TRY(gate(a->stream, SIG_SIDE));   // join: all weight-gradient partials complete
    } else if (sched == 2) {
        BDR_TRY(l1_dx());  BDR_TRY(head_bwd()); BDR_TRY(l1_dw());        // barrier, any-order, any-order
        BDR_TRY(c3_dx());  BDR_TRY(adam_l1_l2()); BDR_TRY(c3_dw());
        BDR_TRY(c2_dx());  BDR_TRY(c2_dw());
        BDR_TRY(c1_dw());
    } else {
        BDR_TRY(head_bwd()); BDR_TRY(l1_dw());
        BDR_TRY(l1_dx());
        BDR_TRY(fork(1));              // dy3 ready
        BDR_TRY(adam_l1_l2()); BDR_TRY(c3_dw());
        BDR_TRY(c3_dx());
        BDR_TRY(fork(2));              // dy2 ready
        BDR_TRY(c2_dw());
        BDR_TRY(c2_dx());
        BDR_TRY(c1_dw());
        if (ov) {                      // join: all weight-gradient partials complete
            if (!a->kev) BDR_HIP(hipEventRecord(a->ev_join, a->side));
            BDR_HIP(hipStreamWaitEvent(a->stream, a->ev_join, 0));
        }
    }
    {   // conv1..conv3 partials -> gradient arena, one launch
        Reduce3Args r{};
        const int Ms[3] = {B * 400, B * 81, B * 49};
        const int plc[3] = {pl.chunks_c1, pl.chunks_c2, pl.chunks_c3};
        const size_t offs[3] = {pl.off_c1, pl.off_c2, pl.off_c3}, strides[3] = {pl.stride_c1, pl.stride_c2, pl.stride_c3};
        const size_t gw[3] = {ar.w1, ar.w2, ar.w3};
        const int nw[3] = {(int)ar.n_w1(), 512 * 64, 576 * 64}, nb[3] = {32, 64, 64};
        int wg = 0;
        for (int k = 0; k < 3; ++k) {
            const int nchunks = k == 0 ? std::min(plc[0], B) : std::min(plc[k], (Ms[k] + 31) / 32);   // conv1: one partial per workgroup
            r.seg[k] = ReduceSeg{a->part + offs[k], strides[k], nchunks, a->grad + gw[k], nw[k] + nb[k], nw[k],
                                 k == 0 ? INV255 : 1.0f, wg};
            wg += (nw[k] + nb[k] + 31) / 32;
        }
        // the conv layers' Adam step rides on their partial reduction (k_reduce_adam); l1 / l2: adam_l1_l2 above
        ConvReduceAdamArgs ra{};
        ra.r = r; ra.p = a->q; ra.g = a->grad; ra.m = a->m; ra.v = a->v; ra.gbase = a->grad;
        ra.rest0_4 = ra.n4 = ar.w4 / 4; ra.s = a->conv_step_lag ? adam_scalars(c, (uint64_t)std::max<int64_t>((int64_t)a->adam_step - a->conv_step_lag, 1)) : adam_s;
        ra.applied = defer ? nullptr : a->applied_step + 1; ra.step = (unsigned long long)((int64_t)a->adam_step - a->conv_step_lag);
        ra.reduce_blocks = wg; ra.poison = a->sig + SIG_ERR; ra.reduce_only = defer ? 1 : 0;
        ra.cpl = a->conv_b3 && !defer ? a->cpl[0] : nullptr;
        Bracket br(a, "reduce_adam");
        hipLaunchKernelGGL(k_reduce_adam, dim3(wg), dim3(256), 0, a->stream, ra);
        BDR_HIP(hipGetLastError());
        if (ra.cpl) a->cpl_fresh[0] = true;   // (a poisoned launch skips parameters and planes alike; on_gate_timeout marks them stale anyway)
    }
    return BDR_OK;
}

int32_t soft_update(DqnCnn* a)
{
    const size_t n4 = a->ar.total / 4;
    const float tau = (float)a->cfg.tau, omt = (float)(1.0 - a->cfg.tau);
    Bracket br(a, "track");
    a->planes_stale(1);
    hipLaunchKernelGGL(k_track, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, a->stream, a->q_tgt, a->q, n4, tau, omt, (const unsigned*)(a->sig + SIG_ERR));
    BDR_HIP(hipGetLastError());
    if (a->sig_epoch != 0 && a->split_fwd && !a->prof) {   // the other queue's next target forward must see the new target parameters
        a->track_epoch = a->sig_epoch;
        hipLaunchKernelGGL(k_signal, dim3(1), dim3(64), 0, a->stream, a->sig, SIG_TRACK, a->track_epoch);
        BDR_HIP(hipGetLastError());
        a->track_wait_pending = true;
    }
    return BDR_OK;
}

// dqn/base.rs:182-200 bookkeeping after the n_updates_per_opt updates
int32_t after_updates(DqnCnn* a)
{
    a->soft_update_counter += 1;
    if (a->soft_update_counter == a->cfg.soft_update_interval) {
        a->soft_update_counter = 0;
        BDR_TRY(soft_update(a));
    }
    a->n_opts += 1;
    return BDR_OK;
}

// the optimizer step over the whole arena (the fused path splits it between k_reduce_adam and the l1 / l2 k_adam)
int32_t adam_all(DqnCnn* a)
{
    a->adam_step += 1;
    a->planes_stale(0);
    Bracket br(a, "adam_all");
    const size_t n4 = a->ar.total / 4;
    const unsigned* poison = (const unsigned*)(a->sig + SIG_ERR);
    // ranges of the arena with their own optimizer step number: one (the whole arena) unless a gate time-out once fell between the fused
    // path's two passes - then the conv segment [0, w4) continues from the step number ITS moments are at
    struct Range { size_t off, n; uint64_t step; int slot; } rg[2] = {{0, a->ar.total, a->adam_step, 0}, {0, 0, 0, 1}};
    int nr = 1;
    if (a->conv_step_lag) {
        const uint64_t cs = (uint64_t)std::max<int64_t>((int64_t)a->adam_step - a->conv_step_lag, 1);
        rg[0] = Range{a->ar.w4, a->ar.total - a->ar.w4, a->adam_step, 0}; rg[1] = Range{0, a->ar.w4, cs, 1}; nr = 2;
    }
    for (int k = 0; k < nr; ++k) {
        const Range& g = rg[k];
        const AdamScalars sc = adam_scalars(a->cfg, g.step);
        if (a->amsgrad) { BDR_TRY(launch_adam_amsgrad(a->stream, a->q + g.off, a->grad + g.off, a->m + g.off, a->v + g.off, a->vmax + g.off, g.n, sc, poison, a->applied_step + g.slot, (unsigned long long)g.step)); continue; }
        hipLaunchKernelGGL(k_adam, dim3((unsigned)((g.n / 4 + 255) / 256)), dim3(256), 0, a->stream, a->q + g.off, (const float*)(a->grad + g.off), a->m + g.off, a->v + g.off, g.n / 4,
                           sc, poison, a->applied_step + g.slot, (unsigned long long)g.step);
        BDR_HIP(hipGetLastError());
    }
    if (nr == 1) {   // the whole-arena pass stands for both segments
        hipLaunchKernelGGL(k_copy_word_unless, dim3(1), dim3(1), 0, a->stream, a->applied_step + 1, (const unsigned long long*)a->applied_step, poison);
        BDR_HIP(hipGetLastError());
    }
    (void)n4;
    return BDR_OK;
}

int32_t opt_inner(DqnCnn* a, bdr_replay* r)
{
    BDR_REQUIRE(r->obs_bytes == (uint64_t)a->cfg.net.n_stack * 84 * 84, "replay obs rows (%llu B) do not match the AtariCnn input",
                (unsigned long long)r->obs_bytes);
    BDR_REQUIRE(r->act_bytes >= 8, "discrete actions are stored as i64");
    BDR_REQUIRE(r->device == a->device, "agent and replay buffer live on different devices");
    { Bracket br(a, "_null"); }   // empty bracket: the event pair's own cost, subtracted by bench.py
    // any buffer growth (hipFree drains the device) happens here, before this update's gates are queued
    BDR_TRY(ensure_batch(a, (int)a->cfg.batch_size));
    BDR_TRY(a->td_buffer(a->cfg.batch_size));
    for (uint64_t u = 0; u < a->cfg.n_updates_per_opt; ++u) {
        if (effective_sched(a) == 3 && a->side_gather) {
            // The gather does not depend on the previous update, and the weight-gradient queue is idle from the end of one
            // update to the next head kernel: the batch is gathered there, into the buffer set the previous update is not
            // using (its conv1 dW may still be reading the other one), while the dX queue finishes the previous update.
            // That queue's first gate of the new update is enqueued right behind the gather and publishes its completion;
            // the dX queue waits for it with its own gate in front of conv1.
            const unsigned epoch = a->sig_epoch + 1;     // the epoch update_critic is about to take
            BDR_TRY(replay_flip_batch(r, a->cfg.batch_size));
            // Third queue (default since round 6; BDR_TQ=0: off): without prioritized replay the agent's third stream (its tree-update
            // queue) is free, and the gather + target forward of update n go there.  They depend on nothing but "head(n-1) is done",
            // so they run beside update n-1's backward instead of waiting behind this update's weight-gradient work.  Round 2 measured
            // this SLOWER (4 420 vs 4 545: the dX chain lost more than the queue gained); with round 6's kernels it is never slower and
            // up to 2 % faster - same box, interleaved: 5 296 / 5 289 vs 5 178 / 5 186, 5 097 / 5 093 vs 5 048 / 5 025, 4 972 / 4 974 vs
            // 4 972 / 4 954.  Same bits (tests/test_gpu_dqn.py::test_opt_stream_is_deterministic_and_overlap_invariant).
            const bool tq3 = a->split_fwd && a->three_queues && !r->per && a->aux_gated;
            hipStream_t tq = tq3 ? a->aux : a->side;
            if (tq3) BDR_TRY(launch_gate(a, tq, SIG_HEAD, epoch - 1, -1));
            BDR_TRY(replay_sample_on_stream(r, a->cfg.batch_size, tq));
            if (a->split_fwd) {
                // Split forward.  qnet_tgt(next_obs) depends on nothing the previous update computes (the target parameters
                // only change at a soft update), and the host runs a step or two ahead of the device: off the dX queue, the
                // target network's forward overlaps the PREVIOUS update's backward, whose GEMMs leave a third of the CUs' issue
                // slots idle (tile quantisation, prologues, epilogues).  The dX queue then runs the online instance only.
                // Order: gather -> signal(GATHER) -> [wait for the previous update's soft update, if it had one] -> target
                // conv1..head -> "target rows complete" (TGT), which the head kernel on the dX queue waits for in-kernel.
                hipLaunchKernelGGL(k_signal, dim3(1), dim3(64), 0, tq, a->sig, SIG_GATHER, epoch);
                BDR_HIP(hipGetLastError());
                if (a->track_wait_pending) {
                    BDR_TRY(launch_gate(a, tq, SIG_TRACK, a->track_epoch, -1));
                    a->track_wait_pending = false;
                }
                NetInst tg[1] = {{r->b_next, a->q_tgt, 1}};
                BDR_TRY(forward(a, tg, 1, (int)a->cfg.batch_size, nullptr, tq));
                if (tq3) {
                    hipLaunchKernelGGL(k_signal, dim3(1), dim3(64), 0, tq, a->sig, SIG_TGT, epoch);
                    BDR_HIP(hipGetLastError());
                    BDR_TRY(launch_gate(a, a->side, SIG_HEAD, epoch, -1));
                } else {
                    BDR_TRY(launch_gate(a, a->side, SIG_HEAD, epoch, SIG_TGT));   // its start publishes TGT
                }
                a->tgt_enqueued = true;
            } else {
                BDR_TRY(launch_gate(a, a->side, SIG_HEAD, epoch, SIG_GATHER));
            }
            a->head_gate_enqueued = true;
            BDR_TRY(launch_gate(a, a->stream, SIG_GATHER, epoch, -1));
        } else {
            Bracket br(a, "sample");
            BDR_TRY(replay_sample_on_stream(r, a->cfg.batch_size, a->stream));
        }
        a->defer_adam = a->grad_comm != nullptr || a->amsgrad;
        const int32_t st = update_critic(a, (int)a->cfg.batch_size, r->b_obs, r->b_next, r->b_act, (int)r->act_bytes, r->b_reward, r->b_term,
                                         replay_batch_weights(r), r);
        a->defer_adam = false;
        BDR_TRY(st);
        if (a->grad_comm) {   // synchronous data-parallel step: mean gradient over the ranks, then everybody's optimizer step
            Bracket br(a, "grad_allreduce"); BDR_TRY(a->grad_reduce(a, a->grad_comm));
        }
        if (a->grad_comm || a->amsgrad) BDR_TRY(adam_all(a));
    }
    return after_updates(a);
}

int32_t fill_record(DqnCnn* a, int B, const float* reward_dev, bdr_dqn_record* rec)
{
    BDR_HIP(hipMemcpyAsync(&rec->loss, a->loss, 4, hipMemcpyDeviceToHost, a->stream));
    BDR_HIP(hipStreamSynchronize(a->stream));
    rec->has_verbose = 0;
    if (a->cfg.record_verbose_level >= 2) {
        std::vector<float> p(B), t(B), rw(B);
        BDR_HIP(hipMemcpy(p.data(), a->pred, B * 4, hipMemcpyDeviceToHost));
        BDR_HIP(hipMemcpy(t.data(), a->tgt, B * 4, hipMemcpyDeviceToHost));
        BDR_HIP(hipMemcpy(rw.data(), reward_dev, B * 4, hipMemcpyDeviceToHost));
        double sp = 0, st = 0, sr = 0, sd = 0;
        for (int i = 0; i < B; ++i) { sp += p[i]; st += t[i]; sr += rw[i]; sd += (double)t[i] - p[i]; }
        rec->pred_mean = (float)(sp / B); rec->tgt_mean = (float)(st / B); rec->reward_mean = (float)(sr / B);
        rec->tgt_minus_pred_mean = (float)(sd / B);
        rec->has_verbose = 1;
    }
    return BDR_OK;
}

// ---- reference <-> internal parameter layouts ------------------------------------------------------
// reference order: c1.weight[32][n_stack][8][8] c1.bias c2.weight[64][32][4][4] c2.bias c3.weight[64][64][3][3]
// c3.bias l1.weight[512][3136 (c,h,w)] l1.bias l2.weight[A][512] l2.bias   (cnn/base.rs:23-36)
size_t ref_param_count(int A, int ns) { return (size_t)2048 * ns + 32 + 32768 + 64 + 36864 + 64 + (size_t)512 * 3136 + 512 + (size_t)A * 512 + A; }

void to_internal(const Arena& ar, const float* ref, float* in)
{
    std::fill(in, in + ar.total, 0.f);
    const float* p = ref;
    const int K1 = 64 * ar.ns;   // (c, kh, kw) of c1.weight[o] is the internal k order
    for (int o = 0; o < 32; ++o) for (int k = 0; k < K1; ++k) in[ar.w1 + (size_t)k * 32 + o] = p[(size_t)o * K1 + k];
    p += (size_t)32 * K1; std::copy(p, p + 32, in + ar.b1); p += 32;
    for (int o = 0; o < 64; ++o) for (int c = 0; c < 32; ++c) for (int kh = 0; kh < 4; ++kh) for (int kw = 0; kw < 4; ++kw)
        in[ar.w2 + (size_t)((kh * 4 + kw) * 32 + c) * 64 + o] = p[((size_t)(o * 32 + c) * 4 + kh) * 4 + kw];
    p += 32768; std::copy(p, p + 64, in + ar.b2); p += 64;
    for (int o = 0; o < 64; ++o) for (int c = 0; c < 64; ++c) for (int kh = 0; kh < 3; ++kh) for (int kw = 0; kw < 3; ++kw)
        in[ar.w3 + (size_t)((kh * 3 + kw) * 64 + c) * 64 + o] = p[((size_t)(o * 64 + c) * 3 + kh) * 3 + kw];
    p += 36864; std::copy(p, p + 64, in + ar.b3); p += 64;
    for (int o = 0; o < 512; ++o) for (int c = 0; c < 64; ++c) for (int hw = 0; hw < 49; ++hw)
        in[ar.w4 + (size_t)(hw * 64 + c) * 512 + o] = p[(size_t)o * 3136 + c * 49 + hw];
    p += (size_t)512 * 3136; std::copy(p, p + 512, in + ar.b4); p += 512;
    std::copy(p, p + (size_t)ar.A * 512, in + ar.w5);   // l2: [A][512] on both sides
    p += (size_t)ar.A * 512; std::copy(p, p + ar.A, in + ar.b5);
}

void to_reference(const Arena& ar, const float* in, float* ref)
{
    float* p = ref;
    const int K1 = 64 * ar.ns;
    for (int o = 0; o < 32; ++o) for (int k = 0; k < K1; ++k) p[(size_t)o * K1 + k] = in[ar.w1 + (size_t)k * 32 + o];
    p += (size_t)32 * K1; std::copy(in + ar.b1, in + ar.b1 + 32, p); p += 32;
    for (int o = 0; o < 64; ++o) for (int c = 0; c < 32; ++c) for (int kh = 0; kh < 4; ++kh) for (int kw = 0; kw < 4; ++kw)
        p[((size_t)(o * 32 + c) * 4 + kh) * 4 + kw] = in[ar.w2 + (size_t)((kh * 4 + kw) * 32 + c) * 64 + o];
    p += 32768; std::copy(in + ar.b2, in + ar.b2 + 64, p); p += 64;
    for (int o = 0; o < 64; ++o) for (int c = 0; c < 64; ++c) for (int kh = 0; kh < 3; ++kh) for (int kw = 0; kw < 3; ++kw)
        p[((size_t)(o * 64 + c) * 3 + kh) * 3 + kw] = in[ar.w3 + (size_t)((kh * 3 + kw) * 64 + c) * 64 + o];
    p += 36864; std::copy(in + ar.b3, in + ar.b3 + 64, p); p += 64;
    for (int o = 0; o < 512; ++o) for (int c = 0; c < 64; ++c) for (int hw = 0; hw < 49; ++hw)
        p[(size_t)o * 3136 + c * 49 + hw] = in[ar.w4 + (size_t)(hw * 64 + c) * 512 + o];
    p += (size_t)512 * 3136; std::copy(in + ar.b4, in + ar.b4 + 512, p); p += 512;
    std::copy(in + ar.w5, in + ar.w5 + (size_t)ar.A * 512, p);
    p += (size_t)ar.A * 512; std::copy(in + ar.b5, in + ar.b5 + ar.A, p);
}

float* arena_ptr(DqnCnn* a, int which)
{
    switch (which) {
        case 0: return a->q; case 1: return a->q_tgt; case 2: return a->m; case 3: return a->v; case 4: return a->grad;
        case 5: return a->vmax;   // AdamW amsgrad: max_exp_avg_sq (nullptr otherwise)
        default: return nullptr;
    }
}

// the library's own initialiser: uniform(+-1/sqrt(fan_in)) from splitmix64 (init scheme is irrelevant to
// parity: tests inject weights through bdr_agent_set_params)
void init_reference_params(int A, int ns, uint64_t seed, std::vector<float>& ref)
{
    ref.resize(ref_param_count(A, ns));
    const size_t sizes[10] = {(size_t)2048 * ns, 32, 32768, 64, 36864, 64, (size_t)512 * 3136, 512, (size_t)A * 512, (size_t)A};
    const int fan[10] = {64 * ns, 64 * ns, 512, 512, 576, 576, 3136, 3136, 512, 512};
    uint64_t s = seed * 0x9E3779B97F4A7C15ull + 0x1234567ull;
    size_t o = 0;
    for (int t = 0; t < 10; ++t) {
        const float bound = 1.0f / std::sqrt((float)fan[t]);
        for (size_t i = 0; i < sizes[t]; ++i) {
            s += 0x9E3779B97F4A7C15ull;
            uint64_t x = s; x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31;
            ref[o + i] = ((float)(x >> 40) * (2.0f / 16777216.0f) - 1.0f) * bound;
        }
        o += sizes[t];
    }
}

}  // namespace


// ---- virtual interface -----------------------------------------------------------------------------
void DqnCnn::on_gate_timeout()
{
    // the flag is cleared by err_report; what was enqueued behind the failed gate has drained (this runs after a
    // synchronisation or with the error already visible on the host), so both queues restart from a clean epoch
    (void)hipStreamSynchronize(stream);
    if (side) (void)hipStreamSynchronize(side);
    if (aux) (void)hipStreamSynchronize(aux);
    if (comm_st) (void)hipStreamSynchronize(comm_st);
    xchg_pending_conv = xchg_pending_fc = false;
    planes_stale(0); planes_stale(1);
    if (sig) (void)hipMemset(sig, 0, 16 * sizeof(unsigned));
    if (applied_step) {
        // Updates whose parameter-writing kernels ran while the poison word was up were skipped on the device: the host's step numbers go
        // back to the last update that was applied, so Adam's bias corrections continue from the state the parameters are in
        // (the l1 / l2 pass, 95 % of the arena, records the step number it applied; opt.rs:74-83).
        // The two optimizer passes of the fused path (l1 / l2 on the weight-gradient queue, the conv segment at the end of the dX queue)
        // each record the step number they applied; a time-out that fell BETWEEN them leaves the conv segment one step behind: it keeps
        // its own step number from then on (conv_step_lag), so every segment's bias corrections match the moments it holds.
        unsigned long long applied[2] = {0, 0};
        if (hipMemcpy(applied, applied_step, sizeof applied, hipMemcpyDeviceToHost) == hipSuccess) {
            const int64_t conv_now = (int64_t)adam_step - conv_step_lag;
            if (applied[0] < adam_step || (int64_t)applied[1] < conv_now) {
                const uint64_t skipped = adam_step - std::min<uint64_t>(applied[0], adam_step);
                adam_step -= skipped;
                conv_step_lag = (int64_t)adam_step - (int64_t)std::min<uint64_t>(applied[1], (uint64_t)std::max<int64_t>(conv_now, 0));
                const uint64_t opts_back = std::min(n_opts, skipped / std::max<uint64_t>(1, cfg.n_updates_per_opt));
                n_opts -= opts_back;
                // the soft updates of the skipped opts were skipped on the device with them (k_track is poison-gated): their counter goes back too
                const uint64_t iv = std::max<uint64_t>(1, cfg.soft_update_interval);
                soft_update_counter = (soft_update_counter + iv - opts_back % iv) % iv;
                fprintf(stderr, "border_amd: %llu update(s) behind the failed gate were skipped on the device; the step counters were rolled back with them%s\n",
                        (unsigned long long)skipped, conv_step_lag ? " (the conv segment is one optimizer step behind l1 / l2 and keeps its own step number)" : "");
            }
        }
    }
    sig_epoch = 0; head_gate_enqueued = false;
    if (sched == 3) {
        fprintf(stderr, "border_amd: a cross-queue gate timed out; this agent continues with event ordering (schedule 1)\n");
        sched = 1;
    }
    if (holds_gate_token) { DqnCnn* me = this; g_gate_owner.compare_exchange_strong(me, nullptr); holds_gate_token = false; }
}

// ---- overlapped parameter exchange ------------------------------------------------------------------------------------------
// Segment 0 = the l1 / l2 parameters (95 % of the arena): final as soon as their Adam pass on the weight-gradient queue has run
// (flag SIDE of the step's epoch - published every step anyway), first read again by the NEXT update's l1 forward.
// Segment 1 = the conv parameters: final after k_reduce_adam at the end of the dX queue (a k_signal queued behind it when the
// exchange is requested), first read by the next update's conv1.  The collectives run on their own queue in that order, so
// the big segment's all-reduce overlaps the conv dX / dW tail of the step and the next forward up to l1; what stays exposed is
// the 0.3 MB conv segment.  A step that ended with a soft update (track reads the parameters) exchanges after the track.
int DqnCnn::exchange_plan(int which, ExchangeSeg* segs, int cap, hipStream_t* comm)
{
    if (which != 0 || cap < 2 || sig_epoch == 0 || getenv("BDR_NO_XCHG_OVERLAP")) return 0;
    if (effective_sched(this) != 3 || xchg_pending_conv || xchg_pending_fc) return 0;
    if (comm_state == 0) {
        comm_state = -1;
        if (hipStreamCreateWithFlags(&comm_st, hipStreamNonBlocking) == hipSuccess) {
            bool ok1 = false, ok2 = false, ok3 = false;
            if (queues_independent(this, comm_st, stream, &ok1) == BDR_OK && queues_independent(this, comm_st, side, &ok2) == BDR_OK &&
                queues_independent(this, stream, comm_st, &ok3) == BDR_OK && ok1 && ok2 && ok3) comm_state = 1;
        }
    }
    if (comm_state != 1) return 0;
    segs[0] = ExchangeSeg{ar.w4, ar.total - ar.w4};
    segs[1] = ExchangeSeg{0, ar.w4};
    planes_stale(which);   // the conv segment is about to be overwritten on the communication queue
    *comm = comm_st;
    xchg_epoch = sig_epoch;
    xchg_tracked = track_wait_pending && track_epoch == sig_epoch;   // this step ended with a soft update
    // "the dX queue has finished the step" (behind k_reduce_adam and the track, if any)
    hipLaunchKernelGGL(k_signal, dim3(1), dim3(64), 0, stream, sig, SIG_MAIN_END, xchg_epoch);
    return hipGetLastError() == hipSuccess ? 2 : 0;
}

int32_t DqnCnn::exchange_begin(int seg)
{
    const int flag = (seg == 0 && !xchg_tracked) ? SIG_SIDE : SIG_MAIN_END;
    return launch_gate(this, comm_st, flag, xchg_epoch, -1);
}

int32_t DqnCnn::exchange_end(int seg)
{
    hipLaunchKernelGGL(k_signal, dim3(1), dim3(64), 0, comm_st, sig, seg == 0 ? SIG_XFC : SIG_XCONV, xchg_epoch);
    BDR_HIP(hipGetLastError());
    if (seg == 0) xchg_pending_fc = true; else xchg_pending_conv = true;
    return BDR_OK;
}

int32_t DqnCnn::join_exchange(bool conv, bool fc)
{
    if (conv && xchg_pending_conv) { BDR_TRY(launch_gate(this, stream, SIG_XCONV, xchg_epoch, -1)); xchg_pending_conv = false; }
    if (fc && xchg_pending_fc) { BDR_TRY(launch_gate(this, stream, SIG_XFC, xchg_epoch, -1)); xchg_pending_fc = false; }
    return BDR_OK;
}

DqnCnn::~DqnCnn()
{
    (void)hipSetDevice(device);
    if (comm_st) { (void)hipStreamSynchronize(comm_st); (void)hipStreamDestroy(comm_st); }
    if (holds_gate_token) { DqnCnn* me = this; g_gate_owner.compare_exchange_strong(me, nullptr); holds_gate_token = false; }
    (void)hipStreamSynchronize(stream);
    if (side) (void)hipStreamSynchronize(side);
    free_batch_buffers(this);
    (void)hipFree(q); (void)hipFree(q_tgt); (void)hipFree(grad); (void)hipFree(m); (void)hipFree(v); (void)hipFree(vmax);
    (void)hipFree(loss); (void)hipFree(cpl[0]); (void)hipFree(cpl[1]);
    (void)hipFree(u_obs); (void)hipFree(u_next); (void)hipFree(u_act); (void)hipFree(u_rew); (void)hipFree(u_term);
    for (auto& e : ev_fork) if (e) (void)hipEventDestroy(e);
    if (ev_join) (void)hipEventDestroy(ev_join);
    if (gate_trace) {
        unsigned long long t[32] = {0};
        (void)hipMemcpy(t, gate_trace, sizeof t, hipMemcpyDeviceToHost);
        static const char* site[5] = {"side<-head", "side<-DxL1", "side<-DxC3", "-", "main<-side (join)"};
        for (int k = 0; k < 5; ++k)
            if (t[2 * k + 1]) fprintf(stderr, "gate %-18s mean wait %.2f us, longest %.1f us, over %llu gates\n", site[k], t[2 * k] / 100.0 / t[2 * k + 1], t[22 + k] / 100.0, t[2 * k + 1]);
        if (t[9]) fprintf(stderr, "the weight-gradient queue finished on average %.2f us before the dX queue reached the join\n", t[20] / 100.0 / (t[9] / 2.0));
        (void)hipFree(gate_trace);
    }
    if (sig) (void)hipFree(sig);
    (void)hipFree(applied_step); (void)hipFree(act_part); (void)hipFree(act_tickets);
    if (aux) { (void)hipStreamSynchronize(aux); stream_retire(aux); (void)hipStreamDestroy(aux); }
    if (side) { stream_retire(side); (void)hipStreamDestroy(side); }
}

int32_t DqnCnn::opt(bdr_replay* r) { return opt_inner(this, r); }

int32_t DqnCnn::apply_grads()
{
    BDR_TRY(adam_all(this));
    return after_updates(this);
}

static std::vector<NamedTensor> cnn_meta(int A, int ns);

// Agent::opt_with_record (dqn/base.rs:316-342): "loss"; with record_verbose_level >= 2 also pred_mean, reward_mean, tgt_mean,
// tgt_minus_pred_mean (update_critic :107-121), then - only from opt_with_record - qnet.param_stats() (`<var>_mean`,
// `<var>_std` of c1.weight ... l2.bias, util.rs:64-80) and ratio_best_act = n_samples_best_act / n_samples_act with both
// counters reset (:331-339).
int32_t DqnCnn::record(float* out, int cap, int* n)
{
    bdr_dqn_record rec{};
    BDR_TRY(fill_record(this, last_B, last_reward, &rec));
    std::vector<float> v = {rec.loss};
    if (rec.has_verbose) {
        v.insert(v.end(), {rec.pred_mean, rec.reward_mean, rec.tgt_mean, rec.tgt_minus_pred_mean});
        if (rec_opt) {
            std::vector<float> ref(ref_param_count(ar.A, ar.ns));
            BDR_TRY(get_params(0, ref.data(), ref.size()));
            param_stats(cnn_meta(ar.A, ar.ns), ref.data(), v);
            v.push_back(n_samples_act == 0 ? 0.f : (float)n_samples_best_act / (float)n_samples_act);
            n_samples_act = 0; n_samples_best_act = 0;
        }
    }
    for (size_t i = 0; i < v.size() && (int)i < cap; ++i) out[i] = v[i];
    *n = (int)std::min(v.size(), (size_t)cap);
    return BDR_OK;
}

void DqnCnn::record_keys(std::vector<std::string>& keys)
{
    keys = {"loss"};
    if (cfg.record_verbose_level >= 2) {
        keys.insert(keys.end(), {"pred_mean", "reward_mean", "tgt_mean", "tgt_minus_pred_mean"});
        param_stat_keys(cnn_meta(ar.A, ar.ns), keys);
        keys.push_back("ratio_best_act");
    }
}

uint64_t DqnCnn::param_count(int which) { return which == -1 ? (uint64_t)ar.A : ref_param_count(ar.A, ar.ns); }

int32_t DqnCnn::get_params(int which, float* out, uint64_t n)
{
    BDR_TRY(join_exchange(true, true));
    float* src = arena_ptr(this, which);
    BDR_REQUIRE(src, "which must be 0..4");
    BDR_REQUIRE(n == ref_param_count(ar.A, ar.ns), "parameter count mismatch (%llu vs %llu)", (unsigned long long)n,
                (unsigned long long)ref_param_count(ar.A, ar.ns));
    std::vector<float> in(ar.total);
    BDR_HIP(hipMemcpyAsync(in.data(), src, ar.total * 4, hipMemcpyDeviceToHost, stream));
    BDR_HIP(hipStreamSynchronize(stream));
    to_reference(ar, in.data(), out);
    return BDR_OK;
}

int32_t DqnCnn::set_params(int which, const float* inp, uint64_t n)
{
    BDR_TRY(join_exchange(true, true));
    float* dst = arena_ptr(this, which);
    BDR_REQUIRE(dst, "which must be 0..4");
    BDR_REQUIRE(n == ref_param_count(ar.A, ar.ns), "parameter count mismatch");
    std::vector<float> in(ar.total);
    to_internal(ar, inp, in.data());
    BDR_HIP(hipMemcpyAsync(dst, in.data(), ar.total * 4, hipMemcpyHostToDevice, stream));
    BDR_HIP(hipStreamSynchronize(stream));
    planes_stale(which);
    return BDR_OK;
}

float* DqnCnn::arena(int which, size_t* n)
{
    (void)join_exchange(true, true);   // whoever asks for the arena is about to enqueue work on it behind this queue
    planes_stale(which);               // ... possibly a write
    if (n) *n = ar.total;
    return arena_ptr(this, which);
}

static std::vector<NamedTensor> cnn_meta(int A, int ns)
{
    return {{"c1.weight", {32, (uint64_t)ns, 8, 8}}, {"c1.bias", {32}}, {"c2.weight", {64, 32, 4, 4}}, {"c2.bias", {64}},
            {"c3.weight", {64, 64, 3, 3}}, {"c3.bias", {64}}, {"l1.weight", {512, 3136}}, {"l1.bias", {512}},
            {"l2.weight", {(uint64_t)A, 512}}, {"l2.bias", {(uint64_t)A}}};
}

int32_t DqnCnn::save(const char* dir)
{
    // dqn/base.rs:348-356: qnet.pt.tch, qnet_tgt.pt.tch
    std::vector<float> ref(ref_param_count(ar.A, ar.ns));
    BDR_TRY(get_params(0, ref.data(), ref.size()));
    BDR_TRY(save_named(ckpt_save_path(this, dir, "qnet"), cnn_meta(ar.A, ar.ns), ref.data(), ref.size()));
    BDR_TRY(get_params(1, ref.data(), ref.size()));
    return save_named(ckpt_save_path(this, dir, "qnet_tgt"), cnn_meta(ar.A, ar.ns), ref.data(), ref.size());
}

int32_t DqnCnn::load(const char* dir)
{
    std::vector<float> ref(ref_param_count(ar.A, ar.ns));
    BDR_TRY(load_named(ckpt_load_path(this, dir, "qnet"), cnn_meta(ar.A, ar.ns), ref.data(), ref.size()));
    BDR_TRY(set_params(0, ref.data(), ref.size()));
    BDR_TRY(load_named(ckpt_load_path(this, dir, "qnet_tgt"), cnn_meta(ar.A, ar.ns), ref.data(), ref.size()));
    return set_params(1, ref.data(), ref.size());
}

namespace bdr {
int32_t dqn_cnn_create(const bdr_dqn_config* cfg, bdr_agent** out)
{
    BDR_REQUIRE(cfg->net.n_stack >= 1 && cfg->net.n_stack <= bdr::C1_MAX_STACK, "AtariCnnConfig::n_stack must be in [1, %d] (conv1's kernels are instantiated "
                "for these; the reference's examples use 4)", bdr::C1_MAX_STACK);
    BDR_REQUIRE(cfg->net.out_dim >= 1 && cfg->net.out_dim <= 64, "out_dim must be in [1,64]");
    DqnCnn* a = new DqnCnn();
    a->cfg = *cfg; a->device = cfg->device; a->train = cfg->train != 0;
    a->ar = make_arena(cfg->net.out_dim, cfg->net.n_stack);
    BDR_HIP(hipStreamCreateWithFlags(&a->stream, hipStreamNonBlocking));
    BDR_HIP(hipStreamCreateWithFlags(&a->side, hipStreamNonBlocking));
    for (auto& e : a->ev_fork) BDR_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventDisableSystemFence));
    BDR_HIP(hipEventCreateWithFlags(&a->ev_join, hipEventDisableTiming | hipEventDisableSystemFence));
    if (const char* e = getenv("BDR_SCHED")) a->sched = std::max(0, std::min(3, atoi(e)));
    BDR_HIP(hipMalloc((void**)&a->sig, 16 * sizeof(unsigned)));
    BDR_HIP(hipMalloc((void**)&a->applied_step, 2 * sizeof(unsigned long long)));
    BDR_HIP(hipMemsetAsync(a->applied_step, 0, 2 * sizeof(unsigned long long), a->stream));
    BDR_HIP(hipMemsetAsync(a->sig, 0, 16 * sizeof(unsigned), a->stream));   // synchronised with the parameter upload below
    if (getenv("BDR_GATE_TRACE")) {
        BDR_HIP(hipMalloc((void**)&a->gate_trace, 32 * sizeof(unsigned long long)));
        BDR_HIP(hipMemsetAsync(a->gate_trace, 0, 32 * sizeof(unsigned long long), a->stream));
    }
    if (getenv("BDR_NO_OVERLAP")) a->sched = 0;
    if (const char* e = getenv("BDR_GATE_LIMIT_MS")) a->gate_limit = (unsigned long long)std::max(1, atoi(e)) * 100000ull;
    BDR_TRY(a->err_init());
    a->kev = getenv("BDR_NO_KEV") == nullptr;
    a->side_gather = getenv("BDR_NO_SIDE_GATHER") == nullptr;
    a->split_fwd = getenv("BDR_NO_SPLIT_FWD") == nullptr;
    a->conv_b3 = arith_is_split(cfg->arithmetic, "BDR_DQN_F32_EXACT");   // bdr_dqn_config::arithmetic; the variable overrides it for A/B runs only
    if (a->conv_b3) for (auto& pl : a->cpl) BDR_HIP(hipMalloc((void**)&pl, CPL_U16 * 2));
    { const char* e = getenv("BDR_TQ"); a->three_queues = !(e && e[0] == '0'); }
    float** arenas[5] = {&a->q, &a->q_tgt, &a->grad, &a->m, &a->v};
    for (auto p : arenas) {
        BDR_TRY(alloc_f(p, a->ar.total));
        BDR_HIP(hipMemsetAsync(*p, 0, a->ar.total * 4, a->stream));
    }
    a->amsgrad = cfg->opt_kind == BDR_OPT_ADAMW && cfg->amsgrad != 0;
    if (a->amsgrad) { BDR_TRY(alloc_f(&a->vmax, a->ar.total)); BDR_HIP(hipMemsetAsync(a->vmax, 0, a->ar.total * 4, a->stream)); }
    BDR_TRY(alloc_f(&a->loss, 4));
    std::vector<float> ref, in(a->ar.total);
    init_reference_params(a->ar.A, a->ar.ns, cfg->param_seed, ref);
    to_internal(a->ar, ref.data(), in.data());
    BDR_HIP(hipMemcpyAsync(a->q, in.data(), a->ar.total * 4, hipMemcpyHostToDevice, a->stream));
    BDR_HIP(hipMemcpyAsync(a->q_tgt, in.data(), a->ar.total * 4, hipMemcpyHostToDevice, a->stream));  // DqnModel::clone
    BDR_HIP(hipStreamSynchronize(a->stream));
    BDR_TRY(ensure_batch(a, (int)cfg->batch_size));
    BDR_HIP(hipStreamCreateWithFlags(&a->aux, hipStreamNonBlocking));   // prioritized-replay tree updates
    if (a->sched == 3 && !getenv("BDR_SKIP_QUEUE_CHECK")) {   // both directions are used: the side queue waits for the dX queue, the join waits the other way
        // (BDR_SKIP_QUEUE_CHECK: tests only - lets a gate deadlock for real so that the timeout path can be exercised)
        bool ok1 = false, ok2 = false;
        BDR_TRY(queues_independent(a, a->side, a->stream, &ok1));
        BDR_TRY(queues_independent(a, a->stream, a->side, &ok2));
        BDR_TRY(queues_independent(a, a->aux, a->stream, &a->aux_gated));   // aliased: its dependency falls back to an event
        if (!(ok1 && ok2)) {
            fprintf(stderr, "border_amd: the agent's two streams share a hardware queue (GPU_MAX_HW_QUEUES?): serial backward schedule\n");
            a->sched = 0;
        }
    }
    *out = a;
    return BDR_OK;
}

int32_t dqn_cnn_update_on_batch(bdr_agent* base, uint64_t n, const void* obs, const int64_t* act, const void* next_obs,
                                const float* reward, const int8_t* term, const float* weight)
{
    DqnCnn* a = static_cast<DqnCnn*>(base);
    const size_t ob = (size_t)a->cfg.net.n_stack * 84 * 84;
    if (n > a->u_cap) {
        BDR_HIP(hipStreamSynchronize(a->stream));
        (void)hipFree(a->u_obs); (void)hipFree(a->u_next); (void)hipFree(a->u_act); (void)hipFree(a->u_rew); (void)hipFree(a->u_term);
        BDR_HIP(hipMalloc((void**)&a->u_obs, n * ob)); BDR_HIP(hipMalloc((void**)&a->u_next, n * ob));
        BDR_HIP(hipMalloc((void**)&a->u_act, n * 8)); BDR_HIP(hipMalloc((void**)&a->u_rew, n * 4));
        BDR_HIP(hipMalloc((void**)&a->u_term, round_up(n, 16)));
        a->u_cap = n;
    }
    BDR_HIP(hipMemcpyAsync(a->u_obs, obs, n * ob, hipMemcpyHostToDevice, a->stream));
    BDR_HIP(hipMemcpyAsync(a->u_next, next_obs, n * ob, hipMemcpyHostToDevice, a->stream));
    BDR_HIP(hipMemcpyAsync(a->u_act, act, n * 8, hipMemcpyHostToDevice, a->stream));
    BDR_HIP(hipMemcpyAsync(a->u_rew, reward, n * 4, hipMemcpyHostToDevice, a->stream));
    BDR_HIP(hipMemcpyAsync(a->u_term, term, n, hipMemcpyHostToDevice, a->stream));
    const float* wd = nullptr;
    if (weight) { BDR_TRY(a->td_buffer(n)); BDR_HIP(hipMemcpyAsync(a->w_stage, weight, n * 4, hipMemcpyHostToDevice, a->stream)); wd = a->w_stage; }
    const bool backward_only = a->defer_adam;      // grads_on_batch
    if (a->amsgrad) a->defer_adam = true;          // the amsgrad step is backward -> adam_all
    const int32_t st = update_critic(a, (int)n, a->u_obs, a->u_next, a->u_act, 8, a->u_rew, a->u_term, wd, nullptr);
    a->defer_adam = backward_only;
    if (st == BDR_OK && !backward_only) {
        if (a->amsgrad) BDR_TRY(adam_all(a));
        BDR_TRY(after_updates(a));
    }
    BDR_TRY(st);
    BDR_HIP(hipStreamSynchronize(a->stream));   // host buffers may be reused by the caller
    return BDR_OK;
}

}  // namespace bdr

int32_t DqnCnn::grads_on_batch(uint64_t n, const void* obs, const int64_t* act, const void* next_obs, const float* reward, const int8_t* term)
{
    defer_adam = true;
    const int32_t st = bdr::dqn_cnn_update_on_batch(this, n, obs, act, next_obs, reward, term, nullptr);
    defer_adam = false;
    return st;
}

namespace bdr {
static int getenv_once_no_act_small()
{
    static const int v = getenv("BDR_NO_ACT_SMALL") ? 1 : 0;   // diagnostic: acting calls on the training kernels
    return v;
}

// Q rows of n <= ACT_SMALL_MAX observations with the acting kernels (act_small.hpp): conv1 (staged-image form, one image per workgroup),
// conv2 / conv3 / l1 as 32 x 32 tiles x k-slices, l2 + the hand-over to the host by the workgroup that finishes l1.  Five launches, no copy
// command, no stream synchronisation.
static int32_t act_small_forward(DqnCnn* a, const uint8_t* d, int n, float* q_out)
{
    const Arena& ar = a->ar;
    hipStream_t st = a->stream;
    BDR_TRY(a->join_exchange(true, true));   // an overlapped parameter exchange of the online network may still be in flight
    if (!a->act_part) {
        BDR_HIP(hipMalloc((void**)&a->act_part, act_small_part_floats() * sizeof(float)));
        BDR_HIP(hipMalloc((void**)&a->act_tickets, ACT_SMALL_TICKETS * sizeof(unsigned)));
        BDR_HIP(hipMemsetAsync(a->act_tickets, 0, ACT_SMALL_TICKETS * sizeof(unsigned), st));
    }
    if (!a->rows_host) {
        BDR_HIP(hipHostMalloc((void**)&a->rows_host, (ROWS_PINNED_MAX_FLOATS + 16) * sizeof(float), hipHostMallocMapped));
        memset(a->rows_host, 0, (ROWS_PINNED_MAX_FLOATS + 16) * sizeof(float));
        BDR_HIP(hipHostGetDevicePointer((void**)&a->rows_dev, a->rows_host, 0));
    }
    {
        Conv1Args c{};
        c.M = n * 400; c.nz = 1; c.x[0] = d; c.w1[0] = a->q + ar.w1; c.bias[0] = a->q + ar.b1; c.out[0] = a->a1[0];
        if (ar.ns <= C1IMG_MAX_STACK) BDR_HIP(launch_conv1_bf16_img(ar.ns, 1, 16, dim3(n), st, c));   // 16 waves: an image's 13 units in one round
        else BDR_HIP(conv1_forward(ar.ns, n, st, c));
    }
    auto pad32 = [](int m) { return (m + 31) / 32 * 32; };
    ActLayerArgs l{};
    l.part = a->act_part; l.tickets = a->act_tickets; l.relu = 1;
    l.x = a->a1[0]; l.w = a->q + ar.w2; l.bias = a->q + ar.b2; l.out = a->a2[0]; l.M = n * 81; l.Mpad = pad32(l.M); l.N = 64; l.K = 512; l.KS = 4;
    BDR_HIP(launch_act_layer<2>(st, l));
    l.x = a->a2[0]; l.w = a->q + ar.w3; l.bias = a->q + ar.b3; l.out = a->a3[0]; l.M = n * 49; l.Mpad = pad32(l.M); l.N = 64; l.K = 576; l.KS = 3;
    BDR_HIP(launch_act_layer<3>(st, l));
    const unsigned seq = ++a->rows_seq;
    l.x = a->a3[0]; l.w = a->q + ar.w4; l.bias = a->q + ar.b4; l.out = a->h1[0]; l.M = n; l.Mpad = 32; l.N = 512; l.K = 3136; l.KS = 7;
    l.head = ActLayerArgs::Head{a->q + ar.w5, a->q + ar.b5, a->qv[0], n, ar.A, a->rows_dev + 16, reinterpret_cast<unsigned*>(a->rows_dev), seq,
                                (const unsigned*)a->dev_err, (int)bdr_agent::ERR_WORDS};
    BDR_HIP(launch_act_layer<0>(st, l));
    return a->rows_wait(seq, q_out, (size_t)n * ar.A);
}

int32_t dqn_cnn_qvalues(bdr_agent* base, uint64_t n, const void* obs, float* q_out)
{
    DqnCnn* a = static_cast<DqnCnn*>(base);
    BDR_TRY(ensure_batch(a, (int)n));
    const size_t ob = (size_t)a->cfg.net.n_stack * 84 * 84;
    const uint8_t* d = static_cast<const uint8_t*>(obs);
    const bool small = n <= (uint64_t)ACT_SMALL_MAX && !a->prof && getenv_once_no_act_small() == 0;
    if (small && !a->obs_rows_on_device) {
        // host rows of an acting call: into pinned memory the device reads in place (conv1's image staging fetches the 28 KB over PCIe while
        // the weights are split) - a host -> device copy command from pageable memory costs ~20 us of host time for these few rows
        BDR_TRY(a->host_rows_pinned(obs, n * ob, &d));
    } else if (!a->obs_in_place(ob)) {   // host rows, or device rows with a stride: into the agent's contiguous staging buffer
        uint8_t* stage = nullptr;
        BDR_TRY(a->act_buffer(n * ob, (void**)&stage));
        BDR_TRY(a->stage_obs(stage, obs, ob, n, a->stream));
        d = stage;
    }
    if (small) {   // acting-sized call: act_small.hpp
        const int32_t st = act_small_forward(a, d, (int)n, q_out);
        a->slot_cursor = 0;
        return st;
    }
    NetInst inst[1] = {{d, a->q, 0}};
    int32_t st = forward(a, inst, 1, (int)n);
    if (st == BDR_OK) st = a->rows_to_host(a->qv[0], q_out, n * (size_t)a->ar.A);   // (pinned path for acting-sized results: agent_base.hpp)
    a->slot_cursor = 0;
    return st;
}

int32_t dqn_cnn_probe(bdr_agent* base, int32_t what, float* out, uint64_t n)
{
    DqnCnn* a = static_cast<DqnCnn*>(base);
    const float* src = nullptr;
    switch (what) {
        case 0: src = a->qv[0]; break;
        case 1: src = a->qv[1]; break;
        case 2: src = a->pred; break;
        case 3: src = a->tgt; break;
        case 4: src = a->loss; break;
        case 5: src = a->a1[0]; break;   // the online network's activations on `obs`, position-major: [B][20*20][32], [B][9*9][64], [B][7*7][64]
        case 6: src = a->a2[0]; break;
        case 7: src = a->a3[0]; break;
        default: return fail(BDR_ERR_INVALID, "unknown probe %d", what);
    }
    BDR_REQUIRE(src, "nothing to probe yet (no update has run)");
    BDR_HIP(hipMemcpyAsync(out, src, n * 4, hipMemcpyDeviceToHost, a->stream));
    BDR_HIP(hipStreamSynchronize(a->stream));
    return BDR_OK;
}
}  // namespace bdr
