// One-workgroup DQN step for Q-networks that fit a single CU (BASELINE config 1: CartPole-shaped Mlp[64,64], batch 32).
//
// Dqn::update_critic (border-tch-agent/src/dqn/base.rs:60-160) + Adam (opt.rs:74-83) + track (util.rs:31-45) for an Mlp
// Q-network (mlp/base.rs:13-41) in ONE launch: at this size the generic path is 15 launches of 2-4 us whose cost is launch latency
// (a dependent kernel costs >= 2.4 us on the device, 3-5 us from the host; tools/probes/graph_probe.hip), not arithmetic - a
// 32 x 64 x 64 layer is four 32x32 MFMA blocks.  One 512-thread workgroup walks the phases of the step with a barrier between
// them; every GEMM of the step (forward of the online / target / double-DQN instance, dW, dX) is a set of 32x32 output blocks
// spread over the 8 waves, computed with v_mfma_f32_32x32x2_f32 straight from the (L2-resident, zero-padded) operands - the
// same exact-f32, k-ordered accumulation as the tiled kernels of dense.hpp, same padded layouts, same buffers (so probes,
// records, get_params("grad") and the synchronous-DP split step see exactly what the generic path leaves behind).
#pragma once
#include "agent_base.hpp"

namespace {
using namespace bdr;

constexpr int MF_MAXL = 4;       // layers
constexpr int MF_MAXZ = 3;       // network instances: online(obs), target(next_obs), online(next_obs) for double DQN

#ifdef MF_TRACE   // tools/probes only: phase timestamps (100 MHz wall clock) of thread 0
#define MF_TP(k) do { if (threadIdx.x == 0 && a.trace) { a.trace[k] = wall_clock64(); if ((k) == 0) a.trace[12] = clock64(); if ((k) == 11) a.trace[13] = clock64(); } } while (0)
#else
#define MF_TP(k) do { } while (0)
#endif

struct MlpFusedArgs {
    unsigned long long* trace;
    int L, nz, B, A, in_dim;
    int Kp[MF_MAXL], Np[MF_MAXL], relu[MF_MAXL];
    size_t w[MF_MAXL], b[MF_MAXL];                 // offsets into a parameter arena
    const float* params[MF_MAXZ];                  // arena of each instance (online / target / online)
    const float* in_rows[MF_MAXZ];                 // [B][in_dim] f32 rows of each instance's input
    float* x_in[MF_MAXZ];                          // packed inputs [B][Kp0]
    float* act[MF_MAXZ][MF_MAXL];                  // layer outputs [B][Np_l]
    float* dy[MF_MAXL];                            // gradient w.r.t. layer outputs [B][Np_l]
    // TD step (see k_td_dense)
    const uint8_t* actions; int act_bytes; const float* reward; const int8_t* term;
    float *pred, *tgt, *loss_row, *loss;
    float gamma; int loss_kind, double_dqn;
    const float* weight; float* td_abs; int has_clip; float clip_min, clip_max;
    unsigned* err;
    // optimizer
    float *q, *grad, *m, *v, *q_tgt; size_t total;
    AdamScalars adam; int do_adam, do_track; float tau, omt;
    // do_gather: the kernel also IS the replay buffer's sample of this step (replay_sample_plan): it draws the B indices of the
    // buffer's StdRng stream and copies the rows into the buffer's batch arrays - which in_rows / actions / reward / term point at
    int do_gather; GatherArgs g;
    // LDS-resident variant (k_dqn_mlp_step_lds): offsets in floats into the dynamic LDS block, rows padded by 4 floats
    int lds_x0, lds_act0[MF_MAXL], lds_pp[MF_MAXZ][2], lds_dy[MF_MAXL], lds_floats;
    int in_rows_l[MF_MAXL];   // logical input width of every layer: rows >= it of W_l are structural zeros (padding)
};

// LDS plan of k_dqn_mlp_step_lds for this problem; returns the bytes needed (0: does not fit, use k_dqn_mlp_step)
inline size_t mf_lds_plan(MlpFusedArgs& a)
{
    int o = 0, maxw = a.Kp[0];
    for (int l = 0; l < a.L; ++l) maxw = std::max(maxw, a.Np[l]);
    auto take = [&](int width) { const int at = o; o += a.B * (width + 4); return at; };
    a.lds_x0 = take(a.Kp[0]);
    for (int l = 0; l < a.L; ++l) a.lds_act0[l] = take(a.Np[l]);
    for (int z = 1; z < a.nz; ++z) { a.lds_pp[z][0] = take(maxw); a.lds_pp[z][1] = take(maxw); }
    for (int l = 0; l < a.L; ++l) a.lds_dy[l] = take(a.Np[l]);
    a.lds_floats = o;
    const size_t bytes = (size_t)o * 4;
    return bytes <= 150 * 1024 ? bytes : 0;
}

// one 32x32 block of C = A * B, K % 32 == 0.  Operand fetchers work on QUADS of the reduction index: a4(i, kq) returns
// A[i][kq .. kq+3], b4(kq, j) returns B[kq .. kq+3][j].  The k-slot an element lands in is free as long as A and B agree
// (igemm.hpp), so lane half h of MFMA group u takes k = k0 + 8u + 4h + s: an operand that is contiguous along the reduction
// index is ONE 16-byte load per group instead of four strided 4-byte loads (a row-per-lane scalar load touches 64 cache lines
// per wave instruction - the first version of this kernel spent 60 us on that).  Two chunks (16 k = 8 MFMAs each) are in
// flight: the next chunk's loads are issued before this chunk's MFMAs.
template <bool SHORT_K = false, class FA, class FB>
__device__ __forceinline__ f32x16 mf_block(int K, int lane, FA&& a4, FB&& b4)
{
    const int i = lane & 31, h = lane >> 5;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    f32x4 a0[2], b0[2], a1[2], b1[2];
    auto load = [&](int k0, f32x4 (&av)[2], f32x4 (&bv)[2]) {
#pragma unroll
        for (int u = 0; u < 2; ++u) { av[u] = a4(i, k0 + 8 * u + 4 * h); bv[u] = b4(k0 + 8 * u + 4 * h, i); }
    };
    auto mma = [&](const f32x4 (&av)[2], const f32x4 (&bv)[2]) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][q], bv[u][q], acc, 0, 0, 0);
    };
    if (SHORT_K && K <= 64) {   // short reductions (the 64-wide layers of the CartPole net): every operand load before the first MFMA - one round trip
        f32x4 a2[2], b2[2], a3[2], b3[2];
        load(0, a0, b0); load(16, a1, b1);
        if (K > 32) { load(32, a2, b2); load(48, a3, b3); }
        mma(a0, b0); mma(a1, b1);
        if (K > 32) { mma(a2, b2); mma(a3, b3); }
        return acc;
    }
    load(0, a0, b0);
    for (int k0 = 0; k0 < K; k0 += 32) {
        load(k0 + 16, a1, b1);                       // K % 32 == 0: always inside
        mma(a0, b0);
        if (k0 + 32 < K) load(k0 + 32, a0, b0);
        mma(a1, b1);
    }
    return acc;
}
// acc[r] is element (row(r), col = lane & 31) of the block
__device__ __forceinline__ int mf_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// mf_block with the B operand of a short reduction (K <= 64) already in registers: bq(kq) returns B[kq .. kq+3][lane & 31] and is
// called by mf_prefetch_b long before the A operand exists (the weights of the NEXT phase are fetched while this phase computes;
// the A operand then comes out of LDS), mf_block_pre consumes it in mf_block's k order - same bits.
template <class FB>
__device__ __forceinline__ void mf_prefetch_b(int K, int lane, f32x4 (&bv)[8], FB&& bq)
{
    const int h = lane >> 5;
#pragma unroll
    for (int c = 0; c < 4; ++c)
        if (16 * c < K) {
#pragma unroll
            for (int u = 0; u < 2; ++u) bv[c * 2 + u] = bq(16 * c + 8 * u + 4 * h);
        }
}
template <class FA>
__device__ __forceinline__ f32x16 mf_block_pre(int K, int lane, FA&& a4, const f32x4 (&bv)[8])
{
    const int i = lane & 31, h = lane >> 5;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    f32x4 av[8];
#pragma unroll
    for (int c = 0; c < 4; ++c)
        if (16 * c < K) {
#pragma unroll
            for (int u = 0; u < 2; ++u) av[c * 2 + u] = a4(i, 16 * c + 8 * u + 4 * h);
        }
#pragma unroll
    for (int c = 0; c < 4; ++c)
        if (16 * c < K) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c * 2 + u][q], bv[c * 2 + u][q], acc, 0, 0, 0);
        }
    return acc;
}

__global__ __launch_bounds__(512) void k_dqn_mlp_step(MlpFusedArgs a)
{
    __shared__ float red[512];
    // (wave index as a scalar: block -> (instance, tile) maps and the kernel-argument pointers they select stay in SGPRs; with a
    //  per-lane `wave` the compiler re-loaded a.act[z][l] with a vector load + vmcnt(0) in each of the 16 predicated stores of an epilogue)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int B = a.B, RB = (B + 31) / 32, L = a.L;
    MF_TP(0);
    // ---- sample: what k_gather does for a batch (replay.hip), rows of a few words each
    if (a.do_gather) {
        __shared__ uint64_t s_row[128];
        if (tid < B) {
            const uint64_t row = (uint64_t)chacha12_word(a.g.key, a.g.word_pos + tid) % a.g.size;   // (StdRng::next_u32() as usize) % size
            s_row[tid] = row; a.g.ixs[tid] = row;
        }
        __syncthreads();
        const int ow = (int)(a.g.obs_bytes / 4), aw = (int)a.g.act_bytes;   // obs rows are f32 words here; actions are copied as bytes
        for (int e = tid; e < B * ow; e += 512) {
            const int sidx = e / ow, w = e % ow;
            const uint8_t* rec = a.g.ring + s_row[sidx] * a.g.stride;
            reinterpret_cast<uint32_t*>(a.g.b_obs)[e] = reinterpret_cast<const uint32_t*>(rec)[w];
            reinterpret_cast<uint32_t*>(a.g.b_next)[e] = reinterpret_cast<const uint32_t*>(rec + a.g.next_off)[w];
        }
        for (int e = tid; e < B * aw; e += 512) {
            const int sidx = e / aw, w = e % aw;
            a.g.b_act[e] = a.g.ring[s_row[sidx] * a.g.stride + a.g.act_off + w];
        }
        if (tid < B) {
            const uint8_t* rec = a.g.ring + s_row[tid] * a.g.stride;
            a.g.b_reward[tid] = *reinterpret_cast<const float*>(rec + a.g.tail_off);
            a.g.b_term[tid] = *reinterpret_cast<const int8_t*>(rec + a.g.tail_off + 4);
            a.g.b_trunc[tid] = *reinterpret_cast<const int8_t*>(rec + a.g.tail_off + 5);
        }
        __syncthreads();
    }
    // ---- phase 0: pack the input rows into the zero-padded [B][Kp0] matrices
    for (int z = 0; z < a.nz; ++z)
        for (int e = tid; e < B * a.Kp[0]; e += 512) {
            const int r = e / a.Kp[0], c = e % a.Kp[0];
            a.x_in[z][e] = c < a.in_dim ? a.in_rows[z][(size_t)r * a.in_dim + c] : 0.f;
        }
    __syncthreads();
    MF_TP(1);
    // ---- forward, layer by layer, every instance in the same phase
    for (int l = 0; l < L; ++l) {
        const int Kp = a.Kp[l], Np = a.Np[l], NB = Np / 32, nblk = a.nz * RB * NB;
        for (int blk = wave; blk < nblk; blk += 8) {
            const int z = blk / (RB * NB), rb = (blk / NB) % RB, nb = blk % NB;
            const float* x = l == 0 ? a.x_in[z] : a.act[z][l - 1];
            const int ldx = l == 0 ? a.Kp[0] : a.Np[l - 1];
            const float* w = a.params[z] + a.w[l];
            const float* bias = a.params[z] + a.b[l];
            const f32x16 acc = mf_block<true>(Kp, lane,
                                        [&](int i, int kq) {
                                            const int r = min(rb * 32 + i, B - 1);   // rows >= B alias the last row (never stored)
                                            return *reinterpret_cast<const f32x4*>(x + (size_t)r * ldx + kq);
                                        },
                                        [&](int kq, int j) {
                                            const float* p = w + (size_t)kq * Np + nb * 32 + j;
                                            return f32x4{p[0], p[Np], p[2 * Np], p[3 * Np]};
                                        });
            const int col = nb * 32 + (lane & 31);
            const float bv = bias[col];
            // Full row blocks store without per-element predicates: 16 predicated blocks make the compiler wait for vmcnt(0) in
            // each of them - i.e. for the previous block's STORE (vmcnt counts stores) - 16 memory round trips in a row, 3.5 us.
            float* const outp = a.act[z][l];
            const bool full = rb * 32 + 32 <= B;   // wave-uniform
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rb * 32 + mf_row(r, lane);
                float v = acc[r] + bv;
                if (a.relu[l]) v = v > 0.f ? v : 0.f;
                if (full) outp[(size_t)row * Np + col] = v;
                else if (row < B) outp[(size_t)row * Np + col] = v;
            }
        }
        __syncthreads();
        MF_TP(2 + l);
    }
    // ---- TD step of every row (dqn/base.rs:71-74, 91-105, 123-152): one wave per row, dL/dQ as a dense row
    {
        const int ld = a.Np[L - 1];
        const float* q_on = a.act[0][L - 1];
        const float* q_tg = a.act[1][L - 1];
        const float* sel = a.double_dqn ? a.act[2][L - 1] : q_tg;
        float lsum = 0.f;
        // 16 lanes per row (A <= 64: four actions per lane): 32 rows per pass, so the dependent loads (action -> Q(s,a),
        // argmax -> target Q) cost three round trips per PASS instead of three per row
        const int sub = tid & 15;
        for (int row = tid >> 4; row < B; row += 32) {
            long long act = *reinterpret_cast<const long long*>(a.actions + (size_t)row * a.act_bytes);
            if (act < 0 || act >= a.A) {
                if (sub == 0 && a.err) atomicOr(a.err + bdr_agent::ERR_ACTION, 1u);
                act = act < 0 ? 0 : a.A - 1;
            }
            float v = -INFINITY;
            int idx = 0x7fffffff;
#pragma unroll
            for (int q = 0; q < 4; ++q) {   // first maximum (at::argmax): ascending index within the lane, ties keep the earlier
                const int c = sub + 16 * q;
                const float sv = c < a.A ? sel[(size_t)row * ld + c] : -INFINITY;
                if (sv > v || (sv == v && c < idx)) { v = sv; idx = c; }
            }
#pragma unroll
            for (int off = 8; off > 0; off >>= 1) {
                const float ov = __shfl_xor(v, off);
                const int oi = __shfl_xor(idx, off);
                if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
            }
            const float qn = q_tg[(size_t)row * ld + idx];
            const float pred = q_on[(size_t)row * ld + act];
            const float tgt = td_target(a.reward[row], (float)(1 - (int)a.term[row]), a.gamma, qn);
            const TdLossIn li{a.loss_kind, a.weight != nullptr, a.weight ? a.weight[row] : 1.f, a.has_clip, a.clip_min, a.clip_max};
            float lossb, td;
            const float dl = td_loss_row(pred, tgt, li, lossb, td);
            float dq = dl / (float)B;
            if (a.relu[L - 1] && !(pred > 0.f)) dq = 0.f;   // MlpConfig::activation_out (mlp/base.rs:36): Q = relu(z), dL/dz = dL/dQ * [Q > 0]
            if (sub == 0) { a.pred[row] = pred; a.tgt[row] = tgt; a.loss_row[row] = lossb; if (a.td_abs) a.td_abs[row] = td; }
            for (int c = sub; c < ld; c += 16) a.dy[L - 1][(size_t)row * ld + c] = c == act ? dq : 0.f;
        }
        __syncthreads();
        // loss = mean(loss_row): the fixed-order tree of k_mean_rows
        for (int b = tid; b < B; b += 512) lsum += a.loss_row[b];
        red[tid] = lsum;
        __syncthreads();
        for (int w = 256; w > 0; w >>= 1) { if (tid < w) red[tid] += red[tid + w]; __syncthreads(); }
        if (tid == 0) a.loss[0] = red[0] / (float)B;
    }
    MF_TP(6);
    // ---- backward: dW_l, db_l and dX_l (masked by the ReLU of the producing layer) in one phase per layer
    for (int l = L - 1; l >= 0; --l) {
        const int Kp = a.Kp[l], Np = a.Np[l], KB = Kp / 32, NB = Np / 32;
        const float* x = l == 0 ? a.x_in[0] : a.act[0][l - 1];
        const int ldx = l == 0 ? a.Kp[0] : a.Np[l - 1];
        const float* dy = a.dy[l];
        const float* w = a.q + a.w[l];
        float* gw = a.grad + a.w[l];
        float* gb = a.grad + a.b[l];
        const int Bp = RB * 32;
        const int n_dw = KB * NB, n_dx = l > 0 ? RB * KB : 0;
        for (int blk = wave; blk < n_dw + n_dx; blk += 8) {
            if (blk < n_dw) {   // dW[k][n] = sum_b x[b][k] dy[b][n]
                const int kb = blk / NB, nb = blk % NB;
                const f32x16 acc = mf_block(Bp, lane,
                                            [&](int i, int bq) {
                                                f32x4 v;
#pragma unroll
                                                for (int q = 0; q < 4; ++q) v[q] = bq + q < B ? x[(size_t)(bq + q) * ldx + kb * 32 + i] : 0.f;
                                                return v;
                                            },
                                            [&](int bq, int j) {
                                                f32x4 v;
#pragma unroll
                                                for (int q = 0; q < 4; ++q) v[q] = bq + q < B ? dy[(size_t)(bq + q) * Np + nb * 32 + j] : 0.f;
                                                return v;
                                            });
                const int col = nb * 32 + (lane & 31);
#pragma unroll
                for (int r = 0; r < 16; ++r) gw[(size_t)(kb * 32 + mf_row(r, lane)) * Np + col] = acc[r];
            } else {            // dX[b][k] = relu'(x[b][k]) * sum_n dy[b][n] W[k][n]
                const int q = blk - n_dw, rb = q / KB, kb = q % KB;
                const f32x16 acc = mf_block<true>(Np, lane,
                                            [&](int i, int nq) {
                                                const int r = min(rb * 32 + i, B - 1);
                                                return *reinterpret_cast<const f32x4*>(dy + (size_t)r * Np + nq);
                                            },
                                            [&](int nq, int j) { return *reinterpret_cast<const f32x4*>(w + (size_t)(kb * 32 + j) * Np + nq); });
                const int col = kb * 32 + (lane & 31);
                float* const dxp = a.dy[l - 1];
                if (rb * 32 + 32 <= B) {   // full row block: mask loads and stores in straight-line code (see the forward epilogue)
                    float mk[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) mk[r] = x[(size_t)(rb * 32 + mf_row(r, lane)) * ldx + col];
#pragma unroll
                    for (int r = 0; r < 16; ++r) dxp[(size_t)(rb * 32 + mf_row(r, lane)) * Kp + col] = mk[r] > 0.f ? acc[r] : 0.f;
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = rb * 32 + mf_row(r, lane);
                        if (row < B) dxp[(size_t)row * Kp + col] = x[(size_t)row * ldx + col] > 0.f ? acc[r] : 0.f;
                    }
                }
            }
        }
        for (int n = tid; n < Np; n += 512) {   // db[n] = sum_b dy[b][n]; 8 loads in flight (a serial loop is one L2 round trip per row)
            float s = 0.f;
            for (int b0 = 0; b0 < B; b0 += 8) {
                float t[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) t[u] = b0 + u < B ? dy[(size_t)(b0 + u) * Np + n] : 0.f;
#pragma unroll
                for (int u = 0; u < 8; ++u) s += t[u];
            }
            gb[n] = s;
        }
        __syncthreads();
        MF_TP(7 + (L - 1 - l));
    }
    // ---- Adam (libtorch Adam::step, adam_element) and the soft update (track) in the same pass over the arena
    if (a.do_adam || a.do_track) {
        // 16-byte vectors, two per thread in flight: the arena is ~13 k floats, a scalar loop would be 25 dependent round trips
        const size_t n4 = a.total / 4;   // (arena segments are multiples of 64 floats)
        constexpr int V = 4;             // vectors per thread in flight (x up to 5 arrays)
        for (size_t e0 = tid; e0 < n4; e0 += (size_t)512 * V) {
            f32x4 p[V], g[V], mm[V], vv[V], t[V];
            bool ok[V];
#pragma unroll
            for (int u = 0; u < V; ++u) {
                const size_t e = e0 + (size_t)u * 512;
                ok[u] = e < n4;
                const size_t ec = ok[u] ? e : 0;
                p[u] = reinterpret_cast<const f32x4*>(a.q)[ec];
                if (a.do_adam) { g[u] = reinterpret_cast<const f32x4*>(a.grad)[ec]; mm[u] = reinterpret_cast<const f32x4*>(a.m)[ec]; vv[u] = reinterpret_cast<const f32x4*>(a.v)[ec]; }
                if (a.do_track) t[u] = reinterpret_cast<const f32x4*>(a.q_tgt)[ec];
            }
#pragma unroll
            for (int u = 0; u < V; ++u) {
                if (!ok[u]) continue;
                const size_t e = e0 + (size_t)u * 512;
                if (a.do_adam) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) { float pe = p[u][j], me = mm[u][j], ve = vv[u][j]; adam_element(pe, g[u][j], me, ve, a.adam); p[u][j] = pe; mm[u][j] = me; vv[u][j] = ve; }
                    reinterpret_cast<f32x4*>(a.q)[e] = p[u]; reinterpret_cast<f32x4*>(a.m)[e] = mm[u]; reinterpret_cast<f32x4*>(a.v)[e] = vv[u];
                }
                if (a.do_track) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) t[u][j] = track_element(p[u][j], t[u][j], a.tau, a.omt);
                    reinterpret_cast<f32x4*>(a.q_tgt)[e] = t[u];
                }
            }
        }
    }
    MF_TP(11);
}


// ---- the same step with every activation and gradient matrix resident in LDS ---------------------------
// In k_dqn_mlp_step a phase hands its result to the next through global memory: a store that must be acknowledged (the
// workgroup barrier is a release fence: s_waitcnt vmcnt(0)) and a load that goes back to L2 - about 3 us of every 4.3 us
// phase (tools/probes/mlp_fused_probe.hip).  Here the matrices the phases exchange live in LDS (mf_lds_plan: the online
// network's input / activations, ping-pong buffers for the target and double-DQN instances, the gradients), the barrier
// between phases orders LDS only (s_waitcnt lgkmcnt(0); s_barrier - global stores keep draining in the background), and
// the same values are still stored to the global buffers the generic path fills (probes, records, synchronous-DP split
// step).  Same MFMA blocks, same k order, same bits as k_dqn_mlp_step.
__device__ __forceinline__ void mf_lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__global__ __launch_bounds__(512) void k_dqn_mlp_step_lds(MlpFusedArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ float red[512];
    __shared__ uint64_t s_row[128];
    __shared__ int s_act[128];
    __shared__ float s_rew[128], s_nd[128], s_loss[128];
    // (wave index as a scalar: block -> (instance, tile) maps and the kernel-argument pointers they select stay in SGPRs; with a
    //  per-lane `wave` the compiler re-loaded a.act[z][l] with a vector load + vmcnt(0) in each of the 16 predicated stores of an epilogue)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int B = a.B, RB = (B + 31) / 32, L = a.L;
    MF_TP(0);
    // ---- sample (see k_dqn_mlp_step) + the per-row scalars of the TD step
    if (a.do_gather) {
        if (tid < B) {
            const uint64_t row = (uint64_t)chacha12_word(a.g.key, a.g.word_pos + tid) % a.g.size;
            s_row[tid] = row; a.g.ixs[tid] = row;
        }
        mf_lds_sync();
        const int ow = (int)(a.g.obs_bytes / 4), aw = (int)a.g.act_bytes;
        for (int e = tid; e < B * ow; e += 512) {
            const int sidx = e / ow, w = e % ow;
            const uint8_t* rec = a.g.ring + s_row[sidx] * a.g.stride;
            reinterpret_cast<uint32_t*>(a.g.b_obs)[e] = reinterpret_cast<const uint32_t*>(rec)[w];
            reinterpret_cast<uint32_t*>(a.g.b_next)[e] = reinterpret_cast<const uint32_t*>(rec + a.g.next_off)[w];
        }
        for (int e = tid; e < B * aw; e += 512) {
            const int sidx = e / aw, w = e % aw;
            a.g.b_act[e] = a.g.ring[s_row[sidx] * a.g.stride + a.g.act_off + w];
        }
    }
    if (tid < B) {
        long long act; float rew; int term;
        if (a.do_gather) {
            const uint8_t* rec = a.g.ring + s_row[tid] * a.g.stride;
            act = *reinterpret_cast<const long long*>(rec + a.g.act_off);          // (record fields are 8/16-byte aligned, replay.hip)
            rew = *reinterpret_cast<const float*>(rec + a.g.tail_off);
            term = *reinterpret_cast<const int8_t*>(rec + a.g.tail_off + 4);
            a.g.b_reward[tid] = rew; a.g.b_term[tid] = (int8_t)term;
            a.g.b_trunc[tid] = *reinterpret_cast<const int8_t*>(rec + a.g.tail_off + 5);
        } else {
            act = *reinterpret_cast<const long long*>(a.actions + (size_t)tid * a.act_bytes);
            rew = a.reward[tid]; term = a.term[tid];
        }
        if (act < 0 || act >= a.A) {
            if (a.err) atomicOr(a.err + bdr_agent::ERR_ACTION, 1u);
            act = act < 0 ? 0 : a.A - 1;
        }
        s_act[tid] = (int)act; s_rew[tid] = rew; s_nd[tid] = (float)(1 - term);
    }
    // weights of a phase's first block per wave, fetched one phase ahead (short reductions only)
    f32x4 bw[2][8]; float bb[2] = {0.f, 0.f};
    auto fwd_pre_ok = [&](int l) { return a.Kp[l] <= 64 && wave < a.nz * RB * (a.Np[l] / 32); };
    auto fwd_prefetch = [&](int l, f32x4 (&bv)[8], float& bias) {
        const int Np = a.Np[l], NB = Np / 32;
        const int z = wave / (RB * NB), nb = wave % NB;
        const float* w = a.params[z] + a.w[l] + nb * 32 + (lane & 31);
        mf_prefetch_b(a.Kp[l], lane, bv, [&](int kq) { const float* p = w + (size_t)kq * Np; return f32x4{p[0], p[Np], p[2 * Np], p[3 * Np]}; });
        bias = a.params[z][a.b[l] + nb * 32 + (lane & 31)];
    };
    if (fwd_pre_ok(0)) fwd_prefetch(0, bw[0], bb[0]);
    // ---- pack the input rows into the zero-padded matrices (LDS + the global copies)
    {
        const int Kp0 = a.Kp[0], ldx = Kp0 + 4;
        for (int z = 0; z < a.nz; ++z) {
            float* xs = lds + (z == 0 ? a.lds_x0 : a.lds_pp[z][0]);
            for (int e = tid; e < B * Kp0; e += 512) {
                const int r = e / Kp0, c = e % Kp0;
                float v = 0.f;
                if (c < a.in_dim) {
                    if (a.do_gather) v = *reinterpret_cast<const float*>(a.g.ring + s_row[r] * a.g.stride + (z == 0 ? 0 : a.g.next_off) + (size_t)c * 4);
                    else v = a.in_rows[z][(size_t)r * a.in_dim + c];
                }
                xs[r * ldx + c] = v;
                a.x_in[z][e] = v;
            }
        }
    }
    mf_lds_sync();
    MF_TP(1);
    // ---- forward, layer by layer, every instance in the same phase
#pragma unroll
    for (int l = 0; l < MF_MAXL; ++l) {
        if (l >= L) break;
        const int Kp = a.Kp[l], Np = a.Np[l], NB = Np / 32, nblk = a.nz * RB * NB;
        const int ldx = Kp + 4, ldo = Np + 4;
        const bool pre = fwd_pre_ok(l);
        if (l + 1 < L && fwd_pre_ok(l + 1)) fwd_prefetch(l + 1, bw[(l + 1) & 1], bb[(l + 1) & 1]);   // in flight during this layer's MFMAs
        for (int blk = wave; blk < nblk; blk += 8) {
            const int z = blk / (RB * NB), rb = (blk / NB) % RB, nb = blk % NB;
            const int xoff = z == 0 ? (l == 0 ? a.lds_x0 : a.lds_act0[l - 1]) : a.lds_pp[z][l & 1];
            const int ooff = z == 0 ? a.lds_act0[l] : a.lds_pp[z][(l + 1) & 1];
            auto a4 = [&](int i, int kq) {
                const int r = min(rb * 32 + i, B - 1);   // rows >= B alias the last row (never stored)
                return *reinterpret_cast<const f32x4*>(lds + xoff + r * ldx + kq);
            };
            const int col = nb * 32 + (lane & 31);
            f32x16 acc; float bv;
            if (pre && blk == wave) { acc = mf_block_pre(Kp, lane, a4, bw[l & 1]); bv = bb[l & 1]; }
            else {
                const float* w = a.params[z] + a.w[l];
                bv = a.params[z][a.b[l] + col];
                acc = mf_block<true>(Kp, lane, a4, [&](int kq, int j) {
                    const float* p = w + (size_t)kq * Np + nb * 32 + j;
                    return f32x4{p[0], p[Np], p[2 * Np], p[3 * Np]};
                });
            }
            float* const outp = a.act[z][l];
            if (rb * 32 + 32 <= B) {   // full row block: no per-element predicates (see k_dqn_mlp_step)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = rb * 32 + mf_row(r, lane);
                    float v = acc[r] + bv;
                    if (a.relu[l]) v = v > 0.f ? v : 0.f;
                    lds[ooff + row * ldo + col] = v;
                    outp[(size_t)row * Np + col] = v;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = rb * 32 + mf_row(r, lane);
                    if (row < B) {
                        float v = acc[r] + bv;
                        if (a.relu[l]) v = v > 0.f ? v : 0.f;
                        lds[ooff + row * ldo + col] = v;
                        outp[(size_t)row * Np + col] = v;
                    }
                }
            }
        }
        mf_lds_sync();
        MF_TP(2 + l);
    }
    // dX of the last layer: its weights are fetched while the TD step runs
    auto dx_pre_ok = [&](int l) {   // the wave's first block of backward phase l is a dX block with a short reduction
        if (l <= 0 || l >= L || a.Np[l] > 64) return false;
        const int n_dw = (a.Kp[l] / 32) * (a.Np[l] / 32), n_dx = RB * (a.Kp[l] / 32);
        return wave >= n_dw && wave < n_dw + n_dx;
    };
    auto dx_prefetch = [&](int l, f32x4 (&bv)[8]) {
        const int Np = a.Np[l], KB = a.Kp[l] / 32, n_dw = KB * (Np / 32);
        const int kb = (wave - n_dw) % KB;
        const float* w = a.q + a.w[l] + (size_t)(kb * 32 + (lane & 31)) * Np;
        mf_prefetch_b(Np, lane, bv, [&](int nq) { return *reinterpret_cast<const f32x4*>(w + nq); });
    };
    if (dx_pre_ok(L - 1)) dx_prefetch(L - 1, bw[0]);
    // ---- TD step of every row (dqn/base.rs:71-74, 91-105, 123-152), 16 lanes per row
    {
        const int ld = a.Np[L - 1], lds_ld = ld + 4;
        const int q_on = a.lds_act0[L - 1], q_tg = a.lds_pp[1][L & 1], sel = a.double_dqn ? a.lds_pp[2][L & 1] : q_tg;
        const int sub = tid & 15;
        for (int row = tid >> 4; row < B; row += 32) {
            const int act = s_act[row];
            float v = -INFINITY;
            int idx = 0x7fffffff;
#pragma unroll
            for (int q = 0; q < 4; ++q) {   // first maximum (at::argmax)
                const int c = sub + 16 * q;
                const float sv = c < a.A ? lds[sel + row * lds_ld + c] : -INFINITY;
                if (sv > v || (sv == v && c < idx)) { v = sv; idx = c; }
            }
#pragma unroll
            for (int off = 8; off > 0; off >>= 1) {
                const float ov = __shfl_xor(v, off);
                const int oi = __shfl_xor(idx, off);
                if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
            }
            const float qn = lds[q_tg + row * lds_ld + idx];
            const float pred = lds[q_on + row * lds_ld + act];
            const float tgt = td_target(s_rew[row], s_nd[row], a.gamma, qn);
            const TdLossIn li{a.loss_kind, a.weight != nullptr, a.weight ? a.weight[row] : 1.f, a.has_clip, a.clip_min, a.clip_max};
            float lossb, td;
            const float dl = td_loss_row(pred, tgt, li, lossb, td);
            float dq = dl / (float)B;
            if (a.relu[L - 1] && !(pred > 0.f)) dq = 0.f;   // MlpConfig::activation_out (mlp/base.rs:36): Q = relu(z), dL/dz = dL/dQ * [Q > 0]
            if (sub == 0) { a.pred[row] = pred; a.tgt[row] = tgt; a.loss_row[row] = lossb; s_loss[row] = lossb; if (a.td_abs) a.td_abs[row] = td; }
            for (int c = sub; c < ld; c += 16) {
                const float g = c == act ? dq : 0.f;
                lds[a.lds_dy[L - 1] + row * lds_ld + c] = g;
                a.dy[L - 1][(size_t)row * ld + c] = g;
            }
        }
        mf_lds_sync();
        // loss = mean(loss_row): the fixed-order tree of k_mean_rows (red[t] += red[t + w], w = 256 .. 1).  For B <= 64 the
        // levels w >= 64 only add zeros, and the rest of the tree is one wave's shuffle-down - no barriers.
        if (B <= 64) {
            if (wave == 0) {
                float v = lane < B ? s_loss[lane] : 0.f;
#pragma unroll
                for (int w = 32; w > 0; w >>= 1) v += __shfl_down(v, w);
                if (lane == 0) a.loss[0] = v / (float)B;
            }
        } else {
            float lsum = 0.f;
            for (int b = tid; b < B; b += 512) lsum += s_loss[b];
            red[tid] = lsum;
            mf_lds_sync();
            for (int w = 256; w > 0; w >>= 1) { if (tid < w) red[tid] += red[tid + w]; mf_lds_sync(); }
            if (tid == 0) a.loss[0] = red[0] / (float)B;
        }
    }
    MF_TP(6);
    // ---- backward: dW_l, db_l and dX_l (masked by the ReLU of the producing layer) in one phase per layer
#pragma unroll
    for (int li = 0; li < MF_MAXL; ++li) {
        const int l = L - 1 - li;
        if (l < 0) break;
        const int Kp = a.Kp[l], Np = a.Np[l], KB = Kp / 32, NB = Np / 32;
        const int xoff = l == 0 ? a.lds_x0 : a.lds_act0[l - 1], ldx = Kp + 4;
        const int doff = a.lds_dy[l], ldy = Np + 4;
        const float* w = a.q + a.w[l];
        float* gw = a.grad + a.w[l];
        float* gb = a.grad + a.b[l];
        const int Bp = RB * 32;
        const int n_dw = KB * NB, n_dx = l > 0 ? RB * KB : 0;
        const bool pre = dx_pre_ok(l);
        // (bw is indexed by the unrolled phase counter: a compile-time constant, so the prefetch buffers stay in registers)
        if (dx_pre_ok(l - 1)) dx_prefetch(l - 1, bw[(li + 1) & 1]);
        for (int blk = wave; blk < n_dw + n_dx; blk += 8) {
            if (blk < n_dw) {   // dW[k][n] = sum_b x[b][k] dy[b][n]
                const int kb = blk / NB, nb = blk % NB;
                const f32x16 acc = mf_block(Bp, lane,
                                            [&](int i, int bq) {
                                                f32x4 v;
#pragma unroll
                                                for (int q = 0; q < 4; ++q) v[q] = bq + q < B ? lds[xoff + (bq + q) * ldx + kb * 32 + i] : 0.f;
                                                return v;
                                            },
                                            [&](int bq, int j) {
                                                f32x4 v;
#pragma unroll
                                                for (int q = 0; q < 4; ++q) v[q] = bq + q < B ? lds[doff + (bq + q) * ldy + nb * 32 + j] : 0.f;
                                                return v;
                                            });
                const int col = nb * 32 + (lane & 31);
#pragma unroll
                for (int r = 0; r < 16; ++r) gw[(size_t)(kb * 32 + mf_row(r, lane)) * Np + col] = acc[r];
            } else {            // dX[b][k] = relu'(x[b][k]) * sum_n dy[b][n] W[k][n]
                const int q = blk - n_dw, rb = q / KB, kb = q % KB;
                auto a4 = [&](int i, int nq) {
                    const int r = min(rb * 32 + i, B - 1);
                    return *reinterpret_cast<const f32x4*>(lds + doff + r * ldy + nq);
                };
                f32x16 acc;
                if (pre && blk == wave) acc = mf_block_pre(Np, lane, a4, bw[li & 1]);
                else acc = mf_block<true>(Np, lane, a4, [&](int nq, int j) { return *reinterpret_cast<const f32x4*>(w + (size_t)(kb * 32 + j) * Np + nq); });
                const int col = kb * 32 + (lane & 31);
                float* const dxp = a.dy[l - 1];
                const int dxo = a.lds_dy[l - 1];                            // (Kp of this layer == Np of the previous one: row stride ldx)
                if (rb * 32 + 32 <= B) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = rb * 32 + mf_row(r, lane);
                        const float g = lds[xoff + row * ldx + col] > 0.f ? acc[r] : 0.f;
                        lds[dxo + row * ldx + col] = g;
                        dxp[(size_t)row * Kp + col] = g;
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = rb * 32 + mf_row(r, lane);
                        if (row < B) {
                            const float g = lds[xoff + row * ldx + col] > 0.f ? acc[r] : 0.f;
                            lds[dxo + row * ldx + col] = g;
                            dxp[(size_t)row * Kp + col] = g;
                        }
                    }
                }
            }
        }
        // db[n] = sum_b dy[b][n], rows in order; taken from the LAST threads of the workgroup (the first waves hold the MFMA blocks),
        // eight LDS reads in flight
        for (int n = 511 - tid; n < Np; n += 512) {
            float s = 0.f;
            for (int b0 = 0; b0 < B; b0 += 8) {
                float t[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) t[u] = b0 + u < B ? lds[doff + (b0 + u) * ldy + n] : 0.f;
#pragma unroll
                for (int u = 0; u < 8; ++u) s += t[u];
            }
            gb[n] = s;
        }
        if (l > 0) mf_lds_sync();
        MF_TP(7 + (L - 1 - l));
    }
    // ---- Adam + soft update (see k_dqn_mlp_step).  The gradient arena is exchanged through global memory: one full release /
    // acquire barrier - placed AFTER the loads of the parameters and moments of the first pass are in flight (they do not depend
    // on the gradients), so that round trip overlaps the tail of the backward phases.
    if (a.do_adam || a.do_track) {
        // Only the rows of W_l below the layer's logical input width (and the biases) can be non-zero: the zero-padding rows have
        // zero gradients and moments, and Adam / track map (0, 0, 0, 0) to 0 exactly - they are skipped, not recomputed
        // (CartPole: 4 of layer 0's 64 rows are real; 2 160 vectors instead of 3 120, one pass of the loop below).
        unsigned r_off[2 * MF_MAXL], r_cnt[2 * MF_MAXL], T = 0;   // ranges of 16-byte vectors of the arena
#pragma unroll
        for (int l = 0; l < MF_MAXL; ++l) {
            const bool on = l < L;
            r_off[2 * l] = on ? (unsigned)(a.w[l] / 4) : 0u;     r_cnt[2 * l] = on ? (unsigned)(a.in_rows_l[l] * a.Np[l] / 4) : 0u;
            r_off[2 * l + 1] = on ? (unsigned)(a.b[l] / 4) : 0u; r_cnt[2 * l + 1] = on ? (unsigned)(a.Np[l] / 4) : 0u;
            T += r_cnt[2 * l] + r_cnt[2 * l + 1];
        }
        auto locate = [&](unsigned e) {   // e-th vector of the concatenated ranges -> vector index in the arena
            unsigned at = 0;
            bool done = false;
#pragma unroll
            for (int r = 0; r < 2 * MF_MAXL; ++r) {
                if (!done && e < r_cnt[r]) { at = r_off[r] + e; done = true; }
                if (!done) e -= r_cnt[r];
            }
            return at;
        };
        constexpr int V = 5;
        for (unsigned base = 0; base < T; base += 512u * V) {   // uniform trip count (the barrier sits inside)
            f32x4 p[V], g[V], mm[V], vv[V], t[V];
            unsigned at[V];
            bool ok[V];
#pragma unroll
            for (int u = 0; u < V; ++u) {
                const unsigned e = base + tid + (unsigned)u * 512u;
                ok[u] = e < T;
                at[u] = locate(ok[u] ? e : 0u);
                p[u] = reinterpret_cast<const f32x4*>(a.q)[at[u]];
                if (a.do_adam) { mm[u] = reinterpret_cast<const f32x4*>(a.m)[at[u]]; vv[u] = reinterpret_cast<const f32x4*>(a.v)[at[u]]; }
                if (a.do_track) t[u] = reinterpret_cast<const f32x4*>(a.q_tgt)[at[u]];
            }
            if (base == 0) __syncthreads();
            if (a.do_adam) {
#pragma unroll
                for (int u = 0; u < V; ++u) g[u] = reinterpret_cast<const f32x4*>(a.grad)[at[u]];
            }
#pragma unroll
            for (int u = 0; u < V; ++u) {
                if (!ok[u]) continue;
                if (a.do_adam) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) { float pe = p[u][j], me = mm[u][j], ve = vv[u][j]; adam_element(pe, g[u][j], me, ve, a.adam); p[u][j] = pe; mm[u][j] = me; vv[u][j] = ve; }
                    reinterpret_cast<f32x4*>(a.q)[at[u]] = p[u]; reinterpret_cast<f32x4*>(a.m)[at[u]] = mm[u]; reinterpret_cast<f32x4*>(a.v)[at[u]] = vv[u];
                }
                if (a.do_track) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) t[u][j] = track_element(p[u][j], t[u][j], a.tau, a.omt);
                    reinterpret_cast<f32x4*>(a.q_tgt)[at[u]] = t[u];
                }
            }
        }
    }
    MF_TP(11);
}

}  // namespace
