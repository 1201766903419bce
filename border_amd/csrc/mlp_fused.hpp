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
#define MF_TP(k) do { if (threadIdx.x == 0 && a.trace) a.trace[k] = wall_clock64(); } while (0)
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
};

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

__global__ __launch_bounds__(512) void k_dqn_mlp_step(MlpFusedArgs a)
{
    __shared__ float red[512];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rb * 32 + mf_row(r, lane);
                if (row < B) {
                    float v = acc[r] + bv;
                    if (a.relu[l]) v = v > 0.f ? v : 0.f;
                    a.act[z][l][(size_t)row * Np + col] = v;
                }
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
            const float tgt = a.reward[row] + ((float)(1 - (int)a.term[row]) * a.gamma) * qn;
            const TdLossIn li{a.loss_kind, a.weight != nullptr, a.weight ? a.weight[row] : 1.f, a.has_clip, a.clip_min, a.clip_max};
            float lossb, td;
            const float dl = td_loss_row(pred, tgt, li, lossb, td);
            const float dq = dl / (float)B;
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
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = rb * 32 + mf_row(r, lane);
                    if (row < B) a.dy[l - 1][(size_t)row * Kp + col] = x[(size_t)row * ldx + col] > 0.f ? acc[r] : 0.f;
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

}  // namespace
