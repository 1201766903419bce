// SAC agent on MI355X: Sac::opt_ (border-tch-agent/src/sac/base.rs:175-198) with
// action_logp (:73-87), qvals_min (:89-105), update_actor (:151-167), update_critic (:107-149),
// soft_update (:169-173); Actor = Mlp2 (mlp/mlp2.rs:23-50), Critic = Mlp on cat(obs, act)
// (mlp/base.rs:83-107), EntCoef (sac/ent_coef.rs:27-75).
// Dense layers run on the FP32-MFMA kernels (dense.hpp); the elementwise SAC math and its hand-
// derived backward are the kernels below.  Reference quirks are kept on purpose:
//   sigma = exp(clip(exp(head2), min_lstd, max_lstd))  (Mlp2 already exponentiates, SAC does again);
//   log-prob omits -sum(ln sigma); is_truncated ignored; the actor is updated BEFORE the critics and
//   the critic target uses the updated actor; every critic is tracked after every update.
#include <algorithm>
#include <cstdlib>

#include "dense.hpp"
#include "dense_chain.hpp"
#include "queue_flags.hpp"

using namespace bdr;

namespace {

// counter-based N(0,1): splitmix64 hash -> Box-Muller (the reference draws from torch's global CPU RNG)
__device__ __forceinline__ float randn_at(uint64_t seed, uint64_t counter, size_t i)
{
    uint64_t x = (seed + 0x9E3779B97F4A7C15ull) ^ ((counter + i + 1) * 0xBF58476D1CE4E5B9ull);
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31;
    const float u1 = ((float)(x >> 40) + 1.0f) * (1.0f / 16777217.0f);
    const float u2 = (float)((x >> 8) & 0xFFFFFF) * (1.0f / 16777216.0f);
    return sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
}

// a = tanh(sigma*z + mean), log_p = sum(-0.5 ln 2pi - 0.5 z^2) - sum ln(1 - a^2 + eps).   One wave per row.
struct SacActionArgs {
    const float* mean; const float* e; int ld;     // head outputs [B][ld]
    const float* z;                                // [B][A] N(0,1)
    float* xq; int ldq; int col0;                  // action written into the critic input at columns [col0, col0+A)
    float* a_out; float* s_out; float* sd_out;     // [B][ld] (saved for backward; may be null)
    float* logp;                                   // [B]
    int B, A; float lo, hi, eps;
};
__global__ __launch_bounds__(256) void k_sac_action(SacActionArgs p)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    if (row >= p.B) return;
    float nl = 0.f, sl = 0.f;
    for (int j = lane; j < p.A; j += 64) {
        const size_t q = (size_t)row * p.ld + j;
        const float z = p.z[(size_t)row * p.A + j];
        const float s = expf(p.e[q]);                                   // Mlp2::forward .exp()
        const float cl = fminf(fmaxf(s, p.lo), p.hi);                   // lstd.clip(min_lstd, max_lstd)
        const float sd = expf(cl);                                      // .exp()
        const float a = tanhf(sd * z + p.mean[q]);
        p.xq[(size_t)row * p.ldq + p.col0 + j] = a;
        if (p.a_out) { p.a_out[q] = a; p.s_out[q] = s; p.sd_out[q] = sd; }
        nl += -0.91893853320467274178f - 0.5f * (z * z);
        sl += logf((1.0f - a * a) + p.eps);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { nl += __shfl_xor(nl, off); sl += __shfl_xor(sl, off); }
    if (lane == 0) p.logp[row] = nl - sl;
}

// rows: qmin index over critics, per-critic upstream gradient (1 at column 0 for the selected critic),
// and the per-row actor loss term alpha*log_p - qmin
struct SacSelectArgs {
    const float* q[4]; int ldq;          // critic outputs [B][ldq], value in column 0
    float* dout[4];                      // [B][ldq]
    const float* logp; const float* log_alpha;
    float* out; float scale; int accumulate;   // loss_actor += mean(alpha*log_p - qmin)
    int B, NC;
    // EntCoef::update (ent_coef.rs:69-75) first: loss = -(log_alpha * (logp + H)).mean(); Adam on the scalar
    int auto_alpha; float target; float* log_alpha_rw; float* al_m; float* al_v; AdamScalars s;
    const unsigned* poison;              // a cross-queue wait timed out: no EntCoef step
    unsigned long long* applied; unsigned long long step;   // the step number of the last EntCoef step that was NOT skipped (Sac::on_gate_timeout)
};
// Sums over the batch rows, in an order that does not depend on who computes it (the row-block kernels of sac_fused.hpp form the
// block partials in their own workgroups): rows in blocks of 32, a block's partial = the 32-lane butterfly (xor 16, 8, 4, 2, 1) of
// its rows' values, the total = the partials added one by one in block order.
__device__ __forceinline__ float butterfly32(float v)
{
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
__device__ __forceinline__ float add_scaled(float total, float s, float scale)
{
#pragma clang fp contract(off)
    const float t = s * scale;
    return total + t;
}
__device__ __forceinline__ float actor_loss_sum(float base, float alpha, float s_logp, float s_qm, float scale)   // base + (alpha * sum log_p - sum qmin) * scale
{
#pragma clang fp contract(off)
    const float a = alpha * s_logp;
    const float d = a - s_qm;
    const float t = d * scale;
    return base + t;
}

// one 1024-thread workgroup: value(b) for every row b < B
template <class F>
__device__ __forceinline__ float row_sum_1024(int B, F&& value, float* red32)
{
    float total = 0.f;
    for (int base = 0; base < B; base += 1024) {
        const int b = base + (int)threadIdx.x;
        const float part = butterfly32(b < B ? value(b) : 0.f);
        if ((threadIdx.x & 31) == 0) red32[threadIdx.x >> 5] = part;
        __syncthreads();
        const int nb = min(32, (B - base + 31) / 32);
        for (int k = 0; k < nb; ++k) total += red32[k];
        __syncthreads();
    }
    return total;
}

// One workgroup of 1024 threads (a row per thread at SAC's batch sizes: every load of the kernel is in flight at once).  Only
// column 0 of the upstream gradients is written: the other ldq-1 (padding) columns are zero from allocation and nothing else
// stores there.  loss_actor = mean(alpha * log_p - qmin) is formed as (alpha * sum(log_p) - sum(qmin)) / B.
__global__ __launch_bounds__(1024) void k_sac_select(SacSelectArgs p)
{
    __shared__ float red[32];
    float log_alpha = p.log_alpha[0];
    if (p.auto_alpha) {   // update_actor calls ent_coef.update(log_p) before it uses alpha (sac/base.rs:155)
        const float g = -(row_sum_1024(p.B, [&](int b) { return p.logp[b] + p.target; }, red) / (float)p.B);
        const float mm = p.al_m[0] * p.s.b1 + g * p.s.omb1;       // every thread computes the same scalar step; thread 0 stores it
        const float vv = p.al_v[0] * p.s.b2 + p.s.omb2 * g * g;
        const float denom = __fsqrt_rn(vv) / p.s.sqrt_bc2 + p.s.eps;
        log_alpha = log_alpha + p.s.neg_step * mm / denom;
        __syncthreads();                                          // all reads of al_m / al_v / log_alpha are done
        if (threadIdx.x == 0 && !(p.poison && *p.poison)) { p.log_alpha_rw[0] = log_alpha; p.al_m[0] = mm; p.al_v[0] = vv; if (p.applied) *p.applied = p.step; }
    }
    const float alpha = expf(log_alpha);
    const float s_logp = row_sum_1024(p.B, [&](int b) { return p.logp[b]; }, red);
    const float s_qm = row_sum_1024(p.B, [&](int b) {
        int im = 0;
        float qm = p.q[0][(size_t)b * p.ldq];
        for (int i = 1; i < p.NC; ++i) { const float v = p.q[i][(size_t)b * p.ldq]; if (v < qm) { qm = v; im = i; } }
        for (int i = 0; i < p.NC; ++i) p.dout[i][(size_t)b * p.ldq] = i == im ? 1.0f : 0.0f;
        return qm;
    }, red);
    if (threadIdx.x == 0) p.out[0] = actor_loss_sum(p.accumulate ? p.out[0] : 0.f, alpha, s_logp, s_qm, p.scale);
}

// dL/dmean, dL/d(head2) from dL/da = (alpha * 2a/(1-a^2+eps) - d qmin/da) / B
struct SacActorGradArgs {
    const float* a; const float* s; const float* sd; int ld;
    const float* z;
    const float* dxq[4]; int ldq; int col0; int NC;   // critics' input gradients (selected rows only are non-zero)
    const float* log_alpha;
    float* gmean; float* ge;                          // [B][ld]
    int B, A; float lo, hi, eps;
};
__global__ void k_sac_actor_grad(SacActorGradArgs p)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= p.B * p.ld) return;
    const int b = t / p.ld, j = t % p.ld;
    if (j >= p.A) { p.gmean[t] = 0.f; p.ge[t] = 0.f; return; }
    const float a = p.a[t], s = p.s[t], sd = p.sd[t];
    float dq = 0.f;
    for (int i = 0; i < p.NC; ++i) dq += p.dxq[i][(size_t)b * p.ldq + p.col0 + j];
    const float alpha = expf(p.log_alpha[0]);
    const float dlogp = (2.0f * a) / ((1.0f - a * a) + p.eps);
    const float ga = (alpha * dlogp - dq) / (float)p.B;
    const float gu = ga * (1.0f - a * a);
    p.gmean[t] = gu;
    const float inr = (s >= p.lo && s <= p.hi) ? 1.0f : 0.0f;
    p.ge[t] = gu * p.z[(size_t)b * p.A + j] * sd * inr * s;
}

// The TD step's row arithmetic, shared by k_sac_critic_td and the row-block kernel k_sac_td_last.  No fused multiply-add
// contraction (hipcc contracts across statements, and HIP's __fmul_rn is a plain `*`): a product contracted into a sum in one kernel
// but not in the other would break the bit-identity of the two paths.
__device__ __forceinline__ float sac_td_target(float reward_scale, float reward, float term, float gamma, float qmin, float alpha, float logp)
{
#pragma clang fp contract(off)
    // (plain operators: HIP's __fmul_rn / __fsub_rn are inline functions WITHOUT this pragma, their operations stay contractable)
    const float al = alpha * logp;
    const float nq = qmin - al;                                                                 // min_i Qtgt_i - alpha * log p'
    const float a = reward_scale * reward;
    const float k = (1.0f - term) * gamma;
    const float c = k * nq;
    return a + c;
}
__device__ __forceinline__ void sac_td_loss(float q, float tgt, int loss_kind, float& l, float& g)
{
#pragma clang fp contract(off)
    const float d = q - tgt;
    if (loss_kind == 1) { const float z = fabsf(d); const float hz = 0.5f * z; l = z < 1.f ? hz * z : z - 0.5f; g = z < 1.f ? d : (d > 0.f ? 1.f : -1.f); }
    else { l = d * d; g = 2.f * d; }
}

// critic losses + upstream gradients (column 0; see k_sac_select) of every critic in one workgroup:
// loss_critic += sum_i mean(loss(Q_i - tgt)) / NC, critic by critic in k_sum_rows' order
struct SacTdArgs {
    const float* q[4]; float* dout[4]; int ldq; int NC;
    float* out; float scale; int accumulate; int B; int loss_kind;
    // tgt = reward_scale*r + ((1 - term)*gamma) * (min_i Qtgt_i - alpha*logp')   (sac/base.rs:113-122; is_truncated ignored)
    const float* qt[4]; const float* logp; const float* log_alpha; const float* reward; const int8_t* term; float gamma, reward_scale;
    float* tgt;
};
__global__ __launch_bounds__(1024) void k_sac_critic_td(SacTdArgs p)
{
    __shared__ float red[32];
    float total = p.accumulate ? p.out[0] : 0.f;
    const float alpha = expf(p.log_alpha[0]);
    for (int b = threadIdx.x; b < p.B; b += 1024) {
        float qm = p.qt[0][(size_t)b * p.ldq];
        for (int i = 1; i < p.NC; ++i) qm = fminf(qm, p.qt[i][(size_t)b * p.ldq]);
        p.tgt[b] = sac_td_target(p.reward_scale, p.reward[b], (float)p.term[b], p.gamma, qm, alpha, p.logp[b]);
    }
    for (int i = 0; i < p.NC; ++i) {   // critic by critic; a thread reads back only the targets it wrote itself
        const float si = row_sum_1024(p.B, [&](int b) {
            float l, dl;
            sac_td_loss(p.q[i][(size_t)b * p.ldq], p.tgt[b], p.loss_kind, l, dl);
            p.dout[i][(size_t)b * p.ldq] = dl / (float)p.B;
            return l;
        }, red);
        total = add_scaled(total, si, p.scale);
    }
    if (threadIdx.x == 0) p.out[0] = total;
}

// every input matrix of the step from the batch rows in one launch: the actor's zero-padded obs / next_obs, and the critics'
// three inputs (obs | a_pi), (next_obs | a_pi'), (obs | act) - the sampled actions are filled in by k_sac_action
struct SacPackArgs {
    const float* obs; const float* next; const float* act; int O, A, B;
    float* x_o; float* x_no; int ldp;
    float* xq_a; float* xq_n; float* xq_c; int ldq;
    float* z; size_t nz; uint64_t seed, counter;   // nz > 0: also draw the step's N(0,1) numbers (randn_at)
    // do_gather: the kernel is also the replay buffer's sample of this step (replay_sample_plan): eight rows per workgroup draw their
    // indices of the buffer's StdRng stream, and the rows go to the buffer's batch arrays (obs / next / act above, reward, flags)
    // as well as into the padded matrices
    int do_gather; GatherArgs g;
};
constexpr int SAC_PACK_ROWS = 8;
__global__ __launch_bounds__(256) void k_sac_pack(SacPackArgs p)
{
    __shared__ uint64_t s_row[SAC_PACK_ROWS];
    const int tid = threadIdx.x, r0 = (int)blockIdx.x * SAC_PACK_ROWS;
    for (size_t t = (size_t)blockIdx.x * 256 + tid; t < p.nz; t += (size_t)gridDim.x * 256) p.z[t] = randn_at(p.seed, p.counter, t);
    if (p.do_gather) {
        if (tid < SAC_PACK_ROWS && r0 + tid < p.B) {
            const uint64_t row = (uint64_t)chacha12_word(p.g.key, p.g.word_pos + (uint64_t)(r0 + tid)) % p.g.size;   // (StdRng::next_u32() as usize) % size
            s_row[tid] = row; p.g.ixs[r0 + tid] = row;
            const uint8_t* rec = p.g.ring + row * p.g.stride;
            p.g.b_reward[r0 + tid] = *reinterpret_cast<const float*>(rec + p.g.tail_off);
            p.g.b_term[r0 + tid] = *reinterpret_cast<const int8_t*>(rec + p.g.tail_off + 4);
            p.g.b_trunc[r0 + tid] = *reinterpret_cast<const int8_t*>(rec + p.g.tail_off + 5);
        }
        __syncthreads();
    }
    const int W = p.O + p.A;
    for (int e = tid; e < SAC_PACK_ROWS * W; e += 256) {
        const int rr = e / W, c = e % W, b = r0 + rr;
        if (b >= p.B) break;
        if (c < p.O) {
            float o, n;
            if (p.do_gather) {
                const uint8_t* rec = p.g.ring + s_row[rr] * p.g.stride;
                o = reinterpret_cast<const float*>(rec)[c]; n = reinterpret_cast<const float*>(rec + p.g.next_off)[c];
                reinterpret_cast<float*>(p.g.b_obs)[(size_t)b * p.O + c] = o; reinterpret_cast<float*>(p.g.b_next)[(size_t)b * p.O + c] = n;
            } else { o = p.obs[(size_t)b * p.O + c]; n = p.next[(size_t)b * p.O + c]; }
            p.x_o[(size_t)b * p.ldp + c] = o; p.x_no[(size_t)b * p.ldp + c] = n;
            p.xq_a[(size_t)b * p.ldq + c] = o; p.xq_c[(size_t)b * p.ldq + c] = o; p.xq_n[(size_t)b * p.ldq + c] = n;
        } else {
            float av;
            if (p.do_gather) {
                av = reinterpret_cast<const float*>(p.g.ring + s_row[rr] * p.g.stride + p.g.act_off)[c - p.O];
                reinterpret_cast<float*>(p.g.b_act)[(size_t)b * p.A + (c - p.O)] = av;
            } else av = p.act[(size_t)b * p.A + (c - p.O)];
            p.xq_c[(size_t)b * p.ldq + c] = av;
        }
    }
}

}  // namespace
#include "sac_fused.hpp"
namespace {

__global__ void k_randn(float* __restrict__ out, size_t n, uint64_t seed, uint64_t counter)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = randn_at(seed, counter, i);
}

}  // namespace

// ================================================================================================
// The per-batch matrices of one update.  The agent owns TWO sets: with the side queue (Sac::two_queues) the first launches of
// update n+1 - sample/pack, the actor's forward on obs - run while update n's critic phase is still using its own set.
struct SacBatch {
    float *x_o = nullptr, *x_no = nullptr;          // packed obs / next_obs [B][Kp_pi]
    std::vector<float*> t_act;                      // trunk activations
    float *mean = nullptr, *e = nullptr, *a_s = nullptr, *s_s = nullptr, *sd_s = nullptr, *gmean = nullptr, *ge = nullptr;
    std::vector<float*> t_dy;                       // trunk gradients
    float *xq_a = nullptr, *xq_n = nullptr, *xq_c = nullptr;   // critic inputs [B][Kp_q]: (obs | a_pi), (next_obs | a_pi'), (obs | act)
    std::vector<float*> c_act[4];                   // critic activations on xq_a / xq_n
    std::vector<float*> c2_act[4];                  // critic activations on xq_c (the TD pass)
    std::vector<float*> c_dy[4];                    // critic gradients per layer
    float* dxq[4] = {nullptr, nullptr, nullptr, nullptr};   // critic input gradients [B][Kp_q]
    float *logp = nullptr, *tgt = nullptr, *z_a = nullptr;   // z_a: [2][B][A] N(0,1) draws (actor pass, then target pass)
};

struct Sac : bdr_agent, SacBatch {
    bdr_sac_config cfg;
    int O = 0, A = 0, NC = 1;
    MlpLayout pi;               // trunk layers + [ml, sl]
    int n_trunk = 0;
    MlpLayout qn;               // critic layout (same for every critic / target)
    // arenas
    float *pi_p = nullptr, *pi_g = nullptr, *pi_m = nullptr, *pi_v = nullptr;
    float* pi_vmax = nullptr; float* q_vmax[4] = {nullptr};   // AdamW{amsgrad: true} of the actor / the critics: max_exp_avg_sq (parameter model +400)
    float* q_p[4] = {nullptr}; float* q_t[4] = {nullptr}; float* q_g[4] = {nullptr}; float* q_m[4] = {nullptr}; float* q_v[4] = {nullptr};
    float *log_alpha = nullptr, *al_m = nullptr, *al_v = nullptr;
    uint64_t step_pi = 0, step_q[4] = {0}, step_al = 0;
    // device words: the step number of the last EntCoef / actor / critic optimizer pass that was not skipped under the poison word
    static constexpr int AP_AL = 0, AP_PI = 1, AP_Q = 2;
    unsigned long long* applied = nullptr;
    // batch buffers: the current set is the SacBatch base (flip() swaps it with `other`)
    int B = 0; uint64_t batch_gen = 0;   // bumped by every re-allocation (the captured graph holds the old pointers)
    SacBatch other;
    // row-chunk partials of the grouped dW launches: pi layer l at pi_part + pi_off[l]; critic i, layer l at q_part + i * q_part_stride + q_off[l]
    float *pi_part = nullptr, *q_part = nullptr; size_t q_part_stride = 0;
    std::vector<size_t> pi_off, q_off; std::vector<int> pi_chunks, q_chunks;
    float* scal = nullptr;                          // [0] loss_critic (sum over critics / NC), [1] loss_actor
    // host staging for update_on_batch
    float *u_obs = nullptr, *u_next = nullptr, *u_act = nullptr, *u_rew = nullptr; int8_t* u_term = nullptr; uint64_t u_cap = 0;
    uint64_t noise_counter = 0;
    // bdr_sac_probe: the actor-phase values an update overwrites later (Q_i(obs, a_pi), log p(a_pi | obs)), copied aside by
    // update_on_batch only (probe_copy; never inside a captured step)
    bool probe_copy = false; int probe_B = 0; int last_B = 0;
    float* pr_qpi[4] = {nullptr}; float* pr_logp = nullptr;
    StepGraph graph; StepGraphPolicy graph_policy;   // step_graph.hpp
    bool gather_in_pack = true;               // BDR_NO_STEP_GATHER=1: separate gather launch
    bool fuse_rows = true;                    // BDR_NO_SAC_FUSE=1: the narrow layers as launches of their own (sac_fused.hpp)
    bool heads_in_chain = false;              // BDR_SAC_HEADS_FUSE=1: k_sac_heads_action's part by the last workgroup of each row block of the trunk launch (measured: -3 %, LAB.md 5)
    static constexpr int HEAD_TICKETS = 2048;
    unsigned* head_tickets = nullptr;         // [2][HEAD_TICKETS] row-block tickets of k_sac_pi_chain_heads (prologue pass / critic-phase pass)
    bool wait_in_kernel = true;               // BDR_SAC_WAIT_PACKET=1: the main queue's wait for the prologue as a one-wave packet instead of inside the first critic launch
    ChainWait pending_wait;                   // set by opt_enqueue, taken by the first critic launch of update_rest
    bool tail_next = true;                    // BDR_SAC_TAIL_IN_KERNEL=1: the row-block kernels' batch-wide parts by their own last workgroup (ticket) instead of in the next launch
    bool chain2 = true; int chain2_tpw = 0;   // BDR_NO_SAC_CHAIN=1: a two-layer trunk as two launches (dense_chain.hpp); BDR_SAC_CHAIN_TPW=1|4: tile form
    unsigned* tickets = nullptr;              // [2] last-workgroup tickets of k_sac_q_last / k_sac_td_last
    float* lrow = nullptr;                    // [3 + NC][ceil(B / 32)] block partials of the batch-wide sums (k_sac_q_last, k_sac_td_last)
    bool small_gemm = true;                   // BDR_NO_SMALL_GEMM=1: the 64x64-tile kernels of the large-batch agents
    // Side queue (queue_flags.hpp).  The actor's forward on obs needs nothing of the previous update but its actor step, which
    // is over before that update's critic phase starts: the prologue of update n+1 (sample + pack, trunk, heads + action) runs on
    // `side`, in the other buffer set, beside the critic phase of update n on `stream`.
    //   side:    wait PI(n) -> pack, trunk layers, heads+action (set n+1) -> set PRO(n+1)
    //   stream:  wait PRO(n+1) -> critics on (obs, a_pi) ... actor step -> set PI(n+1) -> critic phase
    // BDR_SAC_SIDE_QUEUE=0: everything on `stream` (also taken by profiling, prioritized / single-frame buffers, a captured step
    // and bdr_sac_update_on_batch).  Same kernels, same arguments, same bits either way.
    enum { SIG_PI = 0, SIG_PRO = 1, SIG_SCRATCH = 2 };
    hipStream_t side = nullptr; hipEvent_t ev_main = nullptr;
    unsigned* sig = nullptr;                  // [4] flag words
    unsigned epoch = 0;                       // update counter of the two-queue sequence
    bool two_queues = false;                  // the streams exist and sit on different hardware queues
    unsigned long long flag_limit = FLAG_WAIT_LIMIT;   // 100 MHz ticks (10 s; BDR_GATE_LIMIT_MS for tests)
    long long stall_at = -1;                  // tests (BDR_SAC_STALL_AT=n): update n does not publish PRO - its consumer's wait has to time out
    bool main_ahead = true;                   // `stream` carries work the flags do not cover (a one-queue update, a parameter exchange,
                                              // set_params, the buffers' first memset): the side queue waits for it once, with an event

    ~Sac() override
    {
        (void)hipSetDevice(device);
        (void)hipStreamSynchronize(stream);
        if (side) { (void)hipStreamSynchronize(side); stream_retire(side); (void)hipStreamDestroy(side); }   // (a replay buffer may name it as its last reader)
        if (ev_main) (void)hipEventDestroy(ev_main);
        (void)hipFree(sig);
        free_batch();
        (void)hipFree(pi_p); (void)hipFree(pi_g); (void)hipFree(pi_m); (void)hipFree(pi_v); (void)hipFree(pi_vmax);
        for (auto q : q_vmax) (void)hipFree(q);
        for (int i = 0; i < 4; ++i) { (void)hipFree(q_p[i]); (void)hipFree(q_t[i]); (void)hipFree(q_g[i]); (void)hipFree(q_m[i]); (void)hipFree(q_v[i]); }
        (void)hipFree(log_alpha); (void)hipFree(al_m); (void)hipFree(al_v); (void)hipFree(scal);
        (void)hipFree(u_obs); (void)hipFree(u_next); (void)hipFree(u_act); (void)hipFree(u_rew); (void)hipFree(u_term);
        for (int i = 0; i < 4; ++i) (void)hipFree(pr_qpi[i]);
        (void)hipFree(tickets); (void)hipFree(applied); (void)hipFree(head_tickets);
        (void)hipFree(pr_logp);
    }
    static void free_set(SacBatch& b)
    {
        float** singles[] = {&b.x_o, &b.x_no, &b.mean, &b.e, &b.a_s, &b.s_s, &b.sd_s, &b.gmean, &b.ge, &b.xq_a, &b.xq_n, &b.xq_c, &b.logp, &b.tgt, &b.z_a};
        for (auto p : singles) { (void)hipFree(*p); *p = nullptr; }
        for (auto p : b.t_act) (void)hipFree(p);
        for (auto p : b.t_dy) (void)hipFree(p);
        b.t_act.clear(); b.t_dy.clear();
        for (int i = 0; i < 4; ++i) {
            for (auto p : b.c_act[i]) (void)hipFree(p);
            for (auto p : b.c2_act[i]) (void)hipFree(p);
            for (auto p : b.c_dy[i]) (void)hipFree(p);
            b.c_act[i].clear(); b.c2_act[i].clear(); b.c_dy[i].clear();
            (void)hipFree(b.dxq[i]); b.dxq[i] = nullptr;
        }
    }
    void free_batch()
    {
        free_set(*this); free_set(other);
        float** singles[] = {&pi_part, &q_part, &lrow};
        for (auto p : singles) { (void)hipFree(*p); *p = nullptr; }
    }
    void flip() { std::swap(static_cast<SacBatch&>(*this), other); }
    int32_t zalloc(float** p, size_t n)
    {
        BDR_TRY(alloc_f(p, n));
        BDR_HIP(hipMemsetAsync(*p, 0, std::max<size_t>(n, 4) * 4, stream));
        return BDR_OK;
    }
    int32_t alloc_set(SacBatch& b, int Bn)
    {
        const int Kp = pi.L[0].Kp, Ap = pi.L[n_trunk].Np, Kq = qn.L[0].Kp;
        BDR_TRY(zalloc(&b.x_o, (size_t)Bn * Kp)); BDR_TRY(zalloc(&b.x_no, (size_t)Bn * Kp));
        for (int i = 0; i < n_trunk; ++i) { float* p = nullptr; BDR_TRY(zalloc(&p, (size_t)Bn * pi.L[i].Np)); b.t_act.push_back(p); }
        for (int i = 0; i < n_trunk; ++i) { float* p = nullptr; BDR_TRY(zalloc(&p, (size_t)Bn * pi.L[i].Np)); b.t_dy.push_back(p); }
        float** heads[] = {&b.mean, &b.e, &b.a_s, &b.s_s, &b.sd_s, &b.gmean, &b.ge};
        for (auto p : heads) BDR_TRY(zalloc(p, (size_t)Bn * Ap));
        BDR_TRY(zalloc(&b.xq_a, (size_t)Bn * Kq)); BDR_TRY(zalloc(&b.xq_n, (size_t)Bn * Kq)); BDR_TRY(zalloc(&b.xq_c, (size_t)Bn * Kq));
        for (int i = 0; i < NC; ++i) {
            for (const auto& l : qn.L) {
                float *p = nullptr, *p2 = nullptr, *d = nullptr;
                BDR_TRY(zalloc(&p, (size_t)Bn * l.Np)); BDR_TRY(zalloc(&p2, (size_t)Bn * l.Np)); BDR_TRY(zalloc(&d, (size_t)Bn * l.Np));
                b.c_act[i].push_back(p); b.c2_act[i].push_back(p2); b.c_dy[i].push_back(d);
            }
            BDR_TRY(zalloc(&b.dxq[i], (size_t)Bn * Kq));
        }
        BDR_TRY(zalloc(&b.logp, Bn)); BDR_TRY(zalloc(&b.tgt, Bn));
        BDR_TRY(zalloc(&b.z_a, (size_t)2 * Bn * A));
        return BDR_OK;
    }
    int32_t ensure_batch(int Bn)
    {
        if (Bn <= B) return BDR_OK;
        BDR_HIP(hipStreamSynchronize(stream));
        if (side) BDR_HIP(hipStreamSynchronize(side));
        free_batch();
        BDR_TRY(alloc_set(*this, Bn)); BDR_TRY(alloc_set(other, Bn));
        BDR_TRY(zalloc(&lrow, (size_t)(3 + NC) * ((Bn + 31) / 32)));
        main_ahead = true;   // the memsets above are on `stream`
        // row chunks of the grouped dW launches.  64x64 tiles (k_igemm_red_group): about 512 workgroups per launch over all its
        // GEMMs, >= 64 rows per chunk; 32x32 split-reduction tiles (k_dense_dw_small_group): 256 rows per workgroup
        auto plan = [&](const MlpLayout& net, int jobs, std::vector<size_t>& off, std::vector<int>& chunks) {
            off.clear(); chunks.clear();
            size_t o = 0;
            const int target = std::max(32, 512 / jobs);
            for (const auto& l : net.L) {
                const int tiles = (l.Kp / 64) * (l.Np / 64);
                const int c = small_gemm ? std::max(1, std::min(16, Bn / 256))
                                         : std::max(1, std::min(std::min(16, Bn / 64), (target + tiles - 1) / tiles));
                off.push_back(o); chunks.push_back(c);
                o += (size_t)c * ((size_t)l.Kp * l.Np + l.Np);
            }
            return o;
        };
        BDR_TRY(zalloc(&pi_part, plan(pi, (int)pi.L.size(), pi_off, pi_chunks)));
        q_part_stride = plan(qn, (int)qn.L.size() * NC, q_off, q_chunks);
        BDR_TRY(zalloc(&q_part, q_part_stride * NC));
        B = Bn; batch_gen += 1;
        return BDR_OK;
    }

    // pi trunk + heads on packed input x -> mean, e
    int32_t pi_forward(const float* x, int Bn, hipStream_t st)
    {
        bdr_agent* a = this;
        DenseSrc in{x, pi.L[0].Kp};
        if (chain2 && small_gemm && n_trunk == 2 && dense_chain2_ok(pi.L[0], pi.L[1])) {   // (acting calls: one launch fewer)
            Bracket br(a, "pi_fwd");
            const float* pb[1] = {pi_p}; float* h0[1] = {t_act[0]}; float* h1[1] = {t_act[1]};
            BDR_TRY(dense_chain2_z(st, pi.L[0], pi.L[1], 1, pb, &in, h0, h1, Bn, chain2_tpw));
            in = DenseSrc{t_act[1], pi.L[1].Np};
        } else
        for (int i = 0; i < n_trunk; ++i) {
            Bracket br(a, "pi_fwd");
            BDR_TRY(dense_forward(a, st, pi.L[i], pi_p, in, t_act[i], Bn, small_gemm));
            in = DenseSrc{t_act[i], pi.L[i].Np};
        }
        // both heads (same shape, same input, consecutive in the arena) in one launch
        const DenseLayer& h0 = pi.L[n_trunk];
        const float* pb[2] = {pi_p, pi_p + (pi.L[n_trunk + 1].w - h0.w)};
        const DenseSrc ins[2] = {in, in};
        float* outs[2] = {mean, e};
        Bracket br(a, "pi_head");
        return dense_forward_z(st, h0, 2, pb, ins, outs, Bn, small_gemm);
    }
    // action_logp: writes the action into the action columns of the critic input `xq`, log_p into logp
    // the narrow layers fused into row-block kernels (sac_fused.hpp)?  Shapes the kernels do not cover take the layer-by-layer path.
    bool fused() const
    {
        const DenseLayer& hd = pi.L[n_trunk];
        return fuse_rows && small_gemm && A <= 32 && (O % 32) + A <= 32 && qn.L.size() >= 2 && hd.Np <= 64 && qn.L.back().Np <= 64 * 4;
    }
    // action_logp: writes the action into the action columns of the critic input `xq`, log_p into logp
    int32_t action_logp(const float* x, const float* z, int Bn, bool save, float* xq, hipStream_t st, unsigned* sig_flag = nullptr, unsigned sig_epoch = 0)
    {
        // sig_flag: "everything queued before this pass is complete" - carried by the first trunk launch where that kernel can
        if (sig_flag && !small_gemm) { BDR_TRY(flag_set(st, sig_flag, sig_epoch)); sig_flag = nullptr; }
        if (fused()) {   // trunk layer by layer, then heads + action + log-prob in one row-block kernel
            bdr_agent* a = this;
            DenseSrc in{x, pi.L[0].Kp};
            if (heads_in_chain && chain2 && small_gemm && n_trunk == 2 && dense_chain2_ok(pi.L[0], pi.L[1]) && pi.L[1].Np == C2_N0 && pi.L[0].in <= 32 &&
                pi.L[n_trunk].Kp == C2_N0 && (Bn + 31) / 32 <= HEAD_TICKETS) {
                // trunk + heads + action + log-probability in ONE launch: the row block's last workgroup does k_sac_heads_action's part (sac_fused.hpp)
                const float* pb[1] = {pi_p}; float* h0[1] = {t_act[0]}; float* h1[1] = {t_act[1]};
                const Chain2Args c = dense_chain2_args(pi.L[0], pi.L[1], 1, pb, &in, h0, h1, Bn, sig_flag, sig_epoch, nullptr);
                const DenseLayer &hm = pi.L[n_trunk], &hs = pi.L[n_trunk + 1];
                SacHeadsActionArgs p{};
                p.h = t_act[1]; p.ldh = pi.L[1].Np;
                p.hm = HeadRef{pi_p + hm.w, pi_p + hm.b, hm.relu}; p.hs = HeadRef{pi_p + hs.w, pi_p + hs.b, hs.relu}; p.kred = hm.Kp; p.w_ld = hm.Np;
                p.mean = mean; p.e = e; p.ld = hm.Np; p.z = z; p.xq = xq; p.ldq = qn.L[0].Kp; p.col0 = O;
                p.a_out = save ? a_s : nullptr; p.s_out = s_s; p.sd_out = sd_s; p.logp = logp;
                p.B = Bn; p.A = A; p.lo = (float)cfg.min_lstd; p.hi = (float)cfg.max_lstd; p.eps = (float)cfg.epsilon;
                Bracket br(a, "pi_fwd_heads");
                // (the prologue of the next update may run beside the critic phase's pass of this one: a ticket array each)
                BDR_HIP(step_launch(st, false, k_sac_pi_chain_heads, dim3(((Bn + 31) / 32) * (pi.L[1].Np / 32)), dim3(256), c, p, head_tickets + (save ? 0 : HEAD_TICKETS)));
                return BDR_OK;
            }
            if (chain2 && small_gemm && n_trunk == 2 && dense_chain2_ok(pi.L[0], pi.L[1])) {   // both trunk layers in one launch, same bits (dense_chain.hpp)
                Bracket br(a, "pi_fwd");
                const float* pb[1] = {pi_p}; float* h0[1] = {t_act[0]}; float* h1[1] = {t_act[1]};
                BDR_TRY(dense_chain2_z(st, pi.L[0], pi.L[1], 1, pb, &in, h0, h1, Bn, chain2_tpw, sig_flag, sig_epoch));
                in = DenseSrc{t_act[1], pi.L[1].Np};
            } else
            for (int i = 0; i < n_trunk; ++i) {
                Bracket br(a, "pi_fwd");
                BDR_TRY(dense_forward(a, st, pi.L[i], pi_p, in, t_act[i], Bn, small_gemm, i == 0 ? sig_flag : nullptr, sig_epoch));
                in = DenseSrc{t_act[i], pi.L[i].Np};
            }
            const DenseLayer &hm = pi.L[n_trunk], &hs = pi.L[n_trunk + 1];
            SacHeadsActionArgs p{};
            p.h = in.p; p.ldh = in.ld;
            p.hm = HeadRef{pi_p + hm.w, pi_p + hm.b, hm.relu}; p.hs = HeadRef{pi_p + hs.w, pi_p + hs.b, hs.relu}; p.kred = hm.Kp; p.w_ld = hm.Np;
            p.mean = mean; p.e = e; p.ld = hm.Np; p.z = z; p.xq = xq; p.ldq = qn.L[0].Kp; p.col0 = O;
            p.a_out = save ? a_s : nullptr; p.s_out = s_s; p.sd_out = sd_s; p.logp = logp;
            p.B = Bn; p.A = A; p.lo = (float)cfg.min_lstd; p.hi = (float)cfg.max_lstd; p.eps = (float)cfg.epsilon;
            Bracket br(this, "sac_heads_action");
            BDR_HIP(step_launch(st, false, k_sac_heads_action, dim3((Bn + 31) / 32), dim3(512), p));
            return BDR_OK;
        }
        if (sig_flag) BDR_TRY(flag_set(st, sig_flag, sig_epoch));
        BDR_TRY(pi_forward(x, Bn, st));
        SacActionArgs p{};
        p.mean = mean; p.e = e; p.ld = pi.L[n_trunk].Np; p.z = z; p.xq = xq; p.ldq = qn.L[0].Kp; p.col0 = O;
        p.a_out = save ? a_s : nullptr; p.s_out = s_s; p.sd_out = sd_s; p.logp = logp;
        p.B = Bn; p.A = A; p.lo = (float)cfg.min_lstd; p.hi = (float)cfg.max_lstd; p.eps = (float)cfg.epsilon;
        Bracket br(this, "sac_action");
        BDR_HIP(step_launch(st, false, k_sac_action, dim3((Bn + 3) / 4), dim3(256), p));
        return BDR_OK;
    }
    // Forward passes of the critic architecture, layer by layer, up to 4 (parameters, input) pairs per launch:
    // pass j runs parameters params[j] on input x[j] into acts[j][layer].
    int32_t critic_forward_n(int n, const float* const* params, const float* const* x, std::vector<float*>* const* acts, int Bn, int n_layers = -1)
    {
        bdr_agent* a = this;
        const size_t nl = n_layers < 0 ? qn.L.size() : (size_t)n_layers;
        for (int j0 = 0; j0 < n; j0 += 4) {
            const int nz = std::min(4, n - j0);
            DenseSrc in[4]; float* out[4];
            for (int j = 0; j < nz; ++j) in[j] = DenseSrc{x[j0 + j], qn.L[0].Kp};
            if (chain2 && small_gemm && nl == 2 && dense_chain2_ok(qn.L[0], qn.L[1])) {   // the two wide layers in one launch, same bits (dense_chain.hpp)
                float* h0[4]; float* h1[4];
                for (int j = 0; j < nz; ++j) { h0[j] = (*acts[j0 + j])[0]; h1[j] = (*acts[j0 + j])[1]; }
                Bracket br(a, "q_fwd");
                const ChainWait w = pending_wait;
                pending_wait = ChainWait{};
                BDR_TRY(dense_chain2_z(stream, qn.L[0], qn.L[1], nz, params + j0, in, h0, h1, Bn, chain2_tpw, nullptr, 0, &w));
                continue;
            }
            for (size_t l = 0; l < nl; ++l) {
                for (int j = 0; j < nz; ++j) out[j] = (*acts[j0 + j])[l];
                Bracket br(a, "q_fwd");
                if (nz == 1) BDR_TRY(dense_forward(a, stream, qn.L[l], params[j0], in[0], out[0], Bn, small_gemm));
                else BDR_TRY(dense_forward_z(stream, qn.L[l], nz, params + j0, in, out, Bn, small_gemm));
                for (int j = 0; j < nz; ++j) in[j] = DenseSrc{out[j], qn.L[l].Np};
            }
        }
        return BDR_OK;
    }
    // dX of layer l for all critics in one launch: c_dy[i][l] -> (l == 0 ? dxq[i] : c_dy[i][l-1]), ReLU mask from `acts`
    // tail: the batch-wide part of the row-block kernel in front of this launch rides in it as one more workgroup (sac_fused.hpp k_dense_small_dx_tail)
    int32_t critic_dx_all(int l, int Bn, std::vector<float*>* acts, const SacTailArgs* tail = nullptr, bool tail_varying = false)
    {
        const float* pb[4]; const float* dy[4]; float* dx[4]; const float* mask[4];
        for (int i = 0; i < NC; ++i) {
            pb[i] = q_p[i]; dy[i] = c_dy[i][l]; dx[i] = l == 0 ? dxq[i] : c_dy[i][l - 1]; mask[i] = l == 0 ? nullptr : acts[i][l - 1];
        }
        if (tail) {   // (small_gemm; the arguments of dense_dx_z)
            const DenseLayer& ly = qn.L[l];
            DenseArgsZ dz{};
            for (int z = 0; z < NC; ++z) {
                DenseArgs& d = dz.a[z];
                d.x = DenseSrc{dy[z], ly.Np}; d.w = pb[z] + ly.w; d.out = dx[z]; d.ldo = ly.Kp; d.mask = mask[z]; d.ldm = ly.Kp;
                d.M = Bn; d.ncols = ly.Kp; d.kred = ly.Np; d.w_ld = ly.Np;
            }
            BDR_HIP(step_launch(stream, tail_varying, k_dense_small_dx_tail, dim3(((Bn + 31) / 32) * (ly.Kp / 32) + 1, 1, NC), dim3(256), dz, *tail));
            return BDR_OK;
        }
        if (NC == 1) return dense_dx(stream, qn.L[l], pb[0], dy[0], dx[0], mask[0], Bn, false, small_gemm);
        return dense_dx_z(stream, qn.L[l], NC, pb, dy, dx, l == 0 ? nullptr : mask, Bn, small_gemm);
    }
    // OptimizerConfig::{Adam, AdamW} of one model (opt.rs:30-57) at optimizer step `step`
    static AdamScalars opt_scalars(const bdr_adamw_config& o, double lr, uint64_t step)
    {
        return adam_scalars_for(o.opt_kind == BDR_OPT_ADAMW, lr, o.beta1, o.beta2, o.eps, o.weight_decay, step);
    }
    // partial sums of a grouped dW launch -> gradient arena, Adam, (tracking) for `ninst` networks of one layout
    int32_t reduce_adam(const MlpLayout& net, const std::vector<size_t>& off, const std::vector<int>& chunks, int Bn, const float* part,
                        size_t inst_stride, int ninst, float* const* p, float* const* g, float* const* m, float* const* v,
                        float* const* tgt_p, const AdamScalars* sc, int applied_slot, uint64_t applied_value, float* const* vmax = nullptr)
    {
        ReduceAdamArgs ra{};
        ra.nseg = (int)net.L.size(); ra.inst_part_stride = inst_stride;
        for (size_t l = 0; l < net.L.size(); ++l) {
            const DenseLayer& L = net.L[l];
            const size_t nfl = (size_t)L.Kp * L.Np + L.Np;
            ra.seg[l] = DenseReduceSeg{part + off[l], nfl, std::min(chunks[l], std::max(1, Bn / (small_gemm ? 256 : 64))), (unsigned)(L.w / 4), (unsigned)(nfl / 4)};
        }
        for (int i = 0; i < ninst; ++i) { ra.p[i] = p[i]; ra.g[i] = g[i]; ra.m[i] = m[i]; ra.v[i] = v[i]; ra.tgt[i] = tgt_p ? tgt_p[i] : nullptr; ra.s[i] = sc[i]; ra.vmax[i] = vmax ? vmax[i] : nullptr; }
        ra.n4 = (unsigned)(net.total / 4); ra.track = tgt_p ? 1 : 0; ra.tau = (float)cfg.tau; ra.omt = (float)(1.0 - cfg.tau);
        ra.poison = dev_err + ERR_GATE;
        ra.applied = applied + applied_slot; ra.step = applied_value;
        BDR_HIP(step_launch(stream, true, k_dense_reduce_adam, dim3((ra.n4 + 255) / 256, ninst), dim3(256), ra));
        return BDR_OK;
    }

    // One iteration of the Sac::opt_ loop on a device-resident batch (obs/next_obs/act rows are f32).  The order of the reference is
    // kept where it matters (actor first, the critic target from the UPDATED actor, tracking after every critic step); launches that
    // do not depend on each other are merged, the narrow layers ride in row-block kernels (sac_fused.hpp) and the two wide layers of a forward pass share a launch (dense_chain.hpp): 17 launches for twin
    // critics instead of one per layer and tensor (70).  update() = prologue() + update_rest(); the two-queue sequence (opt_enqueue) runs
    // the prologue of the NEXT update on the side queue.
    // draw_noise: z_actor / z_next ([Bn][A] each, contiguous) are drawn on the device inside the first launch
    int32_t update(int Bn, const float* obs, const float* act, const float* next_obs, const float* reward, const int8_t* term,
                   float* z_actor, float* z_next, bool first, bool draw_noise = false, const GatherArgs* gather = nullptr)
    {
        BDR_TRY(ensure_batch(Bn));
        main_ahead = true;
        BDR_TRY(prologue(stream, Bn, obs, act, next_obs, z_actor, z_next, draw_noise, gather));
        return update_rest(Bn, reward, term, z_actor, z_next, first, false);
    }
    // The part of an update that needs nothing of the previous one but its actor step: pack (+ sample, + noise) and the actor's
    // forward on obs, up to the action in the critic input and its log-probability.
    int32_t prologue(hipStream_t st, int Bn, const float* obs, const float* act, const float* next_obs, float* z_actor, float* z_next,
                     bool draw_noise, const GatherArgs* gather)
    {
        bdr_agent* a = this;
        const int Kq = qn.L[0].Kp;
        {
            SacPackArgs p{};
            p.obs = obs; p.next = next_obs; p.act = act; p.O = O; p.A = A; p.B = Bn; p.x_o = x_o; p.x_no = x_no; p.ldp = pi.L[0].Kp;
            p.xq_a = xq_a; p.xq_n = xq_n; p.xq_c = xq_c; p.ldq = Kq;
            if (draw_noise) {
                BDR_REQUIRE(z_next == z_actor + (size_t)Bn * A, "noise buffers must be contiguous");
                p.z = z_actor; p.nz = (size_t)2 * Bn * A; p.seed = cfg.seed; p.counter = noise_counter;
                noise_counter += p.nz;
            }
            if (gather) { p.do_gather = 1; p.g = *gather; }
            Bracket br(a, "pack");
            BDR_HIP(step_launch(st, draw_noise || gather != nullptr, k_sac_pack, dim3((unsigned)((Bn + SAC_PACK_ROWS - 1) / SAC_PACK_ROWS)), dim3(256), p));
        }
        // ---------------- update_actor (sac/base.rs:151-167) ----------------
        (void)z_next;
        return action_logp(x_o, z_actor, Bn, true, xq_a, st);
    }
    // signal_pi: publish "the actor step of this update is complete" (flag PI, two-queue sequence) once the actor's Adam launch has ended
    int32_t update_rest(int Bn, const float* reward, const int8_t* term, float* z_actor, float* z_next, bool first, bool signal_pi)
    {
        bdr_agent* a = this;
        const int L = (int)qn.L.size(), ldq = qn.L[L - 1].Np, Ap = pi.L[n_trunk].Np, Kq = qn.L[0].Kp;
        const bool fz = fused();
        // the batch-wide parts of k_sac_q_last / k_sac_td_last ride in the launch behind them (needs one: a critic of three layers or more)
        const bool tail_defer = fz && tail_next && L - 2 >= 1 && std::max(3, NC) * ((Bn + 31) / 32) <= SAC_TAIL_LDS;
        SacTailArgs sel_tail{}, td_tail{};
        {   // Q_i(obs, a_pi) for the actor loss and Q_i(obs, act) for the TD loss: the same parameters (the critics only step at the end)
            const float* params[8]; const float* x[8]; std::vector<float*>* acts[8];
            for (int i = 0; i < NC; ++i) { params[i] = q_p[i]; x[i] = xq_a; acts[i] = &c_act[i]; params[NC + i] = q_p[i]; x[NC + i] = xq_c; acts[NC + i] = &c2_act[i]; }
            BDR_TRY(critic_forward_n(2 * NC, params, x, acts, Bn, fz ? L - 1 : -1));
        }
        if (fz) {   // last layer of all 2 NC passes + qmin selection + d qmin / d h2 + (last workgroup) EntCoef::update, actor loss
            const DenseLayer& ll = qn.L[L - 1];
            SacQLastArgs p{};
            p.npairs = 2 * NC; p.NC = NC;
            for (int i = 0; i < NC; ++i) {
                p.hin[i] = c_act[i][L - 2]; p.hin[NC + i] = c2_act[i][L - 2];
                p.wl[i] = p.wl[NC + i] = HeadRef{q_p[i] + ll.w, q_p[i] + ll.b, ll.relu};
                p.q[i] = c_act[i][L - 1]; p.q[NC + i] = c2_act[i][L - 1];
                p.dout[i] = c_dy[i][L - 1]; p.dh[i] = c_dy[i][L - 2]; p.w_last[i] = q_p[i] + ll.w;
            }
            p.ldh = ll.Kp; p.kred = ll.Kp; p.w_ld = ll.Np; p.ldq = ldq; p.B = Bn;
            p.part = lrow; p.ticket = tickets; p.logp = logp; p.log_alpha = log_alpha; p.out = scal + 1; p.scale = 1.0f / (float)Bn; p.accumulate = first ? 0 : 1;
            p.auto_alpha = cfg.ent_coef_auto ? 1 : 0;
            if (cfg.ent_coef_auto) {
                step_al += 1;
                p.target = (float)cfg.target_entropy; p.log_alpha_rw = log_alpha; p.al_m = al_m; p.al_v = al_v; p.poison = dev_err + ERR_GATE;
                p.applied = applied + AP_AL; p.step = step_al;
                p.s = adam_scalars_for(false, cfg.ent_coef_lr, 0, 0, 0, 0, step_al);
            }
            p.tail_here = tail_defer ? 0 : 1;
            if (tail_defer) {   // EntCoef::update + the actor loss: one more workgroup of the layer-(L-2) input-gradient launch below (the next reader of log_alpha is k_sac_actor_bwd)
                sel_tail.kind = 1; sel_tail.part = lrow; sel_tail.nb = (Bn + 31) / 32; sel_tail.NC = NC; sel_tail.B = Bn;
                sel_tail.log_alpha = log_alpha; sel_tail.out = p.out; sel_tail.scale = p.scale; sel_tail.accumulate = p.accumulate;
                sel_tail.auto_alpha = p.auto_alpha; sel_tail.log_alpha_rw = p.log_alpha_rw; sel_tail.al_m = p.al_m; sel_tail.al_v = p.al_v; sel_tail.s = p.s;
                sel_tail.poison = p.poison; sel_tail.applied = p.applied; sel_tail.step = p.step;
            }
            Bracket br(a, "sac_q_last");
            BDR_HIP(step_launch(stream, cfg.ent_coef_auto != 0, k_sac_q_last, dim3((Bn + 31) / 32, ll.Kp / 64 + 1), dim3(512), p));
        } else {
            SacSelectArgs p{};
            for (int i = 0; i < NC; ++i) { p.q[i] = c_act[i][L - 1]; p.dout[i] = c_dy[i][L - 1]; }
            p.ldq = ldq; p.logp = logp; p.log_alpha = log_alpha; p.B = Bn; p.NC = NC;
            p.out = scal + 1; p.scale = 1.0f / (float)Bn; p.accumulate = first ? 0 : 1;
            p.auto_alpha = cfg.ent_coef_auto ? 1 : 0;
            if (cfg.ent_coef_auto) {
                step_al += 1;
                p.target = (float)cfg.target_entropy; p.log_alpha_rw = log_alpha; p.al_m = al_m; p.al_v = al_v; p.poison = dev_err + ERR_GATE;
                p.applied = applied + AP_AL; p.step = step_al;
                p.s = adam_scalars_for(false, cfg.ent_coef_lr, 0, 0, 0, 0, step_al);
            }
            Bracket br(a, "sac_select");
            BDR_HIP(step_launch(stream, cfg.ent_coef_auto != 0, k_sac_select, dim3(1), dim3(1024), p));
        }
        if (probe_copy) {   // parity probes (bdr_sac_probe 5 / 6): the critic phase reuses these buffers
            if (probe_B < Bn) {
                for (int i = 0; i < NC; ++i) { (void)hipFree(pr_qpi[i]); pr_qpi[i] = nullptr; BDR_HIP(hipMalloc((void**)&pr_qpi[i], (size_t)Bn * ldq * 4)); }
                (void)hipFree(pr_logp); pr_logp = nullptr; BDR_HIP(hipMalloc((void**)&pr_logp, (size_t)Bn * 4));
                probe_B = Bn;
            }
            for (int i = 0; i < NC; ++i) BDR_HIP(hipMemcpyAsync(pr_qpi[i], c_act[i][L - 1], (size_t)Bn * ldq * 4, hipMemcpyDeviceToDevice, stream));
            BDR_HIP(hipMemcpyAsync(pr_logp, logp, (size_t)Bn * 4, hipMemcpyDeviceToDevice, stream));
        }
        for (int l = fz ? L - 2 : L - 1; l >= (fz ? 1 : 0); --l) {   // d qmin / d input through the critics (weights untouched here)
            Bracket br(a, "q_dx");
            BDR_TRY(critic_dx_all(l, Bn, c_act, tail_defer && l == L - 2 ? &sel_tail : nullptr, cfg.ent_coef_auto != 0));
        }
        if (fz) {   // d qmin / d a (first layer, action columns) + tanh-Gaussian backward + both heads' input gradient
            const DenseLayer &l0 = qn.L[0], &hm = pi.L[n_trunk], &hs = pi.L[n_trunk + 1];
            SacActorBwdArgs p{};
            p.NC = NC;
            for (int i = 0; i < NC; ++i) { p.dy0[i] = c_dy[i][0]; p.w0[i] = q_p[i] + l0.w; }
            p.ldy0 = l0.Np; p.w0_ld = l0.Np; p.kred0 = l0.Np; p.n0a = (O / 32) * 32; p.col0 = O;
            p.a = a_s; p.s = s_s; p.sd = sd_s; p.ld = Ap; p.z = z_actor; p.log_alpha = log_alpha; p.gmean = gmean; p.ge = ge;
            p.B = Bn; p.A = A; p.lo = (float)cfg.min_lstd; p.hi = (float)cfg.max_lstd; p.eps = (float)cfg.epsilon;
            p.wm = pi_p + hm.w; p.ws = pi_p + hs.w; p.wh_ld = hm.Np; p.kredh = hm.Np;
            p.hmask = t_act[n_trunk - 1]; p.ldh = hm.Kp; p.dh = t_dy[n_trunk - 1];
            Bracket br(a, "sac_actor_bwd");
            BDR_HIP(step_launch(stream, false, k_sac_actor_bwd, dim3((Bn + 31) / 32, hm.Kp / 32), dim3(512), p));
        } else {
            SacActorGradArgs p{};
            p.a = a_s; p.s = s_s; p.sd = sd_s; p.ld = Ap; p.z = z_actor; p.ldq = Kq; p.col0 = O; p.NC = NC;
            for (int i = 0; i < NC; ++i) p.dxq[i] = dxq[i];
            p.log_alpha = log_alpha; p.gmean = gmean; p.ge = ge; p.B = Bn; p.A = A;
            p.lo = (float)cfg.min_lstd; p.hi = (float)cfg.max_lstd; p.eps = (float)cfg.epsilon;
            Bracket br(a, "sac_actor_grad");
            BDR_HIP(step_launch(stream, false, k_sac_actor_grad, dim3((Bn * Ap + 255) / 256), dim3(256), p));
        }
        {   // input gradients down the trunk first, then every weight gradient of the actor in one grouped launch
            DenseDwJob jobs[RA_SEGS];
            auto chunks_rt = [&](int c) { return std::min(c, std::max(1, Bn / (small_gemm ? 256 : 64))); };
            const DenseSrc hin = n_trunk ? DenseSrc{t_act[n_trunk - 1], pi.L[n_trunk - 1].Np} : DenseSrc{x_o, pi.L[0].Kp};
            if (n_trunk) {
                float* dh = t_dy[n_trunk - 1];
                if (!fz) {   // (fused: k_sac_actor_bwd has written dh)
                    { Bracket br(a, "pi_dx"); BDR_TRY(dense_dx(stream, pi.L[n_trunk], pi_p, gmean, dh, t_act[n_trunk - 1], Bn, false, small_gemm)); }
                    { Bracket br(a, "pi_dx"); BDR_TRY(dense_dx(stream, pi.L[n_trunk + 1], pi_p, ge, dh, t_act[n_trunk - 1], Bn, true, small_gemm)); }
                }
                for (int l = n_trunk - 1; l > 0; --l) { Bracket br(a, "pi_dx"); BDR_TRY(dense_dx(stream, pi.L[l], pi_p, t_dy[l], t_dy[l - 1], t_act[l - 1], Bn, false, small_gemm)); }
            }
            int nj = 0;
            for (int l = 0; l < n_trunk; ++l)
                jobs[nj++] = DenseDwJob{&pi.L[l], l == 0 ? DenseSrc{x_o, pi.L[0].Kp} : DenseSrc{t_act[l - 1], pi.L[l - 1].Np}, t_dy[l], pi_part + pi_off[l], chunks_rt(pi_chunks[l])};
            jobs[nj++] = DenseDwJob{&pi.L[n_trunk], hin, gmean, pi_part + pi_off[n_trunk], chunks_rt(pi_chunks[n_trunk])};
            jobs[nj++] = DenseDwJob{&pi.L[n_trunk + 1], hin, ge, pi_part + pi_off[n_trunk + 1], chunks_rt(pi_chunks[n_trunk + 1])};
            { Bracket br(a, "pi_dw"); BDR_TRY(small_gemm ? dense_dw_small_group(stream, jobs, nj, Bn) : dense_dw_group(stream, jobs, nj, Bn)); }
            step_pi += 1;
            const AdamScalars sc = opt_scalars(cfg.opt_actor, cfg.lr_actor, step_pi);   // Actor::backward_step -> opt.rs:74-83
            Bracket br(a, "adam_pi");
            BDR_TRY(reduce_adam(pi, pi_off, pi_chunks, Bn, pi_part, 0, 1, &pi_p, &pi_g, &pi_m, &pi_v, nullptr, &sc, AP_PI, step_pi, &pi_vmax));
        }

        // ---------------- update_critic (sac/base.rs:107-149) ----------------
        // the UPDATED actor; its first launch starts when the actor's Adam step is complete and says so (flag PI)
        BDR_TRY(action_logp(x_no, z_next, Bn, false, xq_n, stream, signal_pi ? sig + SIG_PI : nullptr, epoch));
        {
            const float* params[4]; const float* x[4]; std::vector<float*>* acts[4];
            for (int i = 0; i < NC; ++i) { params[i] = q_t[i]; x[i] = xq_n; acts[i] = &c_act[i]; }
            BDR_TRY(critic_forward_n(NC, params, x, acts, Bn, fz ? L - 1 : -1));
        }
        if (fz) {   // target critics' last layer + TD target + critic losses + d loss / d h2 + (last workgroup) the loss sums
            const DenseLayer& ll = qn.L[L - 1];
            SacTdLastArgs p{};
            p.NC = NC;
            for (int i = 0; i < NC; ++i) {
                p.hin_t[i] = c_act[i][L - 2]; p.wl_t[i] = HeadRef{q_t[i] + ll.w, q_t[i] + ll.b, ll.relu}; p.qt[i] = c_act[i][L - 1];
                p.q[i] = c2_act[i][L - 1]; p.hin[i] = c2_act[i][L - 2]; p.w_last[i] = q_p[i] + ll.w;
                p.dout[i] = c_dy[i][L - 1]; p.dh[i] = c_dy[i][L - 2];
            }
            p.ldh = ll.Kp; p.kred = ll.Kp; p.w_ld = ll.Np; p.ldq = ldq;
            p.logp = logp; p.log_alpha = log_alpha; p.reward = reward; p.term = term; p.gamma = (float)cfg.gamma; p.reward_scale = (float)cfg.reward_scale;
            p.tgt = tgt; p.part = lrow + (size_t)3 * ((Bn + 31) / 32); p.B = Bn; p.loss_kind = cfg.critic_loss;
            p.ticket = tickets + 1; p.out = scal; p.scale = 1.0f / ((float)Bn * (float)NC); p.accumulate = first ? 0 : 1;
            p.tail_here = tail_defer ? 0 : 1;
            if (tail_defer) {   // the critics' loss sums (recorded only): one more workgroup of the launch below
                td_tail.kind = 2; td_tail.part = p.part; td_tail.nb = (Bn + 31) / 32; td_tail.NC = NC; td_tail.B = Bn;
                td_tail.out = p.out; td_tail.scale = p.scale; td_tail.accumulate = p.accumulate;
            }
            Bracket br(a, "sac_td_last");
            BDR_HIP(step_launch(stream, false, k_sac_td_last, dim3((Bn + 31) / 32, ll.Kp / 64), dim3(512), p));
        } else {
            SacTdArgs p{};
            for (int i = 0; i < NC; ++i) { p.q[i] = c2_act[i][L - 1]; p.dout[i] = c_dy[i][L - 1]; }
            for (int i = 0; i < NC; ++i) p.qt[i] = c_act[i][L - 1];
            p.logp = logp; p.log_alpha = log_alpha; p.reward = reward; p.term = term; p.gamma = (float)cfg.gamma; p.reward_scale = (float)cfg.reward_scale;
            p.ldq = ldq; p.NC = NC; p.tgt = tgt; p.out = scal; p.scale = 1.0f / ((float)Bn * (float)NC); p.accumulate = first ? 0 : 1;
            p.B = Bn; p.loss_kind = cfg.critic_loss;
            Bracket br(a, "critic_td");
            BDR_HIP(step_launch(stream, false, k_sac_critic_td, dim3(1), dim3(1024), p));
        }
        for (int l = fz ? L - 2 : L - 1; l > 0; --l) { Bracket br(a, "q_dx"); BDR_TRY(critic_dx_all(l, Bn, c2_act, tail_defer && l == L - 2 ? &td_tail : nullptr, false)); }
        {   // every weight gradient of every critic in one grouped launch; partial sums -> gradients, Adam and soft_update (:169-173) in one more
            std::vector<DenseDwJob> jobs;
            for (int i = 0; i < NC; ++i)
                for (int l = 0; l < L; ++l)
                    jobs.push_back(DenseDwJob{&qn.L[l], l == 0 ? DenseSrc{xq_c, Kq} : DenseSrc{c2_act[i][l - 1], qn.L[l - 1].Np}, c_dy[i][l],
                                              q_part + (size_t)i * q_part_stride + q_off[l], std::min(q_chunks[l], std::max(1, Bn / (small_gemm ? 256 : 64)))});
            { Bracket br(a, "q_dw"); BDR_TRY(small_gemm ? dense_dw_small_group(stream, jobs.data(), (int)jobs.size(), Bn) : dense_dw_group(stream, jobs.data(), (int)jobs.size(), Bn)); }
            AdamScalars sc[4];
            for (int i = 0; i < NC; ++i) { step_q[i] += 1; sc[i] = opt_scalars(cfg.opt_critic, cfg.lr_critic, step_q[i]); }
            Bracket br(a, "adam_q_track");
            BDR_TRY(reduce_adam(qn, q_off, q_chunks, Bn, q_part, q_part_stride, NC, q_p, q_g, q_m, q_v, q_t, sc, AP_Q, step_q[0], q_vmax));
        }
        n_opts += 1;
        last_B = Bn;
        return BDR_OK;
    }

    int32_t gen_noise(float* dst, size_t n)
    {
        BDR_HIP(step_launch(stream, true, k_randn, dim3((unsigned)((n + 255) / 256)), dim3(256), dst, n, cfg.seed, noise_counter));
        noise_counter += n;
        return BDR_OK;
    }

    const char* kind() const override { return "sac"; }
    void drain_queues() override
    {
        (void)hipStreamSynchronize(stream);
        if (side) (void)hipStreamSynchronize(side);   // prologue kernels / flag waits still queued there (they return at once while poisoned)
    }
    void on_gate_timeout() override   // a flag wait timed out: back to one queue
    {
        if (two_queues) fprintf(stderr, "border_amd: a cross-queue wait of the SAC step timed out; this agent continues on one queue\n");
        two_queues = false;
        main_ahead = true;   // whatever ran on `stream` since the failed wait is not covered by the flags
        // The optimizer passes that ran while the poison word was up left parameters and moments alone (k_dense_reduce_adam, the EntCoef
        // step): the host's step numbers go back to the passes that were applied, so the bias corrections of the next update are those of
        // the state on the device (opt.rs:74-83; each optimizer on its own - a time-out between the actor's and the critics' pass of one
        // update leaves the actor one step ahead, as it is on the device).  n_opts and the noise stream follow the critics' pass, the
        // last one of an update.
        unsigned long long ap[3] = {0, 0, 0};
        if (applied && hipMemcpy(ap, applied, sizeof ap, hipMemcpyDeviceToHost) == hipSuccess) {
            const uint64_t lost = step_q[0] > ap[AP_Q] ? step_q[0] - ap[AP_Q] : 0;
            if (cfg.ent_coef_auto && ap[AP_AL] < step_al) step_al = ap[AP_AL];
            if (ap[AP_PI] < step_pi) step_pi = ap[AP_PI];
            for (int i = 0; i < NC; ++i) if (ap[AP_Q] < step_q[i]) step_q[i] = ap[AP_Q];
            n_opts -= std::min<uint64_t>(n_opts, lost);
            noise_counter -= std::min<uint64_t>(noise_counter, lost * 2ull * (uint64_t)last_B * (uint64_t)A);
            if (lost) fprintf(stderr, "border_amd: %llu SAC update(s) behind the failed wait were skipped on the device; the step counters were rolled back with them\n", (unsigned long long)lost);
        }
    }
    // the launch sequence of one opt() (Sac::opt_, sac/base.rs:175-192)
    // does this opt() take the two-queue sequence?
    bool side_queue_for(const bdr_replay* r) const { return two_queues && !prof && graph_policy.mode != 1 && gather_in_pack && !r->per && !r->frame_stack && !r->index_rng; }   // (BDR_STEP_GRAPH=1: the captured step, one queue)
    int32_t opt_enqueue(bdr_replay* r, int Bn)
    {
        const bool tq = side_queue_for(r) && step_graph_current() == nullptr;
        for (uint64_t u = 0; u < cfg.n_updates_per_opt; ++u) {
            if (tq) {
                if (main_ahead) {   // once after anything the flags do not cover
                    BDR_HIP(hipEventRecord(ev_main, stream));
                    BDR_HIP(hipStreamWaitEvent(side, ev_main, 0));
                    main_ahead = false;
                }
                // the other buffer set: the previous update's critic phase is still reading this one (and the buffer's batch arrays)
                flip();
                BDR_TRY(replay_flip_batch(r, Bn));
                epoch += 1;
                GatherArgs plan{};
                BDR_TRY(flag_wait(side, sig + SIG_PI, epoch - 1, dev_err + ERR_GATE, 1u + SIG_PI, flag_limit));   // the actor parameters of update n-1 are final
                BDR_TRY(replay_sample_plan(r, Bn, side, &plan));
                BDR_TRY(prologue(side, Bn, (const float*)r->b_obs, (const float*)r->b_act, (const float*)r->b_next, z_a, z_a + (size_t)Bn * A, true, &plan));
                if ((long long)epoch != stall_at) BDR_TRY(flag_set(side, sig + SIG_PRO, epoch));
                // the main queue waits for the prologue: inside the first launch of update_rest where that is the two-layer critic launch (dense_chain.hpp)
                if (wait_in_kernel && chain2 && small_gemm && fused() && qn.L.size() == 3 && dense_chain2_ok(qn.L[0], qn.L[1]))
                    pending_wait = ChainWait{sig + SIG_PRO, epoch, flag_limit, dev_err + ERR_GATE, 1u + SIG_PRO};
                else BDR_TRY(flag_wait(stream, sig + SIG_PRO, epoch, dev_err + ERR_GATE, 1u + SIG_PRO, flag_limit));
                BDR_TRY(update_rest(Bn, r->b_reward, r->b_term, z_a, z_a + (size_t)Bn * A, u == 0, true));
                continue;
            }
            // a uniform sample over the plain ring is drawn by the pack kernel itself (replay_sample_plan); prioritized and
            // single-frame buffers keep their own gather launch
            GatherArgs plan{};
            const bool in_pack = gather_in_pack && !r->per && !r->frame_stack && !r->index_rng;
            if (in_pack) BDR_TRY(replay_sample_plan(r, Bn, stream, &plan));
            else { Bracket br(this, "sample"); BDR_TRY(replay_sample_on_stream(r, Bn, stream)); }
            // the noise of the actor pass, then of the target pass: 2*Bn*A consecutive draws of the agent's stream
            BDR_TRY(update(Bn, (const float*)r->b_obs, (const float*)r->b_act, (const float*)r->b_next, r->b_reward, r->b_term, z_a, z_a + (size_t)Bn * A, u == 0, true,
                           in_pack ? &plan : nullptr));
        }
        return BDR_OK;
    }
    int32_t opt(bdr_replay* r) override
    {
        BDR_REQUIRE(r->obs_bytes == (uint64_t)O * 4 && r->act_bytes == (uint64_t)A * 4, "replay rows do not match SAC obs/act dims");
        BDR_REQUIRE(r->device == device, "agent and replay buffer live on different devices");
        const int Bn = (int)cfg.batch_size;
        BDR_TRY(ensure_batch(Bn));
        // 17 kernels of 4-14 us.  Default: eager launches on two queues (opt_enqueue).  Without the side queue the sequence can be
        // replayed from a captured graph (step_graph.hpp; the policy measures whether that pays).  Profiling brackets and prioritized
        // replay (tree kernels with their own host state) take the eager one-queue path - the same sequence, launched one by one.
        if (prof || r->per || side_queue_for(r)) return opt_enqueue(r, Bn);
        {
            const int w = graph_policy.want(stream);
            if (w < 0) return fail(BDR_ERR_HIP, "hipStreamQuery failed");
            if (w == 0) return opt_enqueue(r, Bn);
        }
        BDR_TRY(replay_prepare_sample(r, Bn, stream));
        // host state the pass advances, put back if the step has to be enqueued a second time (step_graph_run)
        const ReplaySnap rs(r);
        const uint64_t s_pi = step_pi, s_al = step_al, s_noise = noise_counter, s_opts = n_opts;
        uint64_t s_q[4]; memcpy(s_q, step_q, sizeof s_q);
        return step_graph_run(&graph, stream, r->uid, r->batch_gen, batch_gen, [&]() { return opt_enqueue(r, Bn); },
                              [&]() { rs.restore(r); step_pi = s_pi; step_al = s_al; noise_counter = s_noise; n_opts = s_opts; memcpy(step_q, s_q, sizeof s_q); });
    }
    void record_keys(std::vector<std::string>& keys) override { keys = {"loss_critic", "loss_actor", "ent_coef"}; }
    int32_t noise(float* dev, size_t n) override { return gen_noise(dev, n); }   // the N(0,1) stream of action_logp
    int32_t record(float* out, int cap, int* n) override
    {
        float h[2], la;
        BDR_HIP(hipMemcpyAsync(h, scal, 8, hipMemcpyDeviceToHost, stream));
        BDR_HIP(hipMemcpyAsync(&la, log_alpha, 4, hipMemcpyDeviceToHost, stream));
        BDR_HIP(hipStreamSynchronize(stream));
        const float nu = (float)cfg.n_updates_per_opt;
        if (cap < 3) return fail(BDR_ERR_INVALID, "SAC record needs 3 slots");
        out[0] = h[0] / nu; out[1] = h[1] / nu; out[2] = expf(la);   // loss_critic, loss_actor, ent_coef
        *n = 3;
        return BDR_OK;
    }

    // which: 0 pi, 1+i qnet_i, 1+NC+i qnet_tgt_i, 1+2NC log_alpha;  +100 grad, +200 exp_avg, +300 exp_avg_sq, +400 max_exp_avg_sq (AdamW{amsgrad})
    struct Slot { float* p; const MlpLayout* lay; size_t n; };
    Slot slot(int which)
    {
        const int role = which / 100, id = which % 100;
        if (id == 0) { float* r[5] = {pi_p, pi_g, pi_m, pi_v, pi_vmax}; return role < 5 && r[role] ? Slot{r[role], &pi, pi.total} : Slot{nullptr, nullptr, 0}; }
        if (id >= 1 && id <= NC) { const int i = id - 1; float* r[5] = {q_p[i], q_g[i], q_m[i], q_v[i], q_vmax[i]}; return role < 5 && r[role] ? Slot{r[role], &qn, qn.total} : Slot{nullptr, nullptr, 0}; }
        if (id >= 1 + NC && id <= 2 * NC && role == 0) return Slot{q_t[id - 1 - NC], &qn, qn.total};
        if (id == 1 + 2 * NC) { float* r[4] = {log_alpha, nullptr, al_m, al_v}; return (role < 4 && r[role]) ? Slot{r[role], nullptr, 4} : Slot{nullptr, nullptr, 0}; }
        return Slot{nullptr, nullptr, 0};
    }
    uint64_t param_count(int which) override
    {
        Slot s = slot(which);
        if (!s.p) return 0;
        return s.lay ? s.lay->ref_total : 1;
    }
    int32_t get_params(int which, float* out, uint64_t n) override
    {
        Slot s = slot(which);
        BDR_REQUIRE(s.p, "unknown SAC model %d", which);
        BDR_REQUIRE(n == param_count(which), "parameter count mismatch (%llu vs %llu)", (unsigned long long)n, (unsigned long long)param_count(which));
        std::vector<float> in(s.n);
        BDR_HIP(hipMemcpyAsync(in.data(), s.p, s.n * 4, hipMemcpyDeviceToHost, stream));
        BDR_HIP(hipStreamSynchronize(stream));
        if (s.lay) mlp_to_reference(*s.lay, 0, in.data(), out); else out[0] = in[0];
        return BDR_OK;
    }
    int32_t set_params(int which, const float* inp, uint64_t n) override
    {
        Slot s = slot(which);
        BDR_REQUIRE(s.p, "unknown SAC model %d", which);
        BDR_REQUIRE(n == param_count(which), "parameter count mismatch");
        std::vector<float> in(s.n, 0.f);
        if (s.lay) mlp_to_internal(*s.lay, 0, inp, in.data()); else in[0] = inp[0];
        BDR_HIP(hipMemcpyAsync(s.p, in.data(), s.n * 4, hipMemcpyHostToDevice, stream));
        BDR_HIP(hipStreamSynchronize(stream));
        return BDR_OK;
    }
    // SyncModel ships only `pi` (sac/base.rs:377-386)
    float* arena(int which, size_t* n) override
    {
        main_ahead = true;   // the caller (a parameter exchange) enqueues work on `stream` the flags know nothing about
        Slot s = slot(which); if (n) *n = s.n; return s.p;
    }

    std::vector<NamedTensor> pi_meta() const
    {
        std::vector<NamedTensor> mt;
        for (int i = 0; i < n_trunk; ++i) {
            mt.push_back({"mlp.al" + std::to_string(i) + ".weight", {(uint64_t)pi.L[i].out, (uint64_t)pi.L[i].in}});
            mt.push_back({"mlp.al" + std::to_string(i) + ".bias", {(uint64_t)pi.L[i].out}});
        }
        const auto& h = pi.L[n_trunk];
        mt.push_back({"ml.weight", {(uint64_t)h.out, (uint64_t)h.in}}); mt.push_back({"ml.bias", {(uint64_t)h.out}});
        mt.push_back({"sl.weight", {(uint64_t)h.out, (uint64_t)h.in}}); mt.push_back({"sl.bias", {(uint64_t)h.out}});
        return mt;
    }
    std::vector<NamedTensor> q_meta() const
    {
        std::vector<NamedTensor> mt;
        for (size_t i = 0; i < qn.L.size(); ++i) {
            mt.push_back({"mlp.ln" + std::to_string(i) + ".weight", {(uint64_t)qn.L[i].out, (uint64_t)qn.L[i].in}});
            mt.push_back({"mlp.ln" + std::to_string(i) + ".bias", {(uint64_t)qn.L[i].out}});
        }
        return mt;
    }
    int32_t save(const char* dir) override   // sac/base.rs:313-328: qnet_{i}, qnet_tgt_{i}, pi, ent_coef
    {
        std::vector<float> v(pi.ref_total);
        BDR_TRY(get_params(0, v.data(), v.size()));
        BDR_TRY(save_named(ckpt_save_path(this, dir, "pi"), pi_meta(), v.data(), v.size()));
        v.resize(qn.ref_total);
        for (int i = 0; i < NC; ++i) {
            BDR_TRY(get_params(1 + i, v.data(), v.size()));
            BDR_TRY(save_named(ckpt_save_path(this, dir, "qnet_" + std::to_string(i)), q_meta(), v.data(), v.size()));
            BDR_TRY(get_params(1 + NC + i, v.data(), v.size()));
            BDR_TRY(save_named(ckpt_save_path(this, dir, "qnet_tgt_" + std::to_string(i)), q_meta(), v.data(), v.size()));
        }
        float la = 0;
        BDR_TRY(get_params(1 + 2 * NC, &la, 1));
        return save_named(ckpt_save_path(this, dir, "ent_coef"), {{"log_alpha", {1}}}, &la, 1);
    }
    int32_t load(const char* dir) override
    {
        std::vector<float> v(pi.ref_total);
        BDR_TRY(load_named(ckpt_load_path(this, dir, "pi"), pi_meta(), v.data(), v.size()));
        BDR_TRY(set_params(0, v.data(), v.size()));
        v.resize(qn.ref_total);
        for (int i = 0; i < NC; ++i) {
            BDR_TRY(load_named(ckpt_load_path(this, dir, "qnet_" + std::to_string(i)), q_meta(), v.data(), v.size()));
            BDR_TRY(set_params(1 + i, v.data(), v.size()));
            BDR_TRY(load_named(ckpt_load_path(this, dir, "qnet_tgt_" + std::to_string(i)), q_meta(), v.data(), v.size()));
            BDR_TRY(set_params(1 + NC + i, v.data(), v.size()));
        }
        float la = 0;
        BDR_TRY(load_named(ckpt_load_path(this, dir, "ent_coef"), {{"log_alpha", {1}}}, &la, 1));
        return set_params(1 + 2 * NC, &la, 1);
    }
};

extern "C" {

void bdr_sac_config_default(bdr_sac_config* c)
{
    if (!c) return;
    memset(c, 0, sizeof *c);
    // sac/config.rs:85-105
    c->gamma = 0.99; c->tau = 0.005; c->ent_coef_auto = 0; c->ent_coef_alpha = 1.0; c->epsilon = 1e-4;
    c->min_lstd = -20.0; c->max_lstd = 2.0; c->n_updates_per_opt = 1; c->batch_size = 1; c->train = 0;
    c->critic_loss = BDR_LOSS_MSE; c->reward_scale = 1.0; c->n_critics = 1; c->device = -1;
    for (bdr_adamw_config* o : {&c->opt_actor, &c->opt_critic}) { o->opt_kind = BDR_OPT_ADAM; o->beta1 = 0.9; o->beta2 = 0.999; o->weight_decay = 0.0; o->eps = 1e-8; }
}

int32_t bdr_sac_create(const bdr_sac_config* cfg, bdr_agent** out)
{
    BDR_REQUIRE(cfg && out, "null argument");
    BDR_REQUIRE(cfg->device >= 0, "No device is given for SAC agent");
    BDR_REQUIRE(cfg->obs_dim >= 1 && cfg->act_dim >= 1 && cfg->act_dim <= 64, "bad obs/act dims");
    BDR_REQUIRE(cfg->n_pi_units >= 1 && cfg->n_pi_units <= BDR_MAX_UNITS && cfg->n_q_units >= 0 && cfg->n_q_units <= BDR_MAX_UNITS, "bad layer counts");
    BDR_REQUIRE(cfg->n_critics >= 1 && cfg->n_critics <= 4, "n_critics must be in [1,4]");
    BDR_REQUIRE(cfg->batch_size >= 1 && cfg->batch_size <= 65536 && cfg->n_updates_per_opt >= 1, "bad batch / update counts");
    for (const bdr_adamw_config* o : {&cfg->opt_actor, &cfg->opt_critic}) BDR_REQUIRE(o->opt_kind == BDR_OPT_ADAM || o->opt_kind == BDR_OPT_ADAMW, "unknown optimizer");
    BDR_TRY(ensure_device(cfg->device));
    Sac* a = new Sac();
    a->cfg = *cfg; a->device = cfg->device; a->train = cfg->train != 0;
    a->O = cfg->obs_dim; a->A = cfg->act_dim; a->NC = cfg->n_critics; a->n_trunk = cfg->n_pi_units;
    // actor: trunk al0.. (ReLU) then heads ml, sl on the last hidden layer
    {
        MlpLayout t = make_mlp(a->O, cfg->pi_units, cfg->n_pi_units - 1, cfg->pi_units[cfg->n_pi_units - 1], true);
        a->pi = t;
        size_t o = t.total;
        const int h = cfg->pi_units[cfg->n_pi_units - 1];
        for (int k = 0; k < 2; ++k) {
            DenseLayer l; l.in = h; l.out = a->A; l.Kp = pad64(h); l.Np = pad64(a->A); l.w = o; o += (size_t)l.Kp * l.Np; l.b = o; o += l.Np; l.relu = 0;
            a->pi.L.push_back(l);
            a->pi.ref_total += (size_t)a->A * h + a->A;
        }
        a->pi.total = o; a->pi.out_dim = a->A;
    }
    a->qn = make_mlp(a->O + a->A, cfg->q_units, cfg->n_q_units, 1, false);
    BDR_HIP(hipStreamCreateWithFlags(&a->stream, hipStreamNonBlocking));
    BDR_TRY(a->err_init());
    a->graph_policy.from_env();
    { const char* e = getenv("BDR_NO_SMALL_GEMM"); a->small_gemm = !(e && e[0] == '1'); }
    a->gather_in_pack = getenv("BDR_NO_STEP_GATHER") == nullptr;
    a->fuse_rows = getenv("BDR_NO_SAC_FUSE") == nullptr;
    a->chain2 = getenv("BDR_NO_SAC_CHAIN") == nullptr;
    a->tail_next = getenv("BDR_SAC_TAIL_IN_KERNEL") == nullptr;
    a->wait_in_kernel = getenv("BDR_SAC_WAIT_PACKET") == nullptr;
    a->heads_in_chain = getenv("BDR_SAC_HEADS_FUSE") != nullptr;
    BDR_HIP(hipMalloc((void**)&a->head_tickets, 2 * Sac::HEAD_TICKETS * sizeof(unsigned)));
    BDR_HIP(hipMemsetAsync(a->head_tickets, 0, 2 * Sac::HEAD_TICKETS * sizeof(unsigned), a->stream));
    { const char* e = getenv("BDR_SAC_CHAIN_TPW"); a->chain2_tpw = e ? atoi(e) : 0; }
    BDR_HIP(hipMalloc((void**)&a->tickets, 2 * sizeof(unsigned))); BDR_HIP(hipMemsetAsync(a->tickets, 0, 2 * sizeof(unsigned), a->stream));
    BDR_HIP(hipMalloc((void**)&a->applied, 3 * sizeof(unsigned long long))); BDR_HIP(hipMemsetAsync(a->applied, 0, 3 * sizeof(unsigned long long), a->stream));
    {
        const char* e = getenv("BDR_SAC_SIDE_QUEUE");
        if (!(e && e[0] == '0')) {
            BDR_HIP(hipStreamCreateWithFlags(&a->side, hipStreamNonBlocking));
            BDR_HIP(hipEventCreateWithFlags(&a->ev_main, hipEventDisableTiming));
            BDR_HIP(hipMalloc((void**)&a->sig, 4 * sizeof(unsigned)));
            BDR_HIP(hipMemset(a->sig, 0, 4 * sizeof(unsigned)));
            BDR_TRY(flag_queues_independent(a->stream, a->side, a->sig + Sac::SIG_SCRATCH, &a->two_queues));
            if (const char* l = getenv("BDR_GATE_LIMIT_MS")) a->flag_limit = (unsigned long long)std::max(1, atoi(l)) * 100000ull;
            if (const char* l = getenv("BDR_SAC_STALL_AT")) a->stall_at = atoll(l);
        }
    }
    float** pis[4] = {&a->pi_p, &a->pi_g, &a->pi_m, &a->pi_v};
    for (auto p : pis) BDR_TRY(a->zalloc(p, a->pi.total));
    if (cfg->opt_actor.opt_kind == BDR_OPT_ADAMW && cfg->opt_actor.amsgrad) BDR_TRY(a->zalloc(&a->pi_vmax, a->pi.total));
    for (int i = 0; i < a->NC; ++i) {
        float** qs[5] = {&a->q_p[i], &a->q_t[i], &a->q_g[i], &a->q_m[i], &a->q_v[i]};
        for (auto p : qs) BDR_TRY(a->zalloc(p, a->qn.total));
        if (cfg->opt_critic.opt_kind == BDR_OPT_ADAMW && cfg->opt_critic.amsgrad) BDR_TRY(a->zalloc(&a->q_vmax[i], a->qn.total));
    }
    BDR_TRY(a->zalloc(&a->log_alpha, 4)); BDR_TRY(a->zalloc(&a->al_m, 4)); BDR_TRY(a->zalloc(&a->al_v, 4)); BDR_TRY(a->zalloc(&a->scal, 4));
    // initial parameters: library initialiser; critics cloned into their targets (Critic::clone)
    std::vector<float> ref(a->pi.ref_total);
    mlp_init_reference(a->pi, cfg->seed * 7 + 1, ref.data());
    BDR_TRY(a->set_params(0, ref.data(), ref.size()));
    ref.resize(a->qn.ref_total);
    for (int i = 0; i < a->NC; ++i) {
        mlp_init_reference(a->qn, cfg->seed * 7 + 2 + i, ref.data());
        BDR_TRY(a->set_params(1 + i, ref.data(), ref.size()));
        BDR_TRY(a->set_params(1 + a->NC + i, ref.data(), ref.size()));
    }
    const float la = cfg->ent_coef_auto ? 0.0f : (float)std::log(cfg->ent_coef_alpha);   // ent_coef.rs:33-41
    BDR_TRY(a->set_params(1 + 2 * a->NC, &la, 1));
    BDR_TRY(a->ensure_batch((int)cfg->batch_size));
    *out = a;
    return BDR_OK;
}

int32_t bdr_sac_update_on_batch(bdr_agent* base, uint64_t n, const float* obs, const float* act, const float* next_obs,
                                const float* reward, const int8_t* term, const float* z_actor, const float* z_next, float* rec3)
{
    BDR_REQUIRE(base && obs && act && next_obs && reward && term && z_actor && z_next, "null argument");
    BDR_REQUIRE(!strcmp(base->kind(), "sac"), "not a SAC agent");
    BDR_REQUIRE(n >= 1 && n <= 65536, "batch size out of range");
    Sac* a = static_cast<Sac*>(base);
    BDR_HIP(hipSetDevice(a->device));
    BDR_TRY(a->ensure_batch((int)n));
    if (n > a->u_cap) {
        BDR_HIP(hipStreamSynchronize(a->stream));
        (void)hipFree(a->u_obs); (void)hipFree(a->u_next); (void)hipFree(a->u_act); (void)hipFree(a->u_rew); (void)hipFree(a->u_term);
        BDR_HIP(hipMalloc((void**)&a->u_obs, n * a->O * 4)); BDR_HIP(hipMalloc((void**)&a->u_next, n * a->O * 4));
        BDR_HIP(hipMalloc((void**)&a->u_act, n * a->A * 4)); BDR_HIP(hipMalloc((void**)&a->u_rew, n * 4));
        BDR_HIP(hipMalloc((void**)&a->u_term, round_up(n, 16)));
        a->u_cap = n;
    }
    hipStream_t s = a->stream;
    BDR_HIP(hipMemcpyAsync(a->u_obs, obs, n * a->O * 4, hipMemcpyHostToDevice, s));
    BDR_HIP(hipMemcpyAsync(a->u_next, next_obs, n * a->O * 4, hipMemcpyHostToDevice, s));
    BDR_HIP(hipMemcpyAsync(a->u_act, act, n * a->A * 4, hipMemcpyHostToDevice, s));
    BDR_HIP(hipMemcpyAsync(a->u_rew, reward, n * 4, hipMemcpyHostToDevice, s));
    BDR_HIP(hipMemcpyAsync(a->u_term, term, n, hipMemcpyHostToDevice, s));
    BDR_HIP(hipMemcpyAsync(a->z_a, z_actor, n * a->A * 4, hipMemcpyHostToDevice, s));
    BDR_HIP(hipMemcpyAsync(a->z_a + n * a->A, z_next, n * a->A * 4, hipMemcpyHostToDevice, s));
    a->probe_copy = true;
    const int32_t ust = a->update((int)n, a->u_obs, a->u_act, a->u_next, a->u_rew, a->u_term, a->z_a, a->z_a + n * a->A, true);
    a->probe_copy = false;
    BDR_TRY(ust);
    prof_collect(a);
    if (rec3) {
        float h[2], la;
        BDR_HIP(hipMemcpyAsync(h, a->scal, 8, hipMemcpyDeviceToHost, s));
        BDR_HIP(hipMemcpyAsync(&la, a->log_alpha, 4, hipMemcpyDeviceToHost, s));
        BDR_HIP(hipStreamSynchronize(s));
        rec3[0] = h[0]; rec3[1] = h[1]; rec3[2] = expf(la);
    } else {
        BDR_HIP(hipStreamSynchronize(s));
    }
    return BDR_OK;
}

// Parity probes of the LAST update (see include/border_amd.h)
int32_t bdr_sac_probe(bdr_agent* base, int32_t what, float* out, uint64_t n)
{
    BDR_REQUIRE(base && out, "null argument");
    BDR_REQUIRE(!strcmp(base->kind(), "sac"), "not a SAC agent");
    Sac* a = static_cast<Sac*>(base);
    BDR_HIP(hipSetDevice(a->device));
    const int Bn = a->last_B, NC = a->NC, L = (int)a->qn.L.size(), ldq = a->qn.L[L - 1].Np;
    BDR_REQUIRE(Bn > 0, "no update has run yet");
    BDR_HIP(hipStreamSynchronize(a->stream));
    auto column0 = [&](float* const* mats, float* dst) -> int32_t {     // [NC] matrices [Bn][ldq] -> [NC][Bn]
        std::vector<float> m((size_t)Bn * ldq);
        for (int i = 0; i < NC; ++i) {
            BDR_HIP(hipMemcpy(m.data(), mats[i], m.size() * 4, hipMemcpyDeviceToHost));
            for (int b = 0; b < Bn; ++b) dst[(size_t)i * Bn + b] = m[(size_t)b * ldq];
        }
        return BDR_OK;
    };
    float* mats[4];
    switch (what) {
        case 0:   // Q_i(obs, act): the predictions of update_critic (sac/base.rs:128-131, qvals :89-98)
            BDR_REQUIRE(n == (uint64_t)NC * Bn, "q_pred holds n_critics x batch values");
            for (int i = 0; i < NC; ++i) mats[i] = a->c2_act[i][L - 1];
            return column0(mats, out);
        case 1:   // target critics on (next_obs, a' ~ pi(next_obs)) (:112-118)
            BDR_REQUIRE(n == (uint64_t)NC * Bn, "q_next holds n_critics x batch values");
            for (int i = 0; i < NC; ++i) mats[i] = a->c_act[i][L - 1];
            return column0(mats, out);
        case 2: { // qvals_min over the target critics (:100-105)
            BDR_REQUIRE(n == (uint64_t)Bn, "qvals_min holds batch values");
            std::vector<float> q((size_t)NC * Bn);
            for (int i = 0; i < NC; ++i) mats[i] = a->c_act[i][L - 1];
            BDR_TRY(column0(mats, q.data()));
            for (int b = 0; b < Bn; ++b) { float m = q[b]; for (int i = 1; i < NC; ++i) m = std::min(m, q[(size_t)i * Bn + b]); out[b] = m; }
            return BDR_OK;
        }
        case 3:   // log p(a' | next_obs) under the updated actor (:113)
            BDR_REQUIRE(n == (uint64_t)Bn, "next_log_p holds batch values");
            BDR_HIP(hipMemcpy(out, a->logp, (size_t)Bn * 4, hipMemcpyDeviceToHost));
            return BDR_OK;
        case 4:   // the TD target (:119-122)
            BDR_REQUIRE(n == (uint64_t)Bn, "tgt holds batch values");
            BDR_HIP(hipMemcpy(out, a->tgt, (size_t)Bn * 4, hipMemcpyDeviceToHost));
            return BDR_OK;
        case 5:   // Q_i(obs, a_pi) of update_actor (:157-158); bdr_sac_update_on_batch only
            BDR_REQUIRE(a->probe_B >= Bn, "q_pi is kept by bdr_sac_update_on_batch only");
            BDR_REQUIRE(n == (uint64_t)NC * Bn, "q_pi holds n_critics x batch values");
            return column0(a->pr_qpi, out);
        case 6:   // log p(a_pi | obs) of update_actor (:156); bdr_sac_update_on_batch only
            BDR_REQUIRE(a->probe_B >= Bn, "log_p is kept by bdr_sac_update_on_batch only");
            BDR_REQUIRE(n == (uint64_t)Bn, "log_p holds batch values");
            BDR_HIP(hipMemcpy(out, a->pr_logp, (size_t)Bn * 4, hipMemcpyDeviceToHost));
            return BDR_OK;
        case 7: { // a' [B][A]
            BDR_REQUIRE(n == (uint64_t)Bn * a->A, "next_act holds batch x act_dim values");
            // the target pass writes a' straight into the critic input (next_obs | a'), columns [O, O + A) of xq_n
            const int Kq = a->qn.L[0].Kp;
            std::vector<float> m((size_t)Bn * Kq);
            BDR_HIP(hipMemcpy(m.data(), a->xq_n, m.size() * 4, hipMemcpyDeviceToHost));
            for (int b = 0; b < Bn; ++b) for (int j = 0; j < a->A; ++j) out[(size_t)b * a->A + j] = m[(size_t)b * Kq + a->O + j];
            return BDR_OK;
        }
        default: return fail(BDR_ERR_INVALID, "unknown SAC probe %d", what);
    }
}

// Policy::sample (sac/base.rs:215-225): tanh(mean) in eval mode, tanh(sigma*z + mean) in train mode
// (z from the device generator); out: [n][act_dim]
int32_t bdr_sac_sample(bdr_agent* base, uint64_t n, const float* obs, float* act_out)
{
    BDR_REQUIRE(base && obs && act_out, "null argument");
    BDR_REQUIRE(!strcmp(base->kind(), "sac"), "not a SAC agent");
    Sac* a = static_cast<Sac*>(base);
    BDR_HIP(hipSetDevice(a->device));
    BDR_TRY(a->ensure_batch((int)n));
    const float* d = nullptr;   // (a hipMalloc / hipFree pair per call used to synchronise the whole device here)
    int32_t st = BDR_OK;
    if (!a->obs_rows_on_device && n * a->O * 4 <= bdr_agent::HOST_ROWS_PINNED_MAX) {   // host rows: read in place from pinned memory by the packing kernel
        const uint8_t* pd = nullptr;
        BDR_TRY(a->host_rows_pinned(obs, n * a->O * 4, &pd));
        d = reinterpret_cast<const float*>(pd);
    } else {
        float* stage = nullptr;
        BDR_TRY(a->act_buffer(n * a->O * 4, (void**)&stage));
        st = a->stage_obs(stage, obs, (size_t)a->O * 4, n, a->stream);
        d = stage;
    }
    if (st == BDR_OK) st = pack_rows(a->stream, d, a->O, a->O, a->x_o, a->pi.L[0].Kp, 0, (int)n);
    if (st == BDR_OK) {
        if (a->train) st = a->gen_noise(a->z_a, n * a->A);
        else { hipError_t e = hipMemsetAsync(a->z_a, 0, n * a->A * 4, a->stream); if (e != hipSuccess) st = fail(BDR_ERR_HIP, "memset failed"); }
    }
    if (st == BDR_OK) st = a->action_logp(a->x_o, a->z_a, (int)n, true, a->xq_a, a->stream);
    const int Ap = a->pi.L[a->n_trunk].Np;
    std::vector<float> tmp(n * Ap);
    if (st == BDR_OK) st = a->rows_to_host(a->a_s, tmp.data(), tmp.size());
    a->slot_cursor = 0;
    BDR_TRY(st);
    for (uint64_t i = 0; i < n; ++i) for (int j = 0; j < a->A; ++j) act_out[i * a->A + j] = tmp[i * Ap + j];
    return BDR_OK;
}

// bdr_sac_sample for observation rows that are already in HBM (row i at obs_dev + i * row_stride bytes)
int32_t bdr_sac_sample_device(bdr_agent* base, uint64_t n, const void* obs_dev, uint64_t row_stride, float* act_out)
{
    BDR_REQUIRE(base && obs_dev && act_out, "null argument");
    BDR_REQUIRE(!strcmp(base->kind(), "sac"), "not a SAC agent");
    BDR_REQUIRE(row_stride >= (uint64_t)static_cast<Sac*>(base)->O * 4 && row_stride % 4 == 0, "row_stride must be >= the row size and a multiple of 4");
    BDR_HIP(hipSetDevice(base->device));
    BDR_TRY(base->check_device_rows(obs_dev, row_stride));
    bdr_agent::DeviceRowsScope rows(base, row_stride);
    return bdr_sac_sample(base, n, static_cast<const float*>(obs_dev), act_out);
}

}  // extern "C"
