// Mlp-based agents on MI355X: Dqn<E, Mlp, R> (CartPole-shaped, BASELINE config 1).
// Reference: border-tch-agent/src/dqn/base.rs:60-200 (update_critic / opt_), mlp/base.rs:13-41.
// Same FP32-MFMA kernels as the CNN path, driven with runtime (64-padded) dimensions (dense.hpp).
#include <algorithm>
#include <cstdlib>

#include "dense.hpp"
#include "mlp_fused.hpp"

using namespace bdr;

namespace {

constexpr int MAXZ = 3;

// TD on dense Q rows [B][ld] (dqn/base.rs:71-74, :91-105, :146-152): one wave per row.  Writes
// dL/dQ as a dense row (zero except the taken action) so the generic dense backward can follow.
struct TdDenseArgs {
    const float* q_on; const float* q_tg; const float* q_on_next; int ld;
    const uint8_t* act; int act_bytes;
    const float* reward; const int8_t* term;
    float* dq_rows;   // [B][ld]
    float* pred; float* tgt; float* loss_row;
    int B, A; float gamma; int loss_kind;
    const float* weight; float* td_abs; int has_clip; float clip_min, clip_max;   // PER (dqn/base.rs:123-145)
    unsigned* err;    // bdr_agent::dev_err
    int relu_out;     // MlpConfig::activation_out (mlp/base.rs:36): the Q rows are post-ReLU, dL/dz = dL/dQ * [Q > 0]
};
__global__ __launch_bounds__(256) void k_td_dense(TdDenseArgs a)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    if (row >= a.B) return;
    long long act = *reinterpret_cast<const long long*>(a.act + (size_t)row * a.act_bytes);
    if (act < 0 || act >= a.A) {   // the reference's gather raises; here: flag for the host, clamp to stay in bounds
        if (lane == 0 && a.err) atomicOr(a.err + bdr_agent::ERR_ACTION, 1u);
        act = act < 0 ? 0 : a.A - 1;
    }
    const float* sel = a.q_on_next ? a.q_on_next : a.q_tg;
    // first-max argmax over the A real actions (A <= 64)
    float v = lane < a.A ? sel[(size_t)row * a.ld + lane] : -INFINITY;
    int idx = lane;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(v, off);
        const int oi = __shfl_xor(idx, off);
        if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
    }
    const float qn = a.q_tg[(size_t)row * a.ld + idx];
    const float pred = a.q_on[(size_t)row * a.ld + act];
    const float tgt = td_target(a.reward[row], (float)(1 - (int)a.term[row]), a.gamma, qn);   // dqn/base.rs:104
    const TdLossIn li{a.loss_kind, a.weight != nullptr, a.weight ? a.weight[row] : 1.f, a.has_clip, a.clip_min, a.clip_max};
    float lossb, td;
    const float dl = td_loss_row(pred, tgt, li, lossb, td);
    float dq = dl / (float)a.B;
    if (a.relu_out && !(pred > 0.f)) dq = 0.f;
    if (lane == 0) { a.pred[row] = pred; a.tgt[row] = tgt; a.loss_row[row] = lossb; if (a.td_abs) a.td_abs[row] = td; }
    for (int c = lane; c < a.ld; c += 64) a.dq_rows[(size_t)row * a.ld + c] = c == act ? dq : 0.f;
}

// loss = mean(loss_row): fixed-order tree
__global__ __launch_bounds__(256) void k_mean_rows(const float* __restrict__ x, int n, float* __restrict__ out, float scale)
{
    __shared__ float red[256];
    float s = 0.f;
    for (int b = threadIdx.x; b < n; b += 256) s += x[b];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = red[0] * scale;
}

// Row-block head of the layer-by-layer step (the SAC step's sac_fused.hpp pattern): the last layer (hidden -> A actions, narrower than
// one 32-column tile) of every network instance, the TD step of the rows (k_td_dense), the loss mean (k_mean_rows, by the last
// workgroup to finish) and the last layer's input gradient - four launches of ~4 us each at these sizes - in one.  A workgroup takes
// 32 rows; the tiles come from the layer-by-layer path's own tile function, the row arithmetic is k_td_dense's, so both paths give
// the same bits.  grid (row blocks, ldh / 64): every y forms the tiles and the TD rows (cheap) and takes 64 columns of the gradient.
struct MlpHeadTdArgs {
    int nz; const float* hin[MAXZ]; HeadRef wl[MAXZ]; float* q[MAXZ];   // instance z: last hidden activation [B][ldh], last layer, Q rows [B][ldq]
    int ldh, kred, w_ld, ldq;
    int sel_z;                                                           // double DQN: the instance whose rows choose the action (else -1)
    TdDenseArgs t;                                                       // act, reward, term, dq_rows, pred, tgt, loss_row, ...
    const float* w_last; float* dh;                                      // online last-layer weights, gradient w.r.t. hin[0] [B][ldh]
    unsigned* ticket; float* loss; float scale;
};
__global__ __launch_bounds__(512) void k_mlp_head_td(MlpHeadTdArgs p)
{
    __shared__ float red[2][4][32][33];
    __shared__ float qv[MAXZ][32][33];
    __shared__ float s_dq[32];
    __shared__ int s_act[32];
    __shared__ float mred[256];
    __shared__ unsigned s_last;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int team = wave >> 2, w4 = wave & 3;
    const int m0 = (int)blockIdx.x * 32, c0 = (int)blockIdx.y * 64;
    const bool writer = blockIdx.y == 0;
    const TdDenseArgs& a = p.t;
    for (int z0 = 0; z0 < p.nz; z0 += 2) {
        const int z = z0 + team;
        if (z < p.nz) {   // team-uniform
            const float* arow = p.hin[z] + (size_t)min(m0 + (lane & 31), a.B - 1) * p.ldh;
            dense_small_tile<false>([&](int k) { return *reinterpret_cast<const f32x4*>(arow + k); }, p.wl[z].w, p.w_ld, 0, p.kred, w4, lane, red[team]);
        }
        __syncthreads();
        for (int e = tid; e < 2 * 32 * 32; e += 512) {
            const int t = e >> 10, r = (e >> 5) & 31, c = e & 31, zz = z0 + t;
            if (zz >= p.nz) continue;
            float v = dense_small_sum(red[t], r, c) + p.wl[zz].bias[c];
            if (p.wl[zz].relu) v = v > 0.f ? v : 0.f;
            qv[zz][r][c] = v;
            if (writer && m0 + r < a.B) p.q[zz][(size_t)(m0 + r) * p.ldq + c] = v;
        }
        __syncthreads();
    }
    // ---- k_td_dense, one wave per row (lanes over the A <= 32 actions; the butterflies are the original's)
    for (int rr = 0; rr < 4; ++rr) {
        const int r = wave * 4 + rr, row = m0 + r;
        if (row >= a.B) break;
        long long act = *reinterpret_cast<const long long*>(a.act + (size_t)row * a.act_bytes);
        if (act < 0 || act >= a.A) {
            if (lane == 0 && a.err && writer) atomicOr(a.err + bdr_agent::ERR_ACTION, 1u);
            act = act < 0 ? 0 : a.A - 1;
        }
        const float (*sel)[33] = p.sel_z >= 0 ? qv[p.sel_z] : qv[1];
        float v = lane < a.A ? sel[r][lane] : -INFINITY;
        int idx = lane;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float ov = __shfl_xor(v, off);
            const int oi = __shfl_xor(idx, off);
            if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
        }
        const float qn = qv[1][r][idx];
        const float pred = qv[0][r][(int)act];
        const float tgt = td_target(a.reward[row], (float)(1 - (int)a.term[row]), a.gamma, qn);
        const TdLossIn li{a.loss_kind, a.weight != nullptr, a.weight ? a.weight[row] : 1.f, a.has_clip, a.clip_min, a.clip_max};
        float lossb, td;
        const float dl = td_loss_row(pred, tgt, li, lossb, td);
        float dq = dl / (float)a.B;
        if (a.relu_out && !(pred > 0.f)) dq = 0.f;
        if (lane == 0) {
            s_dq[r] = dq; s_act[r] = (int)act;
            if (writer) { a.pred[row] = pred; a.tgt[row] = tgt; st_agent(a.loss_row + row, lossb); if (a.td_abs) a.td_abs[row] = td; }
        }
        if (writer) for (int c = lane; c < p.ldq; c += 64) a.dq_rows[(size_t)row * p.ldq + c] = c == act ? dq : 0.f;
    }
    __syncthreads();
    // ---- last layer's input gradient: relu'(h) * dq * W_last[:, act]   (the dX launch's only non-zero term), columns [c0, c0 + 64)
    {
        float hv[4], wv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = tid + 512 * u, r = e >> 6, k = c0 + (e & 63);
            const int rc = min(m0 + r, a.B - 1);
            hv[u] = p.hin[0][(size_t)rc * p.ldh + k];
            wv[u] = m0 + r < a.B ? p.w_last[(size_t)k * p.w_ld + s_act[r]] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = tid + 512 * u, r = e >> 6, k = c0 + (e & 63);
            if (m0 + r < a.B) p.dh[(size_t)(m0 + r) * p.ldh + k] = hv[u] > 0.f ? s_dq[r] * wv[u] : 0.f;
        }
    }
    if (!last_workgroup(p.ticket, gridDim.x * gridDim.y, &s_last)) return;
    // ---- k_mean_rows, by the last workgroup to finish (threads 0..255 in its order)
    float s = 0.f;
    if (tid < 256) for (int b = tid; b < a.B; b += 256) s += ld_agent(a.loss_row + b);
    if (tid < 256) mred[tid] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (tid < w) mred[tid] += mred[tid + w];
        __syncthreads();
    }
    if (tid == 0) p.loss[0] = mred[0] * p.scale;
}

// [B][in_dim] f32 rows of up to MAXZ network instances into their zero-padded [B][Kp0] input matrices, one launch
struct MlpPackArgs { const float* rows[MAXZ]; float* x[MAXZ]; int nz, B, in_dim, Kp0; };
__global__ void k_mlp_pack_z(MlpPackArgs p)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x, per = p.B * p.in_dim;
    if (t >= p.nz * per) return;
    const int z = t / per, e = t % per, b = e / p.in_dim, c = e % p.in_dim;
    p.x[z][(size_t)b * p.Kp0 + c] = p.rows[z][e];
}

}  // namespace

// ================================================================================================
struct DqnMlp : bdr_agent {
    bdr_dqn_config cfg;
    MlpLayout net;
    float *q = nullptr, *q_tgt = nullptr, *grad = nullptr, *m = nullptr, *v = nullptr;
    float* vmax = nullptr; bool amsgrad = false;   // AdamW{amsgrad: true} (see DqnCnn)
    bool head_fuse = true, head_fuse_wide = false, head_fused = false;     // k_mlp_head_td (BDR_NO_MLP_HEAD_FUSE=1: last layer, TD, loss mean, last dX as four launches)
    unsigned* ticket = nullptr;                    // its last-workgroup ticket
    int B = 0;
    float* x_in[MAXZ] = {nullptr};                 // packed inputs [B][Kp0]
    std::vector<float*> acts[MAXZ];                // per layer [B][Np]
    std::vector<float*> dys;                       // gradient w.r.t. each layer's (pre-activation) output
    float *pred = nullptr, *tgt = nullptr, *loss_row = nullptr, *loss = nullptr;
    // update_on_batch staging
    uint8_t *u_obs = nullptr, *u_next = nullptr, *u_act = nullptr; float* u_rew = nullptr; int8_t* u_term = nullptr;
    uint64_t u_cap = 0;
    const float* last_reward = nullptr; int last_B = 0;
    uint64_t adam_step = 0, soft_update_counter = 0;
    bool defer_adam = false;   // update_critic stops after backward (synchronous-DP mode, grads_on_batch)
    bool lds_step = true; size_t lds_attr = 0;   // BDR_NO_MLP_LDS=1: phases exchange their matrices through global memory
    bool gather_in_step = true;   // the fused step kernel also draws and copies the batch (BDR_NO_STEP_GATHER=1: separate gather launch)
    StepGraph graph; StepGraphPolicy graph_policy; uint64_t batch_gen = 0;   // the layer-by-layer step replayed from a hipGraph (step_graph.hpp)
    bool small_gemm = true;    // layer-by-layer path on the latency-shaped kernels of dense.hpp (BDR_NO_SMALL_GEMM=1: 64x64 tiles, one launch per tensor)
    float* dw_part = nullptr; std::vector<size_t> dw_off; std::vector<int> dw_chunks_l;   // row-chunk partials of the grouped dW launch
    bool fused = true;         // one-workgroup step for nets that fit a CU (mlp_fused.hpp; BDR_NO_MLP_FUSED=1: generic path)
    bool track_with_next = false, track_done = false;   // opt(): the soft update rides on the fused kernel of the last update

    ~DqnMlp() override
    {
        (void)hipSetDevice(device);
        (void)hipStreamSynchronize(stream);
        free_batch();
        (void)hipFree(q); (void)hipFree(q_tgt); (void)hipFree(grad); (void)hipFree(m); (void)hipFree(v); (void)hipFree(vmax); (void)hipFree(loss); (void)hipFree(ticket);
        (void)hipFree(u_obs); (void)hipFree(u_next); (void)hipFree(u_act); (void)hipFree(u_rew); (void)hipFree(u_term);
    }
    void free_batch()
    {
        for (int z = 0; z < MAXZ; ++z) {
            (void)hipFree(x_in[z]); x_in[z] = nullptr;
            for (auto p : acts[z]) (void)hipFree(p);
            acts[z].clear();
        }
        for (auto p : dys) (void)hipFree(p);
        dys.clear();
        (void)hipFree(pred); (void)hipFree(tgt); (void)hipFree(loss_row); (void)hipFree(dw_part);
        pred = tgt = loss_row = dw_part = nullptr;
    }
    int32_t ensure_batch(int Bn)
    {
        if (Bn <= B) return BDR_OK;
        BDR_HIP(hipStreamSynchronize(stream));
        free_batch();
        for (int z = 0; z < MAXZ; ++z) {
            BDR_TRY(alloc_f(&x_in[z], (size_t)Bn * net.L[0].Kp));
            BDR_HIP(hipMemsetAsync(x_in[z], 0, (size_t)Bn * net.L[0].Kp * 4, stream));
            for (const auto& l : net.L) {
                float* p = nullptr;
                BDR_TRY(alloc_f(&p, (size_t)Bn * l.Np));
                acts[z].push_back(p);
            }
        }
        for (const auto& l : net.L) {
            float* p = nullptr;
            BDR_TRY(alloc_f(&p, (size_t)Bn * l.Np));
            dys.push_back(p);
        }
        BDR_TRY(alloc_f(&pred, Bn)); BDR_TRY(alloc_f(&tgt, Bn)); BDR_TRY(alloc_f(&loss_row, Bn));
        {   // grouped dW (dense.hpp k_dense_dw_small_group): 256 batch rows per workgroup
            dw_off.clear(); dw_chunks_l.clear();
            size_t o = 0;
            for (const auto& l : net.L) {
                const int c = std::max(1, std::min(16, Bn / 256));
                dw_off.push_back(o); dw_chunks_l.push_back(c);
                o += (size_t)c * ((size_t)l.Kp * l.Np + l.Np);
            }
            BDR_TRY(alloc_f(&dw_part, o));
        }
        B = Bn; batch_gen += 1;
        return BDR_OK;
    }

    int32_t forward(int z, const float* params, const uint8_t* obs_rows, int Bn)
    {
        bdr_agent* a = this;
        // pack [B][in_dim] f32 rows into the padded input matrix
        BDR_TRY(pack_rows(stream, reinterpret_cast<const float*>(obs_rows), net.in_dim, net.in_dim, x_in[z], net.L[0].Kp, 0, Bn));
        DenseSrc x{x_in[z], net.L[0].Kp};
        for (size_t i = 0; i < net.L.size(); ++i) {
            Bracket br(a, "mlp_fwd");
            BDR_TRY(dense_forward(a, stream, net.L[i], params, x, acts[z][i], Bn));
            x = DenseSrc{acts[z][i], net.L[i].Np};
        }
        return BDR_OK;
    }

    bool fused_ok(int Bn) const
    {
        if (!fused || net.L.size() > MF_MAXL || Bn > 128 || net.out_dim > 64) return false;
        for (const auto& l : net.L) if (l.Kp > 256 || l.Np > 256) return false;
        // One workgroup wins while every phase is a single pass of its eight waves over the phase's 32x32 blocks; beyond that the
        // layer-by-layer path, which spreads a layer over the chip, is faster (tools/probes/mlp_sizes.py: Mlp[256,256] at B = 64 -
        // the reference's own CartPole example - 4.1 k opt-steps/s in one workgroup, 11.7 k layer by layer).
        const int nz = cfg.double_dqn ? 3 : 2, RB = (Bn + 31) / 32;
        for (const auto& l : net.L) if (nz * RB * (l.Np / 32) > 8 || RB * (l.Kp / 32) * (l.Np / 32) > 8) return false;
        return true;
    }
    // the whole update in one launch (mlp_fused.hpp)
    int32_t update_critic_fused(int Bn, const uint8_t* obs, const uint8_t* next_obs, const uint8_t* act, int act_bytes,
                                const float* reward, const int8_t* term, const float* weight, const GatherArgs* gather = nullptr)
    {
        MlpFusedArgs f{};
        if (gather) { f.do_gather = 1; f.g = *gather; }
        const int L = (int)net.L.size();
        f.L = L; f.nz = cfg.double_dqn ? 3 : 2; f.B = Bn; f.A = net.out_dim; f.in_dim = net.in_dim;
        for (int i = 0; i < L; ++i) { f.Kp[i] = net.L[i].Kp; f.Np[i] = net.L[i].Np; f.relu[i] = net.L[i].relu; f.w[i] = net.L[i].w; f.b[i] = net.L[i].b; f.dy[i] = dys[i]; f.in_rows_l[i] = net.L[i].in; }
        const float* par[3] = {q, q_tgt, q};
        const uint8_t* rows[3] = {obs, next_obs, next_obs};
        for (int z = 0; z < f.nz; ++z) {
            f.params[z] = par[z]; f.in_rows[z] = reinterpret_cast<const float*>(rows[z]); f.x_in[z] = x_in[z];
            for (int i = 0; i < L; ++i) f.act[z][i] = acts[z][i];
        }
        f.actions = act; f.act_bytes = act_bytes; f.reward = reward; f.term = term;
        f.pred = pred; f.tgt = tgt; f.loss_row = loss_row; f.loss = loss;
        f.gamma = (float)cfg.discount_factor; f.loss_kind = cfg.critic_loss; f.double_dqn = cfg.double_dqn;
        f.weight = weight; f.td_abs = td_abs; f.has_clip = cfg.has_clip_td_err; f.clip_min = (float)cfg.clip_td_err_min; f.clip_max = (float)cfg.clip_td_err_max;
        f.err = dev_err;
        f.q = q; f.grad = grad; f.m = m; f.v = v; f.q_tgt = q_tgt; f.total = net.total;
        f.do_adam = defer_adam ? 0 : 1;
        if (f.do_adam) {
            adam_step += 1;
            f.adam = adam_scalars_for(cfg.opt_kind == BDR_OPT_ADAMW, cfg.lr, cfg.beta1, cfg.beta2, cfg.eps, cfg.weight_decay, adam_step);
        }
        f.do_track = (track_with_next && f.do_adam) ? 1 : 0;
        f.tau = (float)cfg.tau; f.omt = (float)(1.0 - cfg.tau);
        const size_t lds_bytes = lds_step ? mf_lds_plan(f) : 0;
        Bracket br(this, "mlp_step");
        if (lds_bytes) {   // activations and gradients resident in LDS (mlp_fused.hpp)
            if (lds_bytes > lds_attr) {
                BDR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_dqn_mlp_step_lds), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
                lds_attr = lds_bytes;
            }
            hipLaunchKernelGGL(k_dqn_mlp_step_lds, dim3(1), dim3(512), lds_bytes, stream, f);
        } else {
            hipLaunchKernelGGL(k_dqn_mlp_step, dim3(1), dim3(512), 0, stream, f);
        }
        BDR_HIP(hipGetLastError());
        if (f.do_track) track_done = true;
        track_with_next = false;
        return BDR_OK;
    }

    // Dqn::update_critic (dqn/base.rs:60-160) on a device-resident batch
    int32_t update_critic(int Bn, const uint8_t* obs, const uint8_t* next_obs, const uint8_t* act, int act_bytes,
                          const float* reward, const int8_t* term, const float* weight = nullptr, bdr_replay* per_buffer = nullptr,
                          const GatherArgs* gather = nullptr)
    {
        bdr_agent* a = this;
        BDR_TRY(ensure_batch(Bn));
        BDR_TRY(td_buffer(Bn));
        last_reward = reward; last_B = Bn;
        if (fused_ok(Bn)) {
            BDR_TRY(update_critic_fused(Bn, obs, next_obs, act, act_bytes, reward, term, weight, gather));
            if (per_buffer && weight) BDR_TRY(replay_update_priority_on_stream(per_buffer, Bn, td_abs, stream));
            return BDR_OK;
        }
        const int L = (int)net.L.size();
        const bool lat = small_gemm && L <= RA_SEGS;   // latency-shaped kernels: ~11 launches instead of ~21
        head_fused = false;
        if (lat) {
            const int nz = cfg.double_dqn ? 3 : 2;
            const float* par[MAXZ] = {q, q_tgt, q};
            MlpPackArgs pk{};
            pk.rows[0] = reinterpret_cast<const float*>(obs); pk.rows[1] = pk.rows[2] = reinterpret_cast<const float*>(next_obs);
            for (int z = 0; z < nz; ++z) pk.x[z] = x_in[z];
            pk.nz = nz; pk.B = Bn; pk.in_dim = net.in_dim; pk.Kp0 = net.L[0].Kp;
            { Bracket br(a, "mlp_pack"); BDR_HIP(step_launch(stream, false, k_mlp_pack_z, dim3((nz * Bn * net.in_dim + 255) / 256), dim3(256), pk)); }
            DenseSrc in[MAXZ]; float* out[MAXZ];
            for (int z = 0; z < nz; ++z) in[z] = DenseSrc{x_in[z], net.L[0].Kp};
            // the last layer, the TD rows, the loss mean and the last layer's input gradient in one row-block launch (k_mlp_head_td)
            // measured (tools/probes/mlp_sizes.py, opt-steps/s fused / four launches): [64,64] B=128 27.6k / 24.5k, [128,128] B=64 25.9k / 25.3k,
            // [256,256] B=64 23.1k / 24.8k - a 256-wide last hidden layer makes the row block's tile + gradient pass longer than the launches it saves
            head_fused = head_fuse && L >= 2 && net.out_dim <= 32 && net.L[L - 1].Kp <= (head_fuse_wide ? 4096 : 128) && !(per_buffer && weight);
            for (int i = 0; i < (head_fused ? L - 1 : L); ++i) {
                for (int z = 0; z < nz; ++z) out[z] = acts[z][i];
                Bracket br(a, "mlp_fwd");
                BDR_TRY(dense_forward_z(stream, net.L[i], nz, par, in, out, Bn, true));
                for (int z = 0; z < nz; ++z) in[z] = DenseSrc{out[z], net.L[i].Np};
            }
        } else {
            track_with_next = false;
            BDR_TRY(forward(0, q, obs, Bn));
            BDR_TRY(forward(1, q_tgt, next_obs, Bn));
            if (cfg.double_dqn) BDR_TRY(forward(2, q, next_obs, Bn));
        }
        TdDenseArgs t{};
        t.q_on = acts[0][L - 1]; t.q_tg = acts[1][L - 1]; t.q_on_next = cfg.double_dqn ? acts[2][L - 1] : nullptr;
        t.ld = net.L[L - 1].Np; t.act = act; t.act_bytes = act_bytes; t.reward = reward; t.term = term;
        t.dq_rows = dys[L - 1]; t.pred = pred; t.tgt = tgt; t.loss_row = loss_row;
        t.B = Bn; t.A = net.out_dim; t.gamma = (float)cfg.discount_factor; t.loss_kind = cfg.critic_loss;
        t.weight = weight; t.td_abs = td_abs;
        t.has_clip = cfg.has_clip_td_err; t.clip_min = (float)cfg.clip_td_err_min; t.clip_max = (float)cfg.clip_td_err_max;
        t.err = dev_err; t.relu_out = net.L[L - 1].relu;
        if (lat && head_fused) {
            const DenseLayer& ll = net.L[L - 1];
            const int nz = cfg.double_dqn ? 3 : 2;
            const float* par[MAXZ] = {q, q_tgt, q};
            MlpHeadTdArgs h{};
            h.nz = nz;
            for (int z = 0; z < nz; ++z) { h.hin[z] = acts[z][L - 2]; h.wl[z] = HeadRef{par[z] + ll.w, par[z] + ll.b, ll.relu}; h.q[z] = acts[z][L - 1]; }
            h.ldh = ll.Kp; h.kred = ll.Kp; h.w_ld = ll.Np; h.ldq = ll.Np; h.sel_z = cfg.double_dqn ? 2 : -1;
            h.t = t; h.w_last = q + ll.w; h.dh = dys[L - 2]; h.ticket = ticket; h.loss = loss; h.scale = 1.0f / (float)Bn;
            Bracket br(a, "mlp_head_td");
            BDR_HIP(step_launch(stream, false, k_mlp_head_td, dim3((Bn + 31) / 32, ll.Kp / 64), dim3(512), h));
        } else {
            { Bracket br(a, "td_dense"); BDR_HIP(step_launch(stream, false, k_td_dense, dim3((Bn + 3) / 4), dim3(256), t)); }
            if (per_buffer && weight) { Bracket br(a, "per_update"); BDR_TRY(replay_update_priority_on_stream(per_buffer, Bn, td_abs, stream)); }
            {
                Bracket br(a, "loss_mean");
                BDR_HIP(step_launch(stream, false, k_mean_rows, dim3(1), dim3(256), loss_row, Bn, loss, 1.0f / (float)Bn));
            }
        }
        if (lat) {
            // input gradients down the net, then every weight gradient in one grouped launch; its row-chunk partials are summed
            // into the gradient arena by the kernel that also applies Adam (and the soft update that follows this update)
            for (int i = head_fused ? L - 2 : L - 1; i > 0; --i) { Bracket br(a, "mlp_dx"); BDR_TRY(dense_dx(stream, net.L[i], q, dys[i], dys[i - 1], acts[0][i - 1], Bn, false, true)); }
            DenseDwJob jobs[RA_SEGS];
            for (int i = 0; i < L; ++i)
                jobs[i] = DenseDwJob{&net.L[i], i == 0 ? DenseSrc{x_in[0], net.L[0].Kp} : DenseSrc{acts[0][i - 1], net.L[i - 1].Np}, dys[i], dw_part + dw_off[i],
                                     std::min(dw_chunks_l[i], std::max(1, Bn / 256))};
            { Bracket br(a, "mlp_dw"); BDR_TRY(dense_dw_small_group(stream, jobs, L, Bn)); }
            ReduceAdamArgs ra{};
            ra.nseg = L;
            for (int i = 0; i < L; ++i) {
                const size_t nfl = (size_t)net.L[i].Kp * net.L[i].Np + net.L[i].Np;
                ra.seg[i] = DenseReduceSeg{dw_part + dw_off[i], nfl, jobs[i].chunks, (unsigned)(net.L[i].w / 4), (unsigned)(nfl / 4)};
            }
            ra.p[0] = q; ra.g[0] = grad; ra.m[0] = m; ra.v[0] = v; ra.tgt[0] = q_tgt; ra.n4 = (unsigned)(net.total / 4);
            ra.grads_only = defer_adam ? 1 : 0;
            if (!defer_adam) {
                adam_step += 1;
                ra.s[0] = adam_scalars_for(cfg.opt_kind == BDR_OPT_ADAMW, cfg.lr, cfg.beta1, cfg.beta2, cfg.eps, cfg.weight_decay, adam_step);
                ra.track = track_with_next ? 1 : 0; ra.tau = (float)cfg.tau; ra.omt = (float)(1.0 - cfg.tau);
                if (ra.track) track_done = true;
            }
            track_with_next = false;
            Bracket br(a, "adam");
            BDR_HIP(step_launch(stream, true, k_dense_reduce_adam, dim3((ra.n4 + 255) / 256, 1), dim3(256), ra));
            return BDR_OK;
        }
        for (int i = L - 1; i >= 0; --i) {
            DenseSrc x = i == 0 ? DenseSrc{x_in[0], net.L[0].Kp} : DenseSrc{acts[0][i - 1], net.L[i - 1].Np};
            { Bracket br(a, "mlp_dw"); BDR_TRY(dense_dw(stream, net.L[i], grad, x, dys[i], Bn)); }
            if (i > 0) { Bracket br(a, "mlp_dx"); BDR_TRY(dense_dx(stream, net.L[i], q, dys[i], dys[i - 1], acts[0][i - 1], Bn)); }
        }
        if (!defer_adam) BDR_TRY(adam_all());
        return BDR_OK;
    }
    int32_t adam_all()
    {
        adam_step += 1;
        const AdamScalars s = adam_scalars_for(cfg.opt_kind == BDR_OPT_ADAMW, cfg.lr, cfg.beta1, cfg.beta2, cfg.eps, cfg.weight_decay, adam_step);
        Bracket br(this, "adam");
        if (amsgrad) return launch_adam_amsgrad(stream, q, grad, m, v, vmax, net.total, s);
        return launch_adam(stream, q, grad, m, v, net.total, s);
    }
    int32_t apply_grads() override
    {
        BDR_TRY(adam_all());
        return after_updates();
    }
    int32_t grads_on_batch(uint64_t n, const void* obs, const int64_t* act, const void* next_obs, const float* reward, const int8_t* term) override;

    int32_t after_updates()   // dqn/base.rs:190-198
    {
        soft_update_counter += 1;
        if (soft_update_counter == cfg.soft_update_interval) {
            soft_update_counter = 0;
            if (!track_done) {   // (the fused step kernel of the last update may already have done it)
                Bracket br(this, "track");
                BDR_TRY(launch_track(stream, q_tgt, q, net.total, cfg.tau));
            }
        }
        track_done = false;
        n_opts += 1;
        return BDR_OK;
    }

    const char* kind() const override { return "dqn_mlp"; }
    int32_t opt(bdr_replay* r) override
    {
        BDR_REQUIRE(r->obs_bytes == (uint64_t)net.in_dim * 4, "replay obs rows (%llu B) do not match the Mlp input (%d f32)",
                    (unsigned long long)r->obs_bytes, net.in_dim);
        BDR_REQUIRE(r->act_bytes >= 8, "discrete actions are stored as i64");
        BDR_REQUIRE(r->device == device, "agent and replay buffer live on different devices");
        const int bs = (int)cfg.batch_size;
        // The layer-by-layer step (nets beyond the one-workgroup kernel) is ~11 launches of a few us: replayed from a captured graph
        // when the host is what the device waits for (step_graph.hpp).  Not with prioritized replay (tree kernels with host state),
        // the synchronous-DP exchange, profiling, or a soft update that does not ride on the step's last kernel.
        const bool graphable = !fused_ok(bs) && small_gemm && net.L.size() <= (size_t)RA_SEGS && !r->per && !grad_comm && !amsgrad && !prof;
        if (graphable) {
            const int w = graph_policy.want(stream);
            if (w < 0) return fail(BDR_ERR_HIP, "hipStreamQuery failed");
            if (w == 1) {
                BDR_TRY(ensure_batch(bs));
                BDR_TRY(td_buffer(bs));
                BDR_TRY(replay_prepare_sample(r, bs, stream));
                const ReplaySnap rs(r);   // host state the pass advances, put back if the step has to be enqueued a second time
                const uint64_t s_adam = adam_step, s_soft = soft_update_counter, s_opts = n_opts;
                const bool s_twn = track_with_next, s_td = track_done;
                return step_graph_run(&graph, stream, r->uid, r->batch_gen, batch_gen ^ ((uint64_t)(uintptr_t)td_abs << 8), [&]() { return opt_enqueue(r); },
                                      [&]() { rs.restore(r); adam_step = s_adam; soft_update_counter = s_soft; n_opts = s_opts; track_with_next = s_twn; track_done = s_td; });
            }
        }
        return opt_enqueue(r);
    }
    int32_t opt_enqueue(bdr_replay* r)
    {
        for (uint64_t u = 0; u < cfg.n_updates_per_opt; ++u) {
            // a uniform sample over the plain ring is drawn by the step kernel itself when the step is one kernel anyway
            GatherArgs plan{};
            const bool in_kernel = gather_in_step && fused_ok((int)cfg.batch_size) && !r->per && !r->frame_stack && !r->index_rng && cfg.batch_size <= 128 &&
                                   r->obs_bytes % 4 == 0;
            if (in_kernel) BDR_TRY(replay_sample_plan(r, cfg.batch_size, stream, &plan));
            else { Bracket br(this, "sample"); BDR_TRY(replay_sample_on_stream(r, cfg.batch_size, stream)); }
            defer_adam = grad_comm != nullptr || amsgrad;
            // the soft update that follows the last update of this opt (dqn/base.rs:190-196) rides on its fused kernel
            track_with_next = u + 1 == cfg.n_updates_per_opt && soft_update_counter + 1 == cfg.soft_update_interval && !defer_adam;
            const int32_t st = update_critic((int)cfg.batch_size, r->b_obs, r->b_next, r->b_act, (int)r->act_bytes, r->b_reward, r->b_term,
                                             replay_batch_weights(r), r, in_kernel ? &plan : nullptr);
            defer_adam = false;
            BDR_TRY(st);
            if (grad_comm) {   // synchronous data-parallel step (see bdr_agent::grad_comm)
                Bracket br(this, "grad_allreduce"); BDR_TRY(grad_reduce(this, grad_comm));
            }
            if (grad_comm || amsgrad) BDR_TRY(adam_all());
        }
        return after_updates();
    }
    // Agent::opt_with_record (dqn/base.rs:316-342), see DqnCnn::record
    int32_t record(float* out, int cap, int* n) override
    {
        float l = 0;
        BDR_HIP(hipMemcpyAsync(&l, loss, 4, hipMemcpyDeviceToHost, stream));
        BDR_HIP(hipStreamSynchronize(stream));
        std::vector<float> v = {l};
        if (cfg.record_verbose_level >= 2) {
            std::vector<float> p(last_B), t(last_B), rw(last_B);
            BDR_HIP(hipMemcpy(p.data(), pred, last_B * 4, hipMemcpyDeviceToHost));
            BDR_HIP(hipMemcpy(t.data(), tgt, last_B * 4, hipMemcpyDeviceToHost));
            BDR_HIP(hipMemcpy(rw.data(), last_reward, last_B * 4, hipMemcpyDeviceToHost));
            double sp = 0, st = 0, sr = 0;
            for (int i = 0; i < last_B; ++i) { sp += p[i]; st += t[i]; sr += rw[i]; }
            v.insert(v.end(), {(float)(sp / last_B), (float)(sr / last_B), (float)(st / last_B), (float)((st - sp) / last_B)});
            if (rec_opt) {
                std::vector<float> ref(net.ref_total);
                BDR_TRY(get_params(0, ref.data(), ref.size()));
                param_stats(meta(), ref.data(), v);
                v.push_back(n_samples_act == 0 ? 0.f : (float)n_samples_best_act / (float)n_samples_act);
                n_samples_act = 0; n_samples_best_act = 0;
            }
        }
        for (size_t i = 0; i < v.size() && (int)i < cap; ++i) out[i] = v[i];
        *n = (int)std::min(v.size(), (size_t)cap);
        return BDR_OK;
    }
    void record_keys(std::vector<std::string>& keys) override
    {
        keys = {"loss"};
        if (cfg.record_verbose_level >= 2) {
            keys.insert(keys.end(), {"pred_mean", "reward_mean", "tgt_mean", "tgt_minus_pred_mean"});
            param_stat_keys(meta(), keys);
            keys.push_back("ratio_best_act");
        }
    }
    float* arena_ptr(int which)
    {
        switch (which) { case 0: return q; case 1: return q_tgt; case 2: return m; case 3: return v; case 4: return grad; case 5: return vmax; default: return nullptr; }
    }
    uint64_t param_count(int which) override { return which == -1 ? (uint64_t)net.out_dim : net.ref_total; }
    int32_t get_params(int which, float* out, uint64_t n) override
    {
        float* src = arena_ptr(which);
        BDR_REQUIRE(src, "which must be 0..4");
        BDR_REQUIRE(n == net.ref_total, "parameter count mismatch (%llu vs %llu)", (unsigned long long)n, (unsigned long long)net.ref_total);
        std::vector<float> in(net.total);
        BDR_HIP(hipMemcpyAsync(in.data(), src, net.total * 4, hipMemcpyDeviceToHost, stream));
        BDR_HIP(hipStreamSynchronize(stream));
        mlp_to_reference(net, 0, in.data(), out);
        return BDR_OK;
    }
    int32_t set_params(int which, const float* inp, uint64_t n) override
    {
        float* dst = arena_ptr(which);
        BDR_REQUIRE(dst, "which must be 0..4");
        BDR_REQUIRE(n == net.ref_total, "parameter count mismatch");
        std::vector<float> in(net.total, 0.f);
        mlp_to_internal(net, 0, inp, in.data());
        BDR_HIP(hipMemcpyAsync(dst, in.data(), net.total * 4, hipMemcpyHostToDevice, stream));
        BDR_HIP(hipStreamSynchronize(stream));
        return BDR_OK;
    }
    float* arena(int which, size_t* n) override { if (n) *n = net.total; return arena_ptr(which); }
    std::vector<NamedTensor> meta() const
    {
        std::vector<NamedTensor> mt;
        for (size_t i = 0; i < net.L.size(); ++i) {
            mt.push_back({"mlp.ln" + std::to_string(i) + ".weight", {(uint64_t)net.L[i].out, (uint64_t)net.L[i].in}});
            mt.push_back({"mlp.ln" + std::to_string(i) + ".bias", {(uint64_t)net.L[i].out}});
        }
        return mt;
    }
    int32_t save(const char* dir) override
    {
        std::vector<float> ref(net.ref_total);
        BDR_TRY(get_params(0, ref.data(), ref.size()));
        BDR_TRY(save_named(ckpt_save_path(this, dir, "qnet"), meta(), ref.data(), ref.size()));
        BDR_TRY(get_params(1, ref.data(), ref.size()));
        return save_named(ckpt_save_path(this, dir, "qnet_tgt"), meta(), ref.data(), ref.size());
    }
    int32_t load(const char* dir) override
    {
        std::vector<float> ref(net.ref_total);
        BDR_TRY(load_named(ckpt_load_path(this, dir, "qnet"), meta(), ref.data(), ref.size()));
        BDR_TRY(set_params(0, ref.data(), ref.size()));
        BDR_TRY(load_named(ckpt_load_path(this, dir, "qnet_tgt"), meta(), ref.data(), ref.size()));
        return set_params(1, ref.data(), ref.size());
    }
};

namespace bdr {

int32_t dqn_mlp_create(const bdr_dqn_config* cfg, bdr_agent** out)
{
    BDR_REQUIRE(cfg->net.in_dim >= 1 && cfg->net.n_units >= 0 && cfg->net.n_units <= BDR_MAX_UNITS, "bad Mlp config");
    BDR_REQUIRE(cfg->net.out_dim >= 1 && cfg->net.out_dim <= 64, "out_dim must be in [1,64]");
    DqnMlp* a = new DqnMlp();
    a->cfg = *cfg; a->device = cfg->device; a->train = cfg->train != 0;
    a->net = make_mlp(cfg->net.in_dim, cfg->net.units, cfg->net.n_units, cfg->net.out_dim, cfg->net.activation_out != 0);
    BDR_HIP(hipStreamCreateWithFlags(&a->stream, hipStreamNonBlocking));
    BDR_TRY(a->err_init());
    a->fused = getenv("BDR_NO_MLP_FUSED") == nullptr;
    a->gather_in_step = getenv("BDR_NO_STEP_GATHER") == nullptr;
    { const char* e = getenv("BDR_NO_SMALL_GEMM"); a->small_gemm = !(e && e[0] == '1'); }
    a->graph_policy.from_env();
    a->lds_step = getenv("BDR_NO_MLP_LDS") == nullptr;
    a->head_fuse = getenv("BDR_NO_MLP_HEAD_FUSE") == nullptr;
    a->head_fuse_wide = getenv("BDR_MLP_HEAD_FUSE_WIDE") != nullptr;   // (tests: the row-block head for any width)
    BDR_HIP(hipMalloc((void**)&a->ticket, sizeof(unsigned))); BDR_HIP(hipMemsetAsync(a->ticket, 0, sizeof(unsigned), a->stream));
    float** arenas[5] = {&a->q, &a->q_tgt, &a->grad, &a->m, &a->v};
    for (auto p : arenas) {
        BDR_TRY(alloc_f(p, a->net.total));
        BDR_HIP(hipMemsetAsync(*p, 0, a->net.total * 4, a->stream));
    }
    a->amsgrad = cfg->opt_kind == BDR_OPT_ADAMW && cfg->amsgrad != 0;
    if (a->amsgrad) { BDR_TRY(alloc_f(&a->vmax, a->net.total)); BDR_HIP(hipMemsetAsync(a->vmax, 0, a->net.total * 4, a->stream)); }
    BDR_TRY(alloc_f(&a->loss, 4));
    std::vector<float> ref(a->net.ref_total);
    mlp_init_reference(a->net, cfg->param_seed, ref.data());
    BDR_TRY(a->set_params(0, ref.data(), ref.size()));
    BDR_TRY(a->set_params(1, ref.data(), ref.size()));   // DqnModel::clone
    BDR_TRY(a->ensure_batch((int)cfg->batch_size));
    *out = a;
    return BDR_OK;
}

int32_t dqn_mlp_update_on_batch(bdr_agent* base, uint64_t n, const void* obs, const int64_t* act, const void* next_obs,
                                const float* reward, const int8_t* term, const float* weight)
{
    DqnMlp* a = static_cast<DqnMlp*>(base);
    const size_t ob = (size_t)a->net.in_dim * 4;
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
    const int32_t st = a->update_critic((int)n, a->u_obs, a->u_next, a->u_act, 8, a->u_rew, a->u_term, wd, nullptr);
    a->defer_adam = backward_only;
    BDR_TRY(st);
    if (!backward_only) {
        if (a->amsgrad) BDR_TRY(a->adam_all());
        BDR_TRY(a->after_updates());
    }
    BDR_HIP(hipStreamSynchronize(a->stream));
    return BDR_OK;
}

}  // namespace bdr

int32_t DqnMlp::grads_on_batch(uint64_t n, const void* obs, const int64_t* act, const void* next_obs, const float* reward, const int8_t* term)
{
    defer_adam = true;
    const int32_t st = bdr::dqn_mlp_update_on_batch(this, n, obs, act, next_obs, reward, term, nullptr);
    defer_adam = false;
    return st;
}

namespace bdr {
int32_t dqn_mlp_qvalues(bdr_agent* base, uint64_t n, const void* obs, float* q_out)
{
    DqnMlp* a = static_cast<DqnMlp*>(base);
    BDR_TRY(a->ensure_batch((int)n));
    const size_t ob = (size_t)a->net.in_dim * 4;
    const uint8_t* d = nullptr;
    if (!a->obs_rows_on_device && n * ob <= bdr_agent::HOST_ROWS_PINNED_MAX) BDR_TRY(a->host_rows_pinned(obs, n * ob, &d));   // (read in place by the packing kernel)
    else {
        uint8_t* stage = nullptr;
        BDR_TRY(a->act_buffer(n * ob, (void**)&stage));
        BDR_TRY(a->stage_obs(stage, obs, ob, n, a->stream));
        d = stage;
    }
    int32_t st = a->forward(0, a->q, d, (int)n);
    const int L = (int)a->net.L.size(), ld = a->net.L[L - 1].Np, A = a->net.out_dim;
    std::vector<float> tmp(n * ld);
    if (st == BDR_OK) st = a->rows_to_host(a->acts[0][L - 1], tmp.data(), tmp.size());
    a->slot_cursor = 0;
    BDR_TRY(st);
    for (uint64_t i = 0; i < n; ++i) for (int k = 0; k < A; ++k) q_out[i * A + k] = tmp[i * ld + k];
    return BDR_OK;
}

int32_t dqn_mlp_probe(bdr_agent* base, int32_t what, float* out, uint64_t n)
{
    DqnMlp* a = static_cast<DqnMlp*>(base);
    const int L = (int)a->net.L.size(), ld = a->net.L[L - 1].Np, A = a->net.out_dim;
    BDR_HIP(hipStreamSynchronize(a->stream));
    if (what == 0 || what == 1) {   // [B][A] from the padded rows
        const uint64_t rows = n / A;
        std::vector<float> tmp(rows * ld);
        BDR_HIP(hipMemcpy(tmp.data(), a->acts[what][L - 1], tmp.size() * 4, hipMemcpyDeviceToHost));
        for (uint64_t i = 0; i < rows; ++i) for (int k = 0; k < A; ++k) out[i * A + k] = tmp[i * ld + k];
        return BDR_OK;
    }
    const float* src = what == 2 ? a->pred : what == 3 ? a->tgt : what == 4 ? a->loss : nullptr;
    BDR_REQUIRE(src, "unknown probe %d", what);
    BDR_HIP(hipMemcpy(out, src, n * 4, hipMemcpyDeviceToHost));
    return BDR_OK;
}

}  // namespace bdr
