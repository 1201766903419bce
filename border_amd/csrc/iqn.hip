// IQN agent on MI355X: Iqn::update_critic / opt_ (border-tch-agent/src/iqn/base.rs:63-191),
// IqnModel::forward (iqn/model/base.rs:198-234) with the cosine embedding (:162-191, i = 1..embed_dim
// inclusive) and quantile_huber_loss (util/quantile_loss.rs:7-13).
//   psi  : AtariCnn{skip_linear:true} (conv1 on bf16 MFMA with exact operands, conv2/conv3 FP32 MFMA;
//          features are the NHWC flatten of conv3, the cos / merge weights are permuted accordingly)
//          or Mlp(in -> units -> feature_dim)
//   phi  : relu(cos(tau*pi*i) . Wc + bc)              dense FP32-MFMA, Hadamard merge fused in the epilogue
//   f    : Mlp(feature_dim -> units -> n_actions)      dense FP32-MFMA on B*N rows
//   loss : pairwise quantile-Huber, one workgroup per batch row, fixed-order reductions
// Reference quirk kept: IqnSample::Const32 yields 33 points (range(0,32) inclusive, :361-364).
#include <algorithm>
#include <cstdlib>

#include "dense.hpp"
#include "conv1_bf16_img.hpp"
#include "cnn_layers.hpp"
#include "act_small.hpp"

using namespace bdr;

namespace {

__global__ void k_iqn_cos(const float* __restrict__ tau, float* __restrict__ cosv, int M, int E, int Ep)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= M * Ep) return;
    const int r = t / Ep, i = t % Ep;
    cosv[t] = i < E ? cosf(tau[r] * (3.14159265358979323846f * (float)(i + 1))) : 0.f;
}

// tgt[b][n'] = r + (1-term)*gamma * z_tgt[b,n',a*],  a* = argmax_a mean_n' z_tgt   (iqn/base.rs:113-143)
struct IqnTargetArgs { const float* zt; int ldz; const float* reward; const int8_t* term; float* tgt; int B, Nt, A; float gamma; };
__global__ __launch_bounds__(64) void k_iqn_target(IqnTargetArgs p)
{
    const int b = blockIdx.x, lane = threadIdx.x;
    float mv = -INFINITY;
    if (lane < p.A) {
        float s = 0.f;
        for (int n = 0; n < p.Nt; ++n) s += p.zt[((size_t)b * p.Nt + n) * p.ldz + lane];
        mv = s / (float)p.Nt;
    }
    int idx = lane;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(mv, off);
        const int oi = __shfl_xor(idx, off);
        if (ov > mv || (ov == mv && oi < idx)) { mv = ov; idx = oi; }
    }
    const float k = (float)(1 - (int)p.term[b]) * p.gamma;
    for (int n = lane; n < p.Nt; n += 64) p.tgt[(size_t)b * p.Nt + n] = p.reward[b] + k * p.zt[((size_t)b * p.Nt + n) * p.ldz + idx];
}

// loss_b = sum_{n',n} |tau_p[b,n] - 1{d<0}| huber_1(d),  d = tgt[b,n'] - pred[b,n];  dz = dL/dz rows
struct IqnLossArgs {
    const float* z; int ldz; const uint8_t* act; int act_bytes; const float* tau_p; const float* tgt;
    float* dz; float* loss_row; int B, Np, Nt; float inv;
    int A; unsigned* err;   // bdr_agent::dev_err (ERR_ACTION)
};
__global__ __launch_bounds__(256) void k_iqn_loss(IqnLossArgs p)
{
    __shared__ float red[256];
    const int b = blockIdx.x, t = threadIdx.x;
    long long act = *reinterpret_cast<const long long*>(p.act + (size_t)b * p.act_bytes);
    if (act < 0 || act >= p.A) {   // the reference's gather raises; here: flag for the host, clamp to stay in bounds
        if (t == 0 && p.err) atomicOr(p.err + bdr_agent::ERR_ACTION, 1u);
        act = act < 0 ? 0 : p.A - 1;
    }
    float ls = 0.f;
    for (int n = t; n < p.Np; n += 256) {
        const size_t row = (size_t)b * p.Np + n;
        const float pred = p.z[row * p.ldz + act], tp = p.tau_p[row];
        float g = 0.f;
        for (int m = 0; m < p.Nt; ++m) {
            const float d = p.tgt[(size_t)b * p.Nt + m] - pred;
            const float za = fabsf(d);
            const float hub = za < 1.f ? 0.5f * za * za : za - 0.5f;
            const float dh = za < 1.f ? d : (d > 0.f ? 1.f : -1.f);
            const float w = fabsf(tp - (d < 0.f ? 1.f : 0.f));
            ls += w * hub;
            g -= w * dh;
        }
        for (int c = 0; c < p.ldz; ++c) p.dz[row * p.ldz + c] = c == act ? g * p.inv : 0.f;
    }
    red[t] = ls;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) { if (t < w) red[t] += red[t + w]; __syncthreads(); }
    if (t == 0) p.loss_row[b] = red[0];
}

// backward of m = psi[b] * phi (phi = relu(lin)), in place on dm:  dlin = 1{phi>0} dm psi;  dpsi = sum_n dm phi
__global__ void k_iqn_merge_bwd(float* __restrict__ dm, const float* __restrict__ phi, const float* __restrict__ psi, int ld_psi,
                                float* __restrict__ dpsi, int ld_dpsi, int B, int N, int F, int Fp, int mask_psi)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * F) return;
    const int b = t / F, j = t % F;
    const float ps = psi[(size_t)b * ld_psi + j];
    float s = 0.f;
    for (int n = 0; n < N; ++n) {
        const size_t q = ((size_t)b * N + n) * Fp + j;
        const float d = dm[q], ph = phi[q];
        s = fmaf(d, ph, s);
        dm[q] = ph > 0.f ? d * ps : 0.f;
    }
    dpsi[(size_t)b * ld_dpsi + j] = (mask_psi && !(ps > 0.f)) ? 0.f : s;
}

__global__ __launch_bounds__(256) void k_iqn_sum(const float* __restrict__ x, int n, float* __restrict__ out, float scale, int accumulate)
{
    __shared__ float red[256];
    float s = 0.f;
    for (int b = threadIdx.x; b < n; b += 256) s += x[b];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) { if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w]; __syncthreads(); }
    if (threadIdx.x == 0) out[0] = (accumulate ? out[0] : 0.f) + red[0] * scale;
}

// mean over the percent points: q[b][a] = mean_n z[b,n,a]   (iqn/model/base.rs:394-418 `average`)
__global__ void k_iqn_average(const float* __restrict__ z, int ldz, float* __restrict__ q, int B, int N, int A)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * A) return;
    const int b = t / A, a = t % A;
    float s = 0.f;
    for (int n = 0; n < N; ++n) s += z[((size_t)b * N + n) * ldz + a];
    q[t] = s / (float)N;
}

__global__ void k_rand_uniform(float* __restrict__ out, size_t n, uint64_t seed, uint64_t counter)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t x = (seed + 0x9E3779B97F4A7C15ull) ^ ((counter + i + 1) * 0xBF58476D1CE4E5B9ull);
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31;
    out[i] = (float)(x >> 40) * (1.0f / 16777216.0f);
}

// IqnSample (iqn/model/base.rs:327-387)
int sample_points(int mode) { static const int n[7] = {10, 33, 10, 8, 32, 64, 1}; return n[mode]; }   // Const32 -> 33 points
bool sample_is_uniform(int mode) { return mode == 2 || mode == 3 || mode == 4 || mode == 5; }
void sample_const(int mode, std::vector<float>& t)
{
    t.clear();
    if (mode == 0) for (int i = 0; i < 10; ++i) t.push_back(0.05f + 0.1f * (float)i);
    else if (mode == 1) for (int i = 0; i <= 32; ++i) t.push_back((1.0f / 32.0f) * (float)i);
    else t.push_back(0.5f);
}

inline size_t conv_floats(int ns) { return (size_t)2048 * ns + 32 + 32768 + 64 + 36864 + 64; }   // c1 [32][ns][8][8] .. c3.bias (cnn/base.rs:27-31)

}  // namespace

// ================================================================================================
struct Iqn : bdr_agent {
    bdr_iqn_config cfg;
    bool cnn = true;
    int F = 0, E = 0, A = 0, in_dim = 0;
    Arena conv;                 // conv offsets (cnn psi)
    MlpLayout psi_mlp;          // mlp psi
    MlpLayout hd;               // L[0] = cos-embed layer (E -> F, relu), L[1..] = f layers
    size_t total = 0, ref_total = 0;
    float *p = nullptr, *p_tgt = nullptr, *grad = nullptr, *am = nullptr, *av = nullptr;
    float* avmax = nullptr;     // AdamW{amsgrad: true}: max_exp_avg_sq (arena 5)
    // batch buffers
    int B = 0, Nmax = 0;
    float *a1 = nullptr, *a2 = nullptr, *a3 = nullptr, *dy3 = nullptr, *dy2 = nullptr, *dy1 = nullptr, *part_conv = nullptr;
    float* x_in = nullptr; std::vector<float*> psi_act, psi_dy;
    float *tau_p = nullptr, *tau_t = nullptr, *cosv = nullptr, *phi = nullptr, *mrg = nullptr;
    std::vector<float*> f_act, f_dy;
    float *tgt = nullptr, *loss_row = nullptr, *loss = nullptr, *qavg = nullptr, *part = nullptr;
    size_t part_floats = 0;
    uint8_t *u_obs = nullptr, *u_next = nullptr, *u_act = nullptr; float* u_rew = nullptr; int8_t* u_term = nullptr; uint64_t u_cap = 0;
    uint64_t adam_step = 0, soft_update_counter = 0, noise_counter = 0;
    int n_updates_done = 0;
    // acting calls (bdr_iqn_qvalues on a handful of observations): the trunk's conv2 / conv3 and the merge layer take the acting kernels
    // (act_small.hpp): at 33 percent points per observation the merge layer is a [33 n][3136] x [3136][512] product that the training
    // kernel walks in 8 workgroups of 98 k-tiles (71 us of a 175 us call)
    bool acting = false;
    float* act_part = nullptr; unsigned* act_tickets = nullptr; size_t act_part_floats = 0;
    int32_t act_scratch(size_t floats)
    {
        if (floats > act_part_floats) {
            BDR_HIP(hipStreamSynchronize(stream));
            (void)hipFree(act_part); act_part = nullptr; act_part_floats = 0;
            BDR_HIP(hipMalloc((void**)&act_part, floats * sizeof(float)));
            act_part_floats = floats;
        }
        if (!act_tickets) {
            BDR_HIP(hipMalloc((void**)&act_tickets, 1024 * sizeof(unsigned)));
            BDR_HIP(hipMemsetAsync(act_tickets, 0, 1024 * sizeof(unsigned), stream));
        }
        return BDR_OK;
    }
    // The merge layer f.L[1] ([B*N][F] x [F][units]) on the bf16 matrix cores with split operands (igemm_b3.hpp) when it is large
    // enough to be matrix-bound (config C4: 32 768 x 3 136 x 512): forward of both networks and the input gradient.  The exact
    // FP32-MFMA kernels stay selectable (BDR_IQN_F32_EXACT=1) and serve every smaller shape.
    bool b3_allowed = true;
    bool merge_epilogue = true;
    uint16_t* dypl_tr = nullptr;   // [3][Np][M] bf16 planes of the merge layer's output gradient (split once per update, dense_dw_b3)
    size_t dypl_elems = 0;
    bool dw_b3 = true;
    uint16_t* cpl_tr = nullptr;    // [3][Np][Kp] bf16 planes of the cosine-embedding layer's weights
    bool phi_b3 = true;
    uint16_t *wpl_nat = nullptr, *wpl_tr = nullptr;   // [3][Kp][Np], [3][Np][Kp] bf16 planes of L[1]'s weights, re-split before every use
    bool use_b3(int M) const
    {
        if (!b3_allowed || hd.L.size() < 2) return false;
        const DenseLayer& l = hd.L[1];
        return l.Np % 128 == 0 && l.Kp % 64 == 0 && (size_t)l.Kp * l.Np >= ((size_t)1 << 20) && M >= 4096;
    }
    int32_t split_l1(const float* params, bool need_nat)
    {
        const DenseLayer& l = hd.L[1];
        const size_t n = (size_t)3 * l.Kp * l.Np;
        if (!wpl_tr) { BDR_HIP(hipMalloc((void**)&wpl_tr, n * 2)); BDR_HIP(hipMalloc((void**)&wpl_nat, n * 2)); }
        return dense_split_planes(stream, l, params, need_nat ? wpl_nat : nullptr, wpl_tr);
    }

    ~Iqn() override
    {
        (void)hipSetDevice(device);
        (void)hipStreamSynchronize(stream);
        free_batch();
        (void)hipFree(p); (void)hipFree(p_tgt); (void)hipFree(grad); (void)hipFree(am); (void)hipFree(av); (void)hipFree(avmax); (void)hipFree(loss);
        (void)hipFree(u_obs); (void)hipFree(u_next); (void)hipFree(u_act); (void)hipFree(u_rew); (void)hipFree(u_term);
        (void)hipFree(wpl_nat); (void)hipFree(wpl_tr); (void)hipFree(cpl_tr); (void)hipFree(dypl_tr); (void)hipFree(act_part); (void)hipFree(act_tickets);
    }
    void free_batch()
    {
        float** singles[] = {&a1, &a2, &a3, &dy3, &dy2, &dy1, &part_conv, &x_in, &tau_p, &tau_t, &cosv, &phi, &mrg, &tgt, &loss_row, &qavg, &part};
        for (auto q : singles) { (void)hipFree(*q); *q = nullptr; }
        for (auto v : {&psi_act, &psi_dy, &f_act, &f_dy}) { for (auto q : *v) (void)hipFree(q); v->clear(); }
    }
    int32_t zalloc(float** q, size_t n)
    {
        BDR_TRY(alloc_f(q, n));
        BDR_HIP(hipMemsetAsync(*q, 0, std::max<size_t>(n, 4) * 4, stream));
        return BDR_OK;
    }
    static int dw_chunks(int M) { return M <= 2048 ? 1 : std::min(32, (M / 1024 + 7) / 8 * 8); }
    int32_t ensure_batch(int Bn, int N)
    {
        if (Bn <= B && N <= Nmax) return BDR_OK;
        Bn = std::max(Bn, B); N = std::max(N, Nmax);
        BDR_HIP(hipStreamSynchronize(stream));
        free_batch();
        const size_t M = (size_t)Bn * N;
        const int Fp = hd.L[0].Np, Ep = hd.L[0].Kp;
        if (cnn) {
            BDR_TRY(zalloc(&a1, (size_t)Bn * 400 * 32)); BDR_TRY(zalloc(&a2, (size_t)Bn * 81 * 64)); BDR_TRY(zalloc(&a3, (size_t)Bn * 49 * 64));
            BDR_TRY(zalloc(&dy3, (size_t)Bn * 49 * 64)); BDR_TRY(zalloc(&dy2, (size_t)Bn * 81 * 64)); BDR_TRY(zalloc(&dy1, (size_t)Bn * 400 * 32));
            BDR_TRY(zalloc(&part_conv, dw_plan(Bn, conv.ns).total));
        } else {
            BDR_TRY(zalloc(&x_in, (size_t)Bn * psi_mlp.L[0].Kp));
            for (const auto& l : psi_mlp.L) { float *q = nullptr, *d = nullptr; BDR_TRY(zalloc(&q, (size_t)Bn * l.Np)); BDR_TRY(zalloc(&d, (size_t)Bn * l.Np)); psi_act.push_back(q); psi_dy.push_back(d); }
        }
        BDR_TRY(zalloc(&tau_p, M)); BDR_TRY(zalloc(&tau_t, M));
        BDR_TRY(zalloc(&cosv, M * Ep)); BDR_TRY(zalloc(&phi, M * Fp)); BDR_TRY(zalloc(&mrg, M * Fp));
        size_t pmax = 0;
        const int ch = dw_chunks((int)M);
        for (size_t i = 0; i < hd.L.size(); ++i) {
            const auto& l = hd.L[i];
            if (i >= 1) { float *q = nullptr, *d = nullptr; BDR_TRY(zalloc(&q, M * l.Np)); BDR_TRY(zalloc(&d, M * l.Np)); f_act.push_back(q); f_dy.push_back(d); }
            pmax = std::max(pmax, (size_t)ch * ((size_t)l.Kp * l.Np + l.Np));
        }
        BDR_TRY(zalloc(&part, ch > 1 ? pmax : 4)); part_floats = pmax;
        BDR_TRY(zalloc(&tgt, M)); BDR_TRY(zalloc(&loss_row, Bn)); BDR_TRY(zalloc(&qavg, (size_t)Bn * A));
        B = Bn; Nmax = N;
        return BDR_OK;
    }

    // psi(x) -> feature rows; returns pointer + leading dimension
    int32_t psi_forward(const float* params, const uint8_t* obs, int Bn, const float** feat, int* ld)
    {
        bdr_agent* a = this;
        if (cnn) {
            Conv1Args c{}; c.M = Bn * 400; c.nz = 1;
            c.x[0] = obs; c.w1[0] = params + conv.w1; c.bias[0] = params + conv.b1; c.out[0] = a1;
            { Bracket br(a, "psi_conv1"); BDR_HIP(conv1_forward(conv.ns, Bn, stream, c)); }
            if (acting && Bn <= ACT_SMALL_MAX) {
                auto pad32 = [](int m) { return (m + 31) / 32 * 32; };
                BDR_TRY(act_scratch(act_small_part_floats()));
                ActLayerArgs l{};
                l.part = act_part; l.tickets = act_tickets; l.relu = 1;
                l.x = a1; l.w = params + conv.w2; l.bias = params + conv.b2; l.out = a2; l.M = Bn * 81; l.Mpad = pad32(l.M); l.N = 64; l.K = 512; l.KS = 4;
                BDR_HIP(launch_act_layer<2>(stream, l));
                l.x = a2; l.w = params + conv.w3; l.bias = params + conv.b3; l.out = a3; l.M = Bn * 49; l.Mpad = pad32(l.M); l.N = 64; l.K = 576; l.KS = 3;
                BDR_HIP(launch_act_layer<3>(stream, l));
                *feat = a3; *ld = 3136;
                return BDR_OK;
            }
            FwdArgs f{};
            f.M = Bn * 81; f.x[0] = a1; f.w[0] = params + conv.w2; f.bias[0] = params + conv.b2; f.out[0] = a2;
            { Bracket br(a, "psi_conv2"); LAUNCH(k_igemm<FwdC2>, dim3((f.M + 63) / 64, 1, 1), f); }
            f.M = Bn * 49; f.x[0] = a2; f.w[0] = params + conv.w3; f.bias[0] = params + conv.b3; f.out[0] = a3;
            { Bracket br(a, "psi_conv3"); LAUNCH(k_igemm<FwdC3>, dim3((f.M + 63) / 64, 1, 1), f); }
            *feat = a3; *ld = 3136;
        } else {
            BDR_TRY(pack_rows(stream, reinterpret_cast<const float*>(obs), in_dim, in_dim, x_in, psi_mlp.L[0].Kp, 0, Bn));
            DenseSrc in{x_in, psi_mlp.L[0].Kp};
            for (size_t i = 0; i < psi_mlp.L.size(); ++i) {
                Bracket br(a, "psi_fwd");
                BDR_TRY(dense_forward(a, stream, psi_mlp.L[i], params, in, psi_act[i], Bn));
                in = DenseSrc{psi_act[i], psi_mlp.L[i].Np};
            }
            *feat = psi_act.back(); *ld = psi_mlp.L.back().Np;
        }
        return BDR_OK;
    }

    // IqnModel::forward: z rows [B*N][ldz] in f_act.back()
    int32_t model_forward(const float* params, const uint8_t* obs, const float* tau, int Bn, int N)
    {
        bdr_agent* a = this;
        const float* feat; int ldf;
        BDR_TRY(psi_forward(params, obs, Bn, &feat, &ldf));
        const int M = Bn * N, Ep = hd.L[0].Kp;
        { Bracket br(a, "iqn_cos"); hipLaunchKernelGGL(k_iqn_cos, dim3((M * Ep + 255) / 256), dim3(256), 0, stream, tau, cosv, M, E, Ep); BDR_HIP(hipGetLastError()); }
        // phi = relu(cos-embedding * W + b); the merge m = phi * psi(x)[b] (iqn/model/base.rs) is the input of the next layer and of
        // its weight gradient and is formed inside those two GEMMs on the way into LDS: [B*N][F] floats that are never written
        if (use_b3(M) && phi_b3 && Ep == 64) {   // 13 GFLOP per network at C4 with a 64-long reduction: the resident-rows kernel (dense_k64_b3.hpp)
            Bracket br(a, "iqn_phi_3xbf16");
            const DenseLayer& l = hd.L[0];
            if (!cpl_tr) BDR_HIP(hipMalloc((void**)&cpl_tr, (size_t)3 * l.Kp * l.Np * 2));
            BDR_TRY(dense_split_planes(stream, l, params, nullptr, cpl_tr));    // re-split from THIS network's f32 weights, as for the merge layer
            BDR_TRY(dense_forward_k64_b3(stream, l, params, cpl_tr, DenseSrc{cosv, Ep}, phi, M));
        } else { Bracket br(a, "iqn_phi"); BDR_TRY(dense_forward(a, stream, hd.L[0], params, DenseSrc{cosv, Ep}, phi, M)); }
        DenseSrc in{phi, hd.L[0].Np};
        for (size_t i = 1; i < hd.L.size(); ++i) {
            Bracket br(a, (i == 1 && use_b3(M)) ? "iqn_f_fwd1_3xbf16" : ("iqn_f_fwd" + std::to_string(i)).c_str());   // (the label names the arithmetic that ran)
            if (i == 1 && use_b3(M)) {
                // the planes are re-split from the f32 weights of THIS network right here (~10 us against a 5 ms step): whoever wrote
                // the parameters last - Adam, track, set_params, a model sync, an all-reduce - nothing can leave them stale
                BDR_TRY(split_l1(params, params == p));
                BDR_TRY(dense_forward_had_b3(stream, hd.L[1], params, wpl_tr, in, feat, ldf, N, f_act[0], M));
            } else if (i == 1 && acting && M <= 32 * 16 && hd.L[1].Kp % 448 == 0 && hd.L[1].Np % 32 == 0 && in.ld == hd.L[1].Kp && ldf == hd.L[1].Kp) {
                // 32 x 32 tiles x 7 k-slices instead of 64 x 64 tiles walking the whole reduction
                const DenseLayer& l1 = hd.L[1];
                const int Mpad = (M + 31) / 32 * 32;
                BDR_TRY(act_scratch(std::max(act_small_part_floats(), (size_t)7 * Mpad * l1.Np)));
                ActLayerArgs l{};
                l.part = act_part; l.tickets = act_tickets; l.relu = l1.relu;
                l.x = in.p; l.had = feat; l.had_group = N; l.w = params + l1.w; l.bias = params + l1.b; l.out = f_act[0];
                l.M = M; l.Mpad = Mpad; l.N = l1.Np; l.K = l1.Kp; l.KS = l1.Kp / 448;
                BDR_HIP(launch_act_layer<1>(stream, l));
            } else if (i == 1) BDR_TRY(dense_forward_had(stream, hd.L[1], params, in, feat, ldf, N, f_act[0], M));
            else BDR_TRY(dense_forward(a, stream, hd.L[i], params, in, f_act[i - 1], M));
            in = DenseSrc{f_act[i - 1], hd.L[i].Np};
        }
        return BDR_OK;
    }

    // Iqn::update_critic (iqn/base.rs:63-170) on a device-resident batch with given percent points
    int32_t update_critic(int Bn, const uint8_t* obs, const uint8_t* next_obs, const uint8_t* act, int act_bytes,
                          const float* reward, const int8_t* term, const float* tp, int Np, const float* tt, int Nt, bool first)
    {
        bdr_agent* a = this;
        BDR_TRY(ensure_batch(Bn, std::max(Np, Nt)));
        const int L = (int)hd.L.size(), ldz = hd.L[L - 1].Np, Fp = hd.L[0].Np, Ep = hd.L[0].Kp;
        // target first (no grad): buffers are reused by the prediction pass
        BDR_TRY(model_forward(p_tgt, next_obs, tt, Bn, Nt));
        {
            IqnTargetArgs t{f_act.back(), ldz, reward, term, tgt, Bn, Nt, A, (float)cfg.discount_factor};
            Bracket br(a, "iqn_target");
            hipLaunchKernelGGL(k_iqn_target, dim3(Bn), dim3(64), 0, stream, t);
            BDR_HIP(hipGetLastError());
        }
        BDR_TRY(model_forward(p, obs, tp, Bn, Np));
        const int M = Bn * Np;
        {
            IqnLossArgs l{f_act.back(), ldz, act, act_bytes, tp, tgt, f_dy.back(), loss_row, Bn, Np, Nt, 1.0f / ((float)Bn * (float)Nt * (float)Np), A, dev_err};
            Bracket br(a, "iqn_loss");
            hipLaunchKernelGGL(k_iqn_loss, dim3(Bn), dim3(256), 0, stream, l);
            BDR_HIP(hipGetLastError());
            hipLaunchKernelGGL(k_iqn_sum, dim3(1), dim3(256), 0, stream, loss_row, Bn, loss, 1.0f / ((float)Bn * (float)Nt * (float)Np), first ? 0 : 1);
            BDR_HIP(hipGetLastError());
        }
        const int ch = dw_chunks(M);
        // f backward
        const float* feat = cnn ? a3 : psi_act.back();
        const int ldf = cnn ? 3136 : psi_mlp.L.back().Np;
        for (int i = L - 1; i >= 1; --i) {
            DenseSrc in = i == 1 ? DenseSrc{phi, Fp} : DenseSrc{f_act[i - 2], hd.L[i - 1].Np};
            {
                const bool dw1_b3 = i == 1 && use_b3(M) && dw_b3 && M % 64 == 0 && Np % 8 == 0 && ch > 1;
                Bracket br(a, dw1_b3 ? "iqn_f_dw1_3xbf16" : ("iqn_f_dw" + std::to_string(i)).c_str());
                if (i == 1 && dw1_b3) {   // x = phi * psi[b], on the bf16 matrix cores: dy split once into transposed planes, x in the kernel
                    const size_t n = (size_t)3 * M * hd.L[1].Np;
                    if (dypl_elems < n) { BDR_HIP(hipStreamSynchronize(stream)); (void)hipFree(dypl_tr); dypl_tr = nullptr; BDR_HIP(hipMalloc((void**)&dypl_tr, n * 2)); dypl_elems = n; }
                    BDR_TRY(dense_dw_b3(stream, hd.L[1], grad, in, f_dy[0], dypl_tr, M, part, ch, feat, ldf, Np));
                } else if (i == 1) BDR_TRY(dense_dw(stream, hd.L[i], grad, in, f_dy[i - 1], M, part, ch, feat, ldf, Np));   // x = phi * psi[b]
                else BDR_TRY(dense_dw(stream, hd.L[i], grad, in, f_dy[i - 1], M, part, ch));
            }
            if (i > 1) { Bracket br(a, ("iqn_f_dx" + std::to_string(i)).c_str()); BDR_TRY(dense_dx(stream, hd.L[i], p, f_dy[i - 1], f_dy[i - 2], f_act[i - 2], M)); }
        }
        // dm = dL/dm (no ReLU mask: m is a product, not an activation)
        float* dpsi = cnn ? dy3 : psi_dy.back();
        const int mask_psi = cnn ? 1 : (cfg.psi.activation_out ? 1 : 0);
        bool merged_in_epilogue = false;
        {   // (wpl_nat: split from the online weights by this update's forward; the parameters have not changed since)
            // 64 percent points per sample (Uniform64, BASELINE config 4): a wave of the split-operand kernel owns one sample's rows, and
            // the merge's backward (dlin = 1{phi > 0} dm psi, dpsi = sum_n dm phi) is its epilogue - dm never goes to HBM and back
            const bool fuse_merge = use_b3(M) && Np == 64 && M % 128 == 0 && merge_epilogue;
            Bracket br(a, use_b3(M) ? "iqn_f_dx1_3xbf16" : "iqn_f_dx1");
            if (fuse_merge) BDR_TRY(dense_dx_had_b3(stream, hd.L[1], p, wpl_nat, f_dy[0], mrg, phi, feat, ldf, dpsi, mask_psi, M));
            else if (use_b3(M)) BDR_TRY(dense_dx_b3(stream, hd.L[1], p, wpl_nat, f_dy[0], mrg, nullptr, M));
            else BDR_TRY(dense_dx(stream, hd.L[1], p, f_dy[0], mrg, nullptr, M));
            merged_in_epilogue = fuse_merge;
        }
        if (!merged_in_epilogue) {
            Bracket br(a, "iqn_merge_bwd");
            hipLaunchKernelGGL(k_iqn_merge_bwd, dim3((Bn * F + 255) / 256), dim3(256), 0, stream, mrg, phi, feat, ldf, dpsi, ldf, Bn, Np, F, Fp, mask_psi);
            BDR_HIP(hipGetLastError());
        }
        { Bracket br(a, "iqn_cos_dw"); BDR_TRY(dense_dw(stream, hd.L[0], grad, DenseSrc{cosv, Ep}, mrg, M, part, ch)); }
        // psi backward
        if (cnn) {
            const DwPlan pl = dw_plan(B, conv.ns);
            {
                const int Mr = Bn * 49, chunks = std::min(pl.chunks_c3, (Mr + 31) / 32);
                DwArgs d{a2, dy3, part_conv + pl.off_c3, pl.stride_c3, Mr};
                { Bracket br(a, "psi_conv3_dw"); LAUNCH(k_igemm_red<DwC3>, dim3(9 * chunks), d); }
                const int n = 576 * 64 + 64;
                hipLaunchKernelGGL(k_reduce_partials, dim3((n + 63) / 64), dim3(256), 0, stream, part_conv + pl.off_c3, pl.stride_c3, chunks, grad + conv.w3, n, 576 * 64, 1.0f);
                BDR_HIP(hipGetLastError());
            }
            {   // position-class tiles (cnn_layers.hpp DxC3PosP: only the taps that reach a valid output; bit-identical to the flat row tiles)
                DxArgs d{dy3, p + conv.w3, a2, dy2, Bn * 81, nullptr, 0};
                Bracket br(a, "psi_conv3_dx");
                BDR_HIP((launch_igemm<DxC3Pos, 2>(stream, dim3(((Bn + DxC3Pos::WM * DxC3Pos::TM * 32 - 1) / (DxC3Pos::WM * DxC3Pos::TM * 32)) * n_tiles<DxC3Pos>(), 81, 1), d)));
            }
            {
                const int Mr = Bn * 81, chunks = std::min(pl.chunks_c2, (Mr + 31) / 32);
                DwArgs d{a1, dy2, part_conv + pl.off_c2, pl.stride_c2, Mr};
                { Bracket br(a, "psi_conv2_dw"); LAUNCH(k_igemm_red<DwC2>, dim3(8 * chunks), d); }
                const int n = 512 * 64 + 64;
                hipLaunchKernelGGL(k_reduce_partials, dim3((n + 63) / 64), dim3(256), 0, stream, part_conv + pl.off_c2, pl.stride_c2, chunks, grad + conv.w2, n, 512 * 64, 1.0f);
                BDR_HIP(hipGetLastError());
            }
            {   // the four parity classes as one GEMM over position-class tiles (DxC2MPosP), as the DQN step
                DxArgs d{dy2, p + conv.w2, a1, dy1, Bn * 100, nullptr, 0};
                Bracket br(a, "psi_conv2_dx");
                BDR_HIP((launch_igemm<DxC2MPos, 1>(stream, dxc2_pos_grid<DxC2MPos>(Bn), d)));
            }
            {
                const int chunks = std::min(pl.chunks_c1, Bn);
                Conv1DwArgs d{obs, dy1, part_conv + pl.off_c1, pl.stride_c1, Bn};
                { Bracket br(a, "psi_conv1_dw"); BDR_HIP(launch_conv1_dw_bf16(conv.ns, dim3(chunks), a->stream, d)); }
                const int nw = (int)conv.n_w1(), n = nw + 32;
                hipLaunchKernelGGL(k_reduce_partials, dim3((n + 63) / 64), dim3(256), 0, stream, part_conv + pl.off_c1, pl.stride_c1, chunks, grad + conv.w1, n, nw, INV255);
                BDR_HIP(hipGetLastError());
            }
        } else {
            const int PL = (int)psi_mlp.L.size();
            for (int i = PL - 1; i >= 0; --i) {
                DenseSrc in = i == 0 ? DenseSrc{x_in, psi_mlp.L[0].Kp} : DenseSrc{psi_act[i - 1], psi_mlp.L[i - 1].Np};
                { Bracket br(a, "psi_dw"); BDR_TRY(dense_dw(stream, psi_mlp.L[i], grad, in, psi_dy[i], Bn)); }
                if (i > 0) { Bracket br(a, "psi_dx"); BDR_TRY(dense_dx(stream, psi_mlp.L[i], p, psi_dy[i], psi_dy[i - 1], psi_act[i - 1], Bn)); }
            }
        }
        adam_step += 1;
        {   // IqnModel::backward_step (iqn/model/base.rs) -> opt.rs:74-83; OptimizerConfig::{Adam, AdamW} (opt.rs:30-57)
            const bdr_adamw_config& o = cfg.opt;
            const AdamScalars sc = adam_scalars_for(o.opt_kind == BDR_OPT_ADAMW, cfg.lr, o.beta1, o.beta2, o.eps, o.weight_decay, adam_step);
            Bracket br(a, "adam");
            if (avmax) BDR_TRY(launch_adam_amsgrad(stream, p, grad, am, av, avmax, total, sc));
            else BDR_TRY(launch_adam(stream, p, grad, am, av, total, sc));
        }
        return BDR_OK;
    }

    int32_t after_updates()   // iqn/base.rs:180-188
    {
        soft_update_counter += 1;
        if (soft_update_counter == cfg.soft_update_interval) {
            soft_update_counter = 0;
            Bracket br(this, "track");
            BDR_TRY(launch_track(stream, p_tgt, p, total, cfg.tau));
        }
        n_opts += 1;
        return BDR_OK;
    }

    int32_t fill_tau(float* dst, int mode, int Bn)
    {
        const int N = sample_points(mode);
        if (sample_is_uniform(mode)) {
            const size_t n = (size_t)Bn * N;
            hipLaunchKernelGGL(k_rand_uniform, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, dst, n, cfg.seed, noise_counter);
            BDR_HIP(hipGetLastError());
            noise_counter += n;
        } else {
            std::vector<float> t, all;
            sample_const(mode, t);
            for (int b = 0; b < Bn; ++b) all.insert(all.end(), t.begin(), t.end());
            BDR_HIP(hipMemcpyAsync(dst, all.data(), all.size() * 4, hipMemcpyHostToDevice, stream));
            BDR_HIP(hipStreamSynchronize(stream));
        }
        return BDR_OK;
    }

    const char* kind() const override { return "iqn"; }
    int32_t opt(bdr_replay* r) override
    {
        const uint64_t ob = cnn ? 7056ull * conv.ns : (uint64_t)in_dim * 4;
        BDR_REQUIRE(r->obs_bytes == ob && r->act_bytes >= 8, "replay rows do not match the IQN feature extractor input");
        BDR_REQUIRE(r->device == device, "agent and replay buffer live on different devices");
        const int Bn = (int)cfg.batch_size, Np = sample_points(cfg.sample_percents_pred), Nt = sample_points(cfg.sample_percents_tgt);
        BDR_TRY(ensure_batch(Bn, std::max(Np, Nt)));
        for (uint64_t u = 0; u < cfg.n_updates_per_opt; ++u) {
            { Bracket br(this, "sample"); BDR_TRY(replay_sample_on_stream(r, Bn, stream)); }
            BDR_TRY(fill_tau(tau_p, cfg.sample_percents_pred, Bn));
            BDR_TRY(fill_tau(tau_t, cfg.sample_percents_tgt, Bn));
            BDR_TRY(update_critic(Bn, r->b_obs, r->b_next, r->b_act, (int)r->act_bytes, r->b_reward, r->b_term, tau_p, Np, tau_t, Nt, u == 0));
        }
        n_updates_done = (int)cfg.n_updates_per_opt;
        return after_updates();
    }
    void record_keys(std::vector<std::string>& keys) override { keys = {"loss_critic"}; }
    int32_t noise(float* dev, size_t n) override   // the U[0,1) stream fill_tau draws percent points from
    {
        hipLaunchKernelGGL(k_rand_uniform, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, dev, n, cfg.seed, noise_counter);
        BDR_HIP(hipGetLastError());
        noise_counter += n;
        return BDR_OK;
    }
    int32_t record(float* out, int, int* n) override   // {"loss_critic"} (iqn/base.rs:190)
    {
        float l = 0;
        BDR_HIP(hipMemcpyAsync(&l, loss, 4, hipMemcpyDeviceToHost, stream));
        BDR_HIP(hipStreamSynchronize(stream));
        out[0] = l / (float)std::max(1, n_updates_done);
        *n = 1;
        return BDR_OK;
    }

    // ---- reference <-> internal parameter layouts ---------------------------------------------------
    int fperm(int j) const { return cnn ? (j % 49) * 64 + j / 49 : j; }   // ref feature (c*49+hw) -> internal (hw*64+c)
    void to_internal(const float* ref, float* in) const
    {
        std::fill(in, in + total, 0.f);
        const float* q = ref;
        if (cnn) {
            const int K1 = 64 * conv.ns;
            for (int o = 0; o < 32; ++o) for (int k = 0; k < K1; ++k) in[conv.w1 + (size_t)k * 32 + o] = q[(size_t)o * K1 + k];
            q += (size_t)32 * K1; std::copy(q, q + 32, in + conv.b1); q += 32;
            for (int o = 0; o < 64; ++o) for (int c = 0; c < 32; ++c) for (int kh = 0; kh < 4; ++kh) for (int kw = 0; kw < 4; ++kw)
                in[conv.w2 + (size_t)((kh * 4 + kw) * 32 + c) * 64 + o] = q[((size_t)(o * 32 + c) * 4 + kh) * 4 + kw];
            q += 32768; std::copy(q, q + 64, in + conv.b2); q += 64;
            for (int o = 0; o < 64; ++o) for (int c = 0; c < 64; ++c) for (int kh = 0; kh < 3; ++kh) for (int kw = 0; kw < 3; ++kw)
                in[conv.w3 + (size_t)((kh * 3 + kw) * 64 + c) * 64 + o] = q[((size_t)(o * 64 + c) * 3 + kh) * 3 + kw];
            q += 36864; std::copy(q, q + 64, in + conv.b3); q += 64;
        } else {
            mlp_to_internal(psi_mlp, 0, q, in);
            q += psi_mlp.ref_total;
        }
        // cos layer: ref weight [F][E], bias [F]
        const auto& c0 = hd.L[0];
        for (int j = 0; j < F; ++j) for (int i = 0; i < E; ++i) in[c0.w + (size_t)i * c0.Np + fperm(j)] = q[(size_t)j * E + i];
        q += (size_t)F * E;
        for (int j = 0; j < F; ++j) in[c0.b + fperm(j)] = q[j];
        q += F;
        for (size_t li = 1; li < hd.L.size(); ++li) {
            const auto& l = hd.L[li];
            for (int o = 0; o < l.out; ++o) for (int k = 0; k < l.in; ++k)
                in[l.w + (size_t)(li == 1 ? fperm(k) : k) * l.Np + o] = q[(size_t)o * l.in + k];
            q += (size_t)l.out * l.in;
            for (int o = 0; o < l.out; ++o) in[l.b + o] = q[o];
            q += l.out;
        }
    }
    void to_reference(const float* in, float* ref) const
    {
        float* q = ref;
        if (cnn) {
            const int K1 = 64 * conv.ns;
            for (int o = 0; o < 32; ++o) for (int k = 0; k < K1; ++k) q[(size_t)o * K1 + k] = in[conv.w1 + (size_t)k * 32 + o];
            q += (size_t)32 * K1; std::copy(in + conv.b1, in + conv.b1 + 32, q); q += 32;
            for (int o = 0; o < 64; ++o) for (int c = 0; c < 32; ++c) for (int kh = 0; kh < 4; ++kh) for (int kw = 0; kw < 4; ++kw)
                q[((size_t)(o * 32 + c) * 4 + kh) * 4 + kw] = in[conv.w2 + (size_t)((kh * 4 + kw) * 32 + c) * 64 + o];
            q += 32768; std::copy(in + conv.b2, in + conv.b2 + 64, q); q += 64;
            for (int o = 0; o < 64; ++o) for (int c = 0; c < 64; ++c) for (int kh = 0; kh < 3; ++kh) for (int kw = 0; kw < 3; ++kw)
                q[((size_t)(o * 64 + c) * 3 + kh) * 3 + kw] = in[conv.w3 + (size_t)((kh * 3 + kw) * 64 + c) * 64 + o];
            q += 36864; std::copy(in + conv.b3, in + conv.b3 + 64, q); q += 64;
        } else {
            mlp_to_reference(psi_mlp, 0, in, q);
            q += psi_mlp.ref_total;
        }
        const auto& c0 = hd.L[0];
        for (int j = 0; j < F; ++j) for (int i = 0; i < E; ++i) q[(size_t)j * E + i] = in[c0.w + (size_t)i * c0.Np + fperm(j)];
        q += (size_t)F * E;
        for (int j = 0; j < F; ++j) q[j] = in[c0.b + fperm(j)];
        q += F;
        for (size_t li = 1; li < hd.L.size(); ++li) {
            const auto& l = hd.L[li];
            for (int o = 0; o < l.out; ++o) for (int k = 0; k < l.in; ++k)
                q[(size_t)o * l.in + k] = in[l.w + (size_t)(li == 1 ? fperm(k) : k) * l.Np + o];
            q += (size_t)l.out * l.in;
            for (int o = 0; o < l.out; ++o) q[o] = in[l.b + o];
            q += l.out;
        }
    }
    float* arena_ptr(int which)
    {
        switch (which) { case 0: return p; case 1: return p_tgt; case 2: return am; case 3: return av; case 4: return grad; case 5: return avmax; default: return nullptr; }
    }
    uint64_t param_count(int which) override { return which == -1 ? (uint64_t)A : ref_total; }
    int32_t get_params(int which, float* out, uint64_t n) override
    {
        float* src = arena_ptr(which);
        BDR_REQUIRE(src, "which must be 0..4 (5: max_exp_avg_sq of AdamW{amsgrad})");
        BDR_REQUIRE(n == ref_total, "parameter count mismatch (%llu vs %llu)", (unsigned long long)n, (unsigned long long)ref_total);
        std::vector<float> in(total);
        BDR_HIP(hipMemcpyAsync(in.data(), src, total * 4, hipMemcpyDeviceToHost, stream));
        BDR_HIP(hipStreamSynchronize(stream));
        to_reference(in.data(), out);
        return BDR_OK;
    }
    int32_t set_params(int which, const float* inp, uint64_t n) override
    {
        float* dst = arena_ptr(which);
        BDR_REQUIRE(dst, "which must be 0..4 (5: max_exp_avg_sq of AdamW{amsgrad})");
        BDR_REQUIRE(n == ref_total, "parameter count mismatch");
        std::vector<float> in(total);
        to_internal(inp, in.data());
        BDR_HIP(hipMemcpyAsync(dst, in.data(), total * 4, hipMemcpyHostToDevice, stream));
        BDR_HIP(hipStreamSynchronize(stream));
        return BDR_OK;
    }
    float* arena(int which, size_t* n) override { if (n) *n = total; return arena_ptr(which); }
    std::vector<NamedTensor> meta() const
    {
        std::vector<NamedTensor> mt;
        if (cnn) {
            mt = {{"c1.weight", {32, (uint64_t)conv.ns, 8, 8}}, {"c1.bias", {32}}, {"c2.weight", {64, 32, 4, 4}}, {"c2.bias", {64}}, {"c3.weight", {64, 64, 3, 3}}, {"c3.bias", {64}}};
        } else {
            for (size_t i = 0; i < psi_mlp.L.size(); ++i) {
                mt.push_back({"psi.mlp.ln" + std::to_string(i) + ".weight", {(uint64_t)psi_mlp.L[i].out, (uint64_t)psi_mlp.L[i].in}});
                mt.push_back({"psi.mlp.ln" + std::to_string(i) + ".bias", {(uint64_t)psi_mlp.L[i].out}});
            }
        }
        mt.push_back({"iqn_cos_to_feature.weight", {(uint64_t)F, (uint64_t)E}});
        mt.push_back({"iqn_cos_to_feature.bias", {(uint64_t)F}});
        for (size_t i = 1; i < hd.L.size(); ++i) {
            mt.push_back({"mlp.ln" + std::to_string(i - 1) + ".weight", {(uint64_t)hd.L[i].out, (uint64_t)hd.L[i].in}});
            mt.push_back({"mlp.ln" + std::to_string(i - 1) + ".bias", {(uint64_t)hd.L[i].out}});
        }
        return mt;
    }
    int32_t save(const char* dir) override   // iqn.pt.tch / iqn_tgt.pt.tch stems
    {
        std::vector<float> ref(ref_total);
        BDR_TRY(get_params(0, ref.data(), ref.size()));
        BDR_TRY(save_named(ckpt_save_path(this, dir, "iqn"), meta(), ref.data(), ref.size()));
        BDR_TRY(get_params(1, ref.data(), ref.size()));
        return save_named(ckpt_save_path(this, dir, "iqn_tgt"), meta(), ref.data(), ref.size());
    }
    int32_t load(const char* dir) override
    {
        std::vector<float> ref(ref_total);
        BDR_TRY(load_named(ckpt_load_path(this, dir, "iqn"), meta(), ref.data(), ref.size()));
        BDR_TRY(set_params(0, ref.data(), ref.size()));
        BDR_TRY(load_named(ckpt_load_path(this, dir, "iqn_tgt"), meta(), ref.data(), ref.size()));
        return set_params(1, ref.data(), ref.size());
    }
    int32_t stage(uint64_t n, const void* obs, const int64_t* act, const void* next_obs, const float* reward, const int8_t* term)
    {
        const size_t ob = cnn ? (size_t)7056 * conv.ns : (size_t)in_dim * 4;
        if (n > u_cap) {
            BDR_HIP(hipStreamSynchronize(stream));
            (void)hipFree(u_obs); (void)hipFree(u_next); (void)hipFree(u_act); (void)hipFree(u_rew); (void)hipFree(u_term);
            BDR_HIP(hipMalloc((void**)&u_obs, n * ob)); BDR_HIP(hipMalloc((void**)&u_next, n * ob));
            BDR_HIP(hipMalloc((void**)&u_act, n * 8)); BDR_HIP(hipMalloc((void**)&u_rew, n * 4)); BDR_HIP(hipMalloc((void**)&u_term, round_up(n, 16)));
            u_cap = n;
        }
        BDR_TRY(stage_obs(u_obs, obs, ob, n, stream));   // (host rows; device rows inside bdr_agent_sample_device)
        if (next_obs) BDR_HIP(hipMemcpyAsync(u_next, next_obs, n * ob, hipMemcpyHostToDevice, stream));
        if (act) BDR_HIP(hipMemcpyAsync(u_act, act, n * 8, hipMemcpyHostToDevice, stream));
        if (reward) BDR_HIP(hipMemcpyAsync(u_rew, reward, n * 4, hipMemcpyHostToDevice, stream));
        if (term) BDR_HIP(hipMemcpyAsync(u_term, term, n, hipMemcpyHostToDevice, stream));
        return BDR_OK;
    }
};

extern "C" {

void bdr_iqn_config_default(bdr_iqn_config* c)
{
    if (!c) return;
    memset(c, 0, sizeof *c);
    // iqn/config.rs:50-67
    c->psi.kind = BDR_NET_ATARI_CNN; c->psi.n_stack = 4; c->feature_dim = 3136; c->embed_dim = 64;
    c->soft_update_interval = 1; c->n_updates_per_opt = 1; c->batch_size = 1; c->discount_factor = 0.99; c->tau = 0.005;
    c->sample_percents_pred = BDR_IQN_UNIFORM8; c->sample_percents_tgt = BDR_IQN_UNIFORM8; c->sample_percents_act = BDR_IQN_CONST32;
    c->train = 0; c->device = -1;
    c->opt.opt_kind = BDR_OPT_ADAM; c->opt.beta1 = 0.9; c->opt.beta2 = 0.999; c->opt.weight_decay = 0.0; c->opt.eps = 1e-8;
    c->arithmetic = BDR_ARITH_BF16X3_6;
}

int32_t bdr_iqn_create(const bdr_iqn_config* cfg, bdr_agent** out)
{
    BDR_REQUIRE(cfg && out, "null argument");
    BDR_REQUIRE(cfg->device >= 0, "No device is given for IQN agent");
    BDR_REQUIRE(cfg->n_actions >= 1 && cfg->n_actions <= 64, "n_actions must be in [1,64]");
    BDR_REQUIRE(cfg->embed_dim >= 1 && cfg->feature_dim >= 1, "bad embed / feature dims");
    BDR_REQUIRE(cfg->n_f_units >= 0 && cfg->n_f_units <= BDR_MAX_UNITS, "bad merge-net layer count");
    for (int m : {cfg->sample_percents_pred, cfg->sample_percents_tgt, cfg->sample_percents_act}) BDR_REQUIRE(m >= 0 && m <= 6, "unknown IqnSample");
    BDR_REQUIRE(cfg->batch_size >= 1 && cfg->batch_size <= 65536 && cfg->n_updates_per_opt >= 1 && cfg->soft_update_interval >= 1, "bad counts");
    BDR_REQUIRE(cfg->opt.opt_kind == BDR_OPT_ADAM || cfg->opt.opt_kind == BDR_OPT_ADAMW, "unknown optimizer");
    BDR_REQUIRE(cfg->arithmetic == BDR_ARITH_BF16X3_6 || cfg->arithmetic == BDR_ARITH_F32_EXACT, "unknown arithmetic %d (BDR_ARITH_*)", cfg->arithmetic);
    BDR_TRY(ensure_device(cfg->device));
    Iqn* a = new Iqn();
    a->cfg = *cfg; a->device = cfg->device; a->train = cfg->train != 0;
    a->cnn = cfg->psi.kind == BDR_NET_ATARI_CNN;
    a->b3_allowed = arith_is_split(cfg->arithmetic, "BDR_IQN_F32_EXACT");   // bdr_iqn_config::arithmetic; the variable overrides it for A/B runs only
    a->merge_epilogue = getenv("BDR_IQN_NO_MERGE_EPILOGUE") == nullptr;
    a->phi_b3 = getenv("BDR_IQN_PHI_F32") == nullptr;                         // (A/B switch: the cosine-embedding layer on the FP32 kernel)
    a->dw_b3 = getenv("BDR_IQN_DW_F32") == nullptr;                        // (A/B switch: the FP32-MFMA weight gradient of the merge layer)   // (A/B switch: the separate k_iqn_merge_bwd pass)
    a->F = cfg->feature_dim; a->E = cfg->embed_dim; a->A = cfg->n_actions;
    size_t o = 0;
    if (a->cnn) {
        BDR_REQUIRE(cfg->feature_dim == 3136, "the AtariCnn{skip_linear} trunk yields 64 x 7 x 7 = 3136 features (cnn/base.rs:38-45)");
        BDR_REQUIRE(cfg->psi.n_stack >= 1 && cfg->psi.n_stack <= bdr::C1_MAX_STACK, "AtariCnnConfig::n_stack must be in [1, %d] (conv1's kernels are instantiated per depth)", bdr::C1_MAX_STACK);
        a->conv = make_arena(1, cfg->psi.n_stack);
        o = a->ref_total = conv_floats(cfg->psi.n_stack);
    } else {
        BDR_REQUIRE(cfg->psi.out_dim == cfg->feature_dim, "psi.out_dim must equal feature_dim");
        a->in_dim = cfg->psi.in_dim;
        a->psi_mlp = make_mlp(cfg->psi.in_dim, cfg->psi.units, cfg->psi.n_units, cfg->feature_dim, cfg->psi.activation_out != 0);
        o = a->psi_mlp.total; a->ref_total = a->psi_mlp.ref_total;
    }
    {   // head: cos-embed layer then f
        DenseLayer c0; c0.in = a->E; c0.out = a->F; c0.Kp = pad64(a->E); c0.Np = pad64(a->F); c0.w = o; o += (size_t)c0.Kp * c0.Np; c0.b = o; o += c0.Np; c0.relu = 1;
        a->hd.L.push_back(c0);
        MlpLayout f = make_mlp(a->F, cfg->f_units, cfg->n_f_units, a->A, false, o);
        for (auto& l : f.L) a->hd.L.push_back(l);
        o += f.total;
        a->ref_total += (size_t)a->F * a->E + a->F + f.ref_total;
    }
    a->total = o;
    BDR_HIP(hipStreamCreateWithFlags(&a->stream, hipStreamNonBlocking));
    BDR_TRY(a->err_init());
    float** arenas[5] = {&a->p, &a->p_tgt, &a->grad, &a->am, &a->av};
    for (auto q : arenas) BDR_TRY(a->zalloc(q, a->total));
    if (cfg->opt.opt_kind == BDR_OPT_ADAMW && cfg->opt.amsgrad) BDR_TRY(a->zalloc(&a->avmax, a->total));
    BDR_TRY(a->zalloc(&a->loss, 4));
    {   // library initialiser (uniform +-1/sqrt(fan_in)); model cloned into its target (IqnModel::clone)
        std::vector<float> ref(a->ref_total);
        uint64_t s = cfg->seed * 0x9E3779B97F4A7C15ull + 0x4242ull;
        for (auto& v : ref) { s += 0x9E3779B97F4A7C15ull; uint64_t x = s; x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31; v = ((float)(x >> 40) * (2.0f / 16777216.0f) - 1.0f) * 0.05f; }
        BDR_TRY(a->set_params(0, ref.data(), ref.size()));
        BDR_TRY(a->set_params(1, ref.data(), ref.size()));
    }
    BDR_TRY(a->ensure_batch((int)cfg->batch_size, std::max(sample_points(cfg->sample_percents_pred), sample_points(cfg->sample_percents_tgt))));
    *out = a;
    return BDR_OK;
}

// One Iqn::opt_ update on a host minibatch with injected percent points tau_pred [n][n_pred], tau_tgt [n][n_tgt]
int32_t bdr_iqn_update_on_batch(bdr_agent* base, uint64_t n, const void* obs, const int64_t* act, const void* next_obs,
                                const float* reward, const int8_t* term, const float* tau_pred, int32_t n_pred,
                                const float* tau_tgt, int32_t n_tgt, float* loss_out)
{
    BDR_REQUIRE(base && obs && act && next_obs && reward && term && tau_pred && tau_tgt, "null argument");
    BDR_REQUIRE(!strcmp(base->kind(), "iqn"), "not an IQN agent");
    BDR_REQUIRE(n >= 1 && n <= 65536 && n_pred >= 1 && n_pred <= 256 && n_tgt >= 1 && n_tgt <= 256, "sizes out of range");
    Iqn* a = static_cast<Iqn*>(base);
    BDR_HIP(hipSetDevice(a->device));
    BDR_TRY(a->ensure_batch((int)n, std::max(n_pred, n_tgt)));
    BDR_TRY(a->stage(n, obs, act, next_obs, reward, term));
    BDR_HIP(hipMemcpyAsync(a->tau_p, tau_pred, n * n_pred * 4, hipMemcpyHostToDevice, a->stream));
    BDR_HIP(hipMemcpyAsync(a->tau_t, tau_tgt, n * n_tgt * 4, hipMemcpyHostToDevice, a->stream));
    BDR_TRY(a->update_critic((int)n, a->u_obs, a->u_next, a->u_act, 8, a->u_rew, a->u_term, a->tau_p, n_pred, a->tau_t, n_tgt, true));
    a->n_updates_done = 1;
    BDR_TRY(a->after_updates());
    prof_collect(a);
    if (loss_out) BDR_HIP(hipMemcpyAsync(loss_out, a->loss, 4, hipMemcpyDeviceToHost, a->stream));
    BDR_HIP(hipStreamSynchronize(a->stream));
    return BDR_OK;
}

// z = IqnModel::forward(obs, tau) -> z_out [n][n_tau][A]; which: 0 online, 1 target  (parity probe)
int32_t bdr_iqn_forward(bdr_agent* base, int32_t which, uint64_t n, const void* obs, const float* tau, int32_t n_tau, float* z_out)
{
    BDR_REQUIRE(base && obs && tau && z_out, "null argument");
    BDR_REQUIRE(!strcmp(base->kind(), "iqn"), "not an IQN agent");
    Iqn* a = static_cast<Iqn*>(base);
    BDR_HIP(hipSetDevice(a->device));
    BDR_TRY(a->ensure_batch((int)n, n_tau));
    BDR_TRY(a->stage(n, obs, nullptr, nullptr, nullptr, nullptr));
    BDR_HIP(hipMemcpyAsync(a->tau_p, tau, n * n_tau * 4, hipMemcpyHostToDevice, a->stream));
    BDR_TRY(a->model_forward(which ? a->p_tgt : a->p, a->u_obs, a->tau_p, (int)n, n_tau));
    const int ldz = a->hd.L.back().Np;
    std::vector<float> tmp((size_t)n * n_tau * ldz);
    BDR_HIP(hipMemcpyAsync(tmp.data(), a->f_act.back(), tmp.size() * 4, hipMemcpyDeviceToHost, a->stream));
    BDR_HIP(hipStreamSynchronize(a->stream));
    a->slot_cursor = 0;
    for (size_t r = 0; r < (size_t)n * n_tau; ++r) for (int k = 0; k < a->A; ++k) z_out[r * a->A + k] = tmp[r * ldz + k];
    return BDR_OK;
}

// Policy::sample, greedy part (iqn/base.rs:204-228): action values averaged over sample_percents_act
int32_t bdr_iqn_qvalues(bdr_agent* base, uint64_t n, const void* obs, float* q_out, int64_t* argmax_out)
{
    BDR_REQUIRE(base && obs, "null argument");
    BDR_REQUIRE(!strcmp(base->kind(), "iqn"), "not an IQN agent");
    Iqn* a = static_cast<Iqn*>(base);
    BDR_HIP(hipSetDevice(a->device));
    const int N = sample_points(a->cfg.sample_percents_act);
    BDR_TRY(a->ensure_batch((int)n, N));
    const uint8_t* rows = nullptr;
    if (a->cnn && n <= (uint64_t)ACT_SMALL_MAX && !a->obs_rows_on_device) {
        // host rows of an acting call: pinned memory the device reads in place (as DqnCnn's acting path)
        const size_t ob = (size_t)7056 * a->conv.ns;
        BDR_TRY(a->host_rows_pinned(obs, n * ob, &rows));
    } else {
        BDR_TRY(a->stage(n, obs, nullptr, nullptr, nullptr, nullptr));
        rows = a->u_obs;
    }
    BDR_TRY(a->fill_tau(a->tau_p, a->cfg.sample_percents_act, (int)n));
    a->acting = true;
    const int32_t st_fwd = a->model_forward(a->p, rows, a->tau_p, (int)n, N);
    a->acting = false;
    BDR_TRY(st_fwd);
    hipLaunchKernelGGL(k_iqn_average, dim3((unsigned)((n * a->A + 255) / 256)), dim3(256), 0, a->stream, a->f_act.back(), a->hd.L.back().Np, a->qavg, (int)n, N, a->A);
    BDR_HIP(hipGetLastError());
    std::vector<float> q(n * a->A);
    BDR_TRY(a->rows_to_host(a->qavg, q.data(), q.size()));   // (pinned path for acting-sized results: agent_base.hpp)
    a->slot_cursor = 0;
    if (q_out) memcpy(q_out, q.data(), q.size() * 4);
    if (argmax_out)
        for (uint64_t i = 0; i < n; ++i) {
            int best = 0;
            for (int k = 1; k < a->A; ++k) if (q[i * a->A + k] > q[i * a->A + best]) best = k;
            argmax_out[i] = best;
        }
    return BDR_OK;
}

}  // extern "C"
