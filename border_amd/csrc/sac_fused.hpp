// Row-block kernels of the SAC step (included by sac.hip): the narrow layers of the step - the actor's two heads (256 -> A), the
// critics' last layer (256 -> 1) and first-layer input gradient (only the A action columns are needed) - are not worth a launch
// of their own (a dependent kernel costs >= 2.4 us whatever it does, LAB.md 5) and neither are the row-wise kernels between
// them.  A workgroup here takes 32 batch rows, forms the narrow layers' 32x32 tiles with the SAME tile function as the
// layer-by-layer path (dense_small_tile / dense_small_sum: same k-slices, same MFMA order, same four-way sum) and does the row-wise
// math of the step on them in registers / LDS:
//   k_sac_heads_action   heads (ml, sl) + action_logp                               (sac/base.rs:73-87;   3 launches -> 1)
//   k_sac_q_last         critics' last layer for Q_i(obs, a_pi) and Q_i(obs, act) + the qmin selection + d qmin / d h2 +
//                        (last workgroup) EntCoef::update and the actor loss          (sac/base.rs:89-105, 151-160; 3 -> 1)
//   k_sac_actor_bwd      d qmin / d a (critics' first layer, action columns only) + the tanh-Gaussian backward + both heads' input
//                        gradient                                                     (sac/base.rs:160-167;  4 -> 1)
//   k_sac_td_last        target critics' last layer + TD target + critic losses + d loss / d h2 + (last workgroup) the loss sums
//                                                                                      (sac/base.rs:107-149;  3 -> 1)
// 30 launches per update become 21 (17 with dense_chain.hpp).  Reductions over the whole batch (mean log p for the entropy coefficient, the recorded losses)
// are done by the LAST workgroup to finish (atomic ticket), in exactly the order of the single-workgroup kernels they replace
// (block_sum_1024 with its 16 wave totals): every path - fused, layer by layer, captured graph - gives the same bits
// (tests/test_gpu_sac.py).
#pragma once

namespace {

#ifndef SF_ABL      // tools only: timing ablations of k_sac_q_last (results are wrong when non-zero): 1 no last-workgroup part, 2 no dh loop, 4 no tiles
#define SF_ABL 0
#endif
constexpr int SF_LDG = 68;   // row stride (floats) of a 32 x 64 A operand kept in LDS (conflict-light ds_read_b128)

// the last workgroup's side of a batch-wide row sum (sac.hip row_sum_1024): the block partials, one by one in block order
__device__ __forceinline__ float sum_partials(const float* part, int nb, float* lds /* [>= nb] */)
{
    for (int k = threadIdx.x; k < nb; k += blockDim.x) lds[k] = __hip_atomic_load(part + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    float t = 0.f;
    for (int k = 0; k < nb; ++k) t += lds[k];
    __syncthreads();
    return t;
}


// ------------------------------------------------------------------------------------------------------------------------
struct SacHeadsActionArgs {
    const float* h; int ldh;            // last trunk activation [B][ldh]
    HeadRef hm, hs; int kred, w_ld;     // heads: W [kred][w_ld] + bias (w_ld = Np of the heads)
    float* mean; float* e; int ld;      // head outputs [B][ld] (kept: probes / parity with the layer-by-layer path)
    const float* z;
    float* xq; int ldq; int col0;
    float* a_out; float* s_out; float* sd_out;
    float* logp;
    int B, A; float lo, hi, eps;
};
// row `row` (r of its block) by one wave, lanes over the action dimension: k_sac_action's arithmetic and its butterfly sums (shared by k_sac_heads_action
// and k_sac_pi_chain_heads: one source, one instruction sequence)
__device__ __forceinline__ void sac_heads_row(const SacHeadsActionArgs& p, const float (*red_m)[32][33], const float (*red_s)[32][33], int r, int row, int lane,
                                              float bias_m, float bias_s, float zv)
{
    float nl = 0.f, sl = 0.f;
    if (lane < 32) {   // columns of tile 0 (the rest of the padded width stays zero from allocation)
        float mv = dense_small_sum(red_m, r, lane) + bias_m;
        float ev = dense_small_sum(red_s, r, lane) + bias_s;
        if (p.hm.relu) mv = mv > 0.f ? mv : 0.f;
        if (p.hs.relu) ev = ev > 0.f ? ev : 0.f;
        const size_t q = (size_t)row * p.ld + lane;
        p.mean[q] = mv; p.e[q] = ev;
        if (lane < p.A) {
            const float z = zv;
            const float s = expf(ev);
            const float cl = fminf(fmaxf(s, p.lo), p.hi);
            const float sd = expf(cl);
            const float a = tanhf(sd * z + mv);
            p.xq[(size_t)row * p.ldq + p.col0 + lane] = a;
            if (p.a_out) { p.a_out[q] = a; p.s_out[q] = s; p.sd_out[q] = sd; }
            nl += -0.91893853320467274178f - 0.5f * (z * z);
            sl += logf((1.0f - a * a) + p.eps);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { nl += __shfl_xor(nl, off); sl += __shfl_xor(sl, off); }
    if (lane == 0) p.logp[row] = nl - sl;
}
// The same arithmetic for the wave's FOUR rows at once, 16 lanes per row (round 6; A <= 16): lane = (row group g, column c).  One pass of
// exponentials / tanh / log and one 4-step butterfly per wave instead of four passes with 6-step butterflies.  Same bits: the butterfly
// tree over the A real columns is the one sac_heads_row walks (its upper steps add exact zeros), and the padded columns 16 .. 31 of
// mean / e, which sac_heads_row rewrites with the zeros they hold from allocation (zero weights, zero bias), are simply left alone.
__device__ __forceinline__ void sac_heads_rows16(const SacHeadsActionArgs& p, const float (*red_m)[32][33], const float (*red_s)[32][33], int r0, int m0, int lane,
                                                 float bias_m, float bias_s, float zv)
{
    const int g = lane >> 4, c = lane & 15;
    const int r = r0 + g, row = m0 + r;
    const bool ok = row < p.B;
    float nl = 0.f, sl = 0.f;
    float mv = dense_small_sum(red_m, r, c) + bias_m;
    float ev = dense_small_sum(red_s, r, c) + bias_s;
    if (p.hm.relu) mv = mv > 0.f ? mv : 0.f;
    if (p.hs.relu) ev = ev > 0.f ? ev : 0.f;
    const size_t q = (size_t)(ok ? row : 0) * p.ld + c;
    float a = 0.f, s = 0.f, sd = 0.f;
    if (c < p.A) {
        const float z = zv;
        s = expf(ev);
        const float cl = fminf(fmaxf(s, p.lo), p.hi);
        sd = expf(cl);
        a = tanhf(sd * z + mv);
        nl = -0.91893853320467274178f - 0.5f * (z * z);
        sl = logf((1.0f - a * a) + p.eps);
    }
    if (ok) {
        p.mean[q] = mv; p.e[q] = ev;
        if (c < p.A) {
            p.xq[(size_t)row * p.ldq + p.col0 + c] = a;
            if (p.a_out) { p.a_out[q] = a; p.s_out[q] = s; p.sd_out[q] = sd; }
        }
    }
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) { nl += __shfl_xor(nl, off); sl += __shfl_xor(sl, off); }
    if (ok && c == 0) p.logp[row] = nl - sl;
}

__global__ __launch_bounds__(512) void k_sac_heads_action(SacHeadsActionArgs p)
{
    __shared__ float red[2][4][32][33];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int team = wave >> 2, w4 = wave & 3;
    const int m0 = (int)blockIdx.x * 32;
    const float* arow = p.h + (size_t)min(m0 + (lane & 31), p.B - 1) * p.ldh;
    const HeadRef& hd = team ? p.hs : p.hm;
    // operands of the row arithmetic that do not depend on the tiles: loaded beside the tile operands, not behind the barrier
    const bool grouped = p.A <= 16;   // (uniform) four rows per wave side by side, 16 lanes each
    const int bcol = grouped ? (lane & 15) : (lane & 31);
    const float bias_m = p.hm.bias[bcol], bias_s = p.hs.bias[bcol];
    float zrow[4];
    if (grouped) {
        const int row = min(m0 + wave * 4 + (lane >> 4), p.B - 1);
        zrow[0] = (lane & 15) < p.A ? p.z[(size_t)row * p.A + (lane & 15)] : 0.f;
        zrow[1] = zrow[2] = zrow[3] = 0.f;
    } else {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) zrow[rr] = lane < p.A ? p.z[(size_t)min(m0 + wave * 4 + rr, p.B - 1) * p.A + lane] : 0.f;
    }
    dense_small_tile<false>([&](int k) { return *reinterpret_cast<const f32x4*>(arow + k); }, hd.w, p.w_ld, 0, p.kred, w4, lane, red[team]);
    asm volatile("" ::"v"(zrow[0]), "v"(zrow[1]), "v"(zrow[2]), "v"(zrow[3]), "v"(bias_m), "v"(bias_s));   // landed before the first store (last_layer_dx_land)
    __syncthreads();
    if (grouped) { sac_heads_rows16(p, red[0], red[1], wave * 4, m0, lane, bias_m, bias_s, zrow[0]); return; }
    // one wave per row, lanes over the action dimension: k_sac_action's arithmetic and its butterfly sums
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int r = wave * 4 + rr, row = m0 + r;
        if (row >= p.B) break;
        sac_heads_row(p, red[0], red[1], r, row, lane, bias_m, bias_s, zrow[rr]);
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// last critic layer for up to 8 (parameters, input) pairs, pair j: h = hin[j] [B][ldh], W = wl[j], out q[j] [B][ldq] (tile 0)
struct SacQLastArgs {
    int npairs, NC;                       // pairs 0..NC-1: Q_i(obs, a_pi) (selection, gradient); NC..2NC-1: Q_i(obs, act) (values only)
    const float* hin[8]; HeadRef wl[8]; float* q[8];
    int ldh, kred, w_ld, ldq;
    float* dout[4];                       // [B][ldq], column 0: 1 for the selected critic
    float* dh[4];                         // d qmin / d h2 of critic i [B][ldh] (masked by relu'(hin[i]))
    const float* w_last[4];               // critic i's last-layer weights (column 0 is the real output)
    int B;
    // last workgroup: EntCoef::update + actor loss (k_sac_select)
    float* part;                          // [3][ceil(B / 32)] block partials: log_p + target, log_p, qmin
    unsigned* ticket;
    const float* logp; const float* log_alpha;
    float* out; float scale; int accumulate;
    int auto_alpha; float target; float* log_alpha_rw; float* al_m; float* al_v; AdamScalars s;
    const unsigned* poison;               // a cross-queue wait timed out: no EntCoef step
    unsigned long long* applied; unsigned long long step;   // as SacSelectArgs
    int tail_here;                        // 0: the batch-wide part runs as one more workgroup of the NEXT launch (k_dense_small_dx_tail), which nothing of it precedes
};
// relu'(h) * d[row] * W_last[:, 0] for 32 rows x the 64 columns [c0, c0 + 64) of one critic, in two halves: the operand loads do not
// depend on anything the kernel computes, so they are issued first thing (next to the tile operands) and are long back when the
// per-row factors exist.  (The whole 32 x ldh block in one workgroup, loaded where it was used, was 16 dependent round trips per
// thread: 11 us.)
struct LastDxRegs { float hv[4], wv[4]; };
__device__ __forceinline__ void last_layer_dx_load(LastDxRegs& g, const float* __restrict__ hin, const float* __restrict__ w, int w_ld, int ldh, int c0,
                                                   int m0, int B, int tid)
{
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int e = tid + 512 * u, r = e >> 6, k = c0 + (e & 63);
        g.hv[u] = hin[(size_t)min(m0 + r, B - 1) * ldh + k];
        g.wv[u] = w[(size_t)k * w_ld];
    }
}
// Lands the prefetched operands in their registers (call once the loads have had time to complete, BEFORE the kernel's first store): on
// gfx9 stores share vmcnt with loads and may be acknowledged out of order, so a load result first used behind a store costs the compiler
// an s_waitcnt vmcnt(0) - the store's round trip as well.
__device__ __forceinline__ void last_layer_dx_land(const LastDxRegs& g)
{
#pragma unroll
    for (int u = 0; u < 4; ++u) asm volatile("" ::"v"(g.hv[u]), "v"(g.wv[u]));
}
__device__ __forceinline__ void last_layer_dx_store(const LastDxRegs& g, const float* drow /* LDS [32] */, float* __restrict__ dh, int ldh, int c0, int m0,
                                                    int B, int tid)
{
    // values first, stores behind them: written as "load drow[r], store" per element the compiler put an s_waitcnt vmcnt(0) lgkmcnt(0) in
    // front of every store (vmcnt counts stores too), i.e. it waited out the previous store's acknowledgement - 16 serialised round trips
    // per call site in k_sac_q_last / k_sac_td_last
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int r = (tid + 512 * u) >> 6;
        v[u] = g.hv[u] > 0.f ? drow[r] * g.wv[u] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) asm volatile("" ::"v"(v[u]));
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int e = tid + 512 * u, r = e >> 6, k = c0 + (e & 63);
        if (m0 + r < B) dh[(size_t)(m0 + r) * ldh + k] = v[u];
    }
}

// grid (row blocks, ldh / 64 + 1): y < ldh / 64: the Q_i(obs, a_pi) passes (every y forms the tiles and the selection - cheap - and
// takes 64 columns of d qmin / d h2; y == 0 also stores the values); y == ldh / 64: the Q_i(obs, act) passes (values for the TD step)
__global__ __launch_bounds__(512) void k_sac_q_last(SacQLastArgs p)
{
    __shared__ float red[2][4][32][33];
    __shared__ float qv[4][32];
    __shared__ float sel[4][32];
    __shared__ float lsum[2048];
    __shared__ unsigned s_last;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int team = wave >> 2, w4 = wave & 3;
    const int CB = p.ldh / 64, y = (int)blockIdx.y;
    const bool act_pass = y == CB, writer = y == 0 || act_pass;
    const int m0 = (int)blockIdx.x * 32, jb = act_pass ? p.NC : 0;
    LastDxRegs dxr[4];
    if (!act_pass && !(SF_ABL & 2))
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < p.NC) last_layer_dx_load(dxr[i], p.hin[i], p.w_last[i], p.w_ld, p.ldh, y * 64, m0, p.B, tid);
    // everything else the kernel reads that it does not compute itself, issued now: each of these was one more dependent memory round
    // trip (1-2 us) behind a barrier.  The scalars are only used by the last workgroup - which one that will be is not known yet.
    float bias_pre[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) bias_pre[i] = i < p.NC ? p.wl[jb + i].bias[tid & 31] : 0.f;
    const float lp_pre = (!act_pass && tid < 32) ? p.logp[min(m0 + tid, p.B - 1)] : 0.f;
    const float la_pre = p.log_alpha[0], out_pre = p.accumulate ? p.out[0] : 0.f;
    const float am_pre = p.auto_alpha ? p.al_m[0] : 0.f, av_pre = p.auto_alpha ? p.al_v[0] : 0.f;
    for (int j0 = 0; j0 < p.NC; j0 += 2) {
        const int j = j0 + team;
        if (j < p.NC && !(SF_ABL & 4)) {   // team-uniform
            const float* arow = p.hin[jb + j] + (size_t)min(m0 + (lane & 31), p.B - 1) * p.ldh;
            dense_small_tile<false>([&](int k) { return *reinterpret_cast<const f32x4*>(arow + k); }, p.wl[jb + j].w, p.w_ld, 0, p.kred, w4, lane, red[team]);
        }
        if (j0 == 0) {   // the prefetched operands land before the first store of the kernel (last_layer_dx_land)
            if (!act_pass && !(SF_ABL & 2)) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (i < p.NC) last_layer_dx_land(dxr[i]);
            }
            asm volatile("" ::"v"(lp_pre), "v"(la_pre), "v"(out_pre), "v"(am_pre), "v"(av_pre));
        }
        __syncthreads();
        for (int e = tid; e < 2 * 32 * 32; e += 512) {
            const int t = e >> 10, r = (e >> 5) & 31, c = e & 31, jj = j0 + t;
            if (jj >= p.NC || m0 + r >= p.B) continue;
            float v = dense_small_sum(red[t], r, c) + (jj == 0 ? bias_pre[0] : jj == 1 ? bias_pre[1] : jj == 2 ? bias_pre[2] : bias_pre[3]);   // (c == tid & 31; selects, not a dynamically indexed register array)
            if (p.wl[jb + jj].relu) v = v > 0.f ? v : 0.f;
            if (c == 0) qv[jj][r] = v;
            if (!writer) continue;
            p.q[jb + jj][(size_t)(m0 + r) * p.ldq + c] = v;
        }
        __syncthreads();
    }
    if (!act_pass) {
        // selection (k_sac_select's row arithmetic) for this block's rows; y == 0 also leaves the block partials of the three sums
        if (tid < 64) {   // wave 0; lanes 32..63 idle through the butterflies
            const bool ok = tid < 32 && m0 + tid < p.B;
            float qm = 0.f, lp = 0.f;
            if (ok) {
                int im = 0;
                qm = qv[0][tid];
                for (int i = 1; i < p.NC; ++i) { const float v = qv[i][tid]; if (v < qm) { qm = v; im = i; } }
                for (int i = 0; i < p.NC; ++i) { const float d = i == im ? 1.0f : 0.0f; sel[i][tid] = d; if (writer) p.dout[i][(size_t)(m0 + tid) * p.ldq] = d; }
                lp = lp_pre;
            }
            if (writer) {
                const int nb = (p.B + 31) / 32;
                const float p1 = butterfly32(ok ? lp + p.target : 0.f), p2 = butterfly32(ok ? lp : 0.f), p3 = butterfly32(ok ? qm : 0.f);
                if (tid == 0) { st_agent(p.part + blockIdx.x, p1); st_agent(p.part + nb + blockIdx.x, p2); st_agent(p.part + 2 * nb + blockIdx.x, p3); }
            }
        }
    }
    if (SF_ABL & 1) return;
    // the partials are out: take the ticket now (its barrier also publishes `sel`) and let its round trip run under the gradient stores
    if (p.tail_here) ticket_take(p.ticket, gridDim.x * gridDim.y, &s_last);
    else __syncthreads();
    if (!act_pass) {
        // d qmin / d h2 = relu'(h2) * dout * W_last[:, 0]   (what the last layer's dX launch computes: its only non-zero term)
        if (!(SF_ABL & 2))
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (i < p.NC) last_layer_dx_store(dxr[i], sel[i], p.dh[i], p.ldh, y * 64, m0, p.B, tid);
    }
    if (!p.tail_here || !ticket_last(&s_last)) return;
    // ---- k_sac_select's batch-wide part, by the last workgroup to finish: the block partials in block order
    const int nb = (p.B + 31) / 32;
    float log_alpha = la_pre;
    if (p.auto_alpha) {
        const float g = -(sum_partials(p.part, nb, lsum) / (float)p.B);
        const float mm = am_pre * p.s.b1 + g * p.s.omb1;
        const float vv = av_pre * p.s.b2 + p.s.omb2 * g * g;
        const float denom = __fsqrt_rn(vv) / p.s.sqrt_bc2 + p.s.eps;
        log_alpha = log_alpha + p.s.neg_step * mm / denom;
        __syncthreads();
        if (tid == 0 && !(p.poison && *p.poison)) { p.log_alpha_rw[0] = log_alpha; p.al_m[0] = mm; p.al_v[0] = vv; if (p.applied) *p.applied = p.step; }
    }
    const float alpha = expf(log_alpha);
    const float s_logp = sum_partials(p.part + nb, nb, lsum);
    const float s_qm = sum_partials(p.part + 2 * nb, nb, lsum);
    if (tid == 0) p.out[0] = actor_loss_sum(out_pre, alpha, s_logp, s_qm, p.scale);
}

// ------------------------------------------------------------------------------------------------------------------------
struct SacActorBwdArgs {
    // d qmin / d a: critic i's first layer, transposed, restricted to the action columns (all inside one 32-column tile)
    int NC; const float* dy0[4]; int ldy0;             // gradient w.r.t. critic i's first hidden layer [B][ldy0]
    const float* w0[4]; int w0_ld; int kred0; int n0a; // first-layer weights W0 [Kq][w0_ld]; kred0 = Np of that layer; n0a = tile start
    int col0;                                          // first action column of the critic input
    // tanh-Gaussian backward (k_sac_actor_grad)
    const float* a; const float* s; const float* sd; int ld; const float* z; const float* log_alpha;
    float* gmean; float* ge;
    int B, A; float lo, hi, eps;
    // heads' input gradient: dh = relu'(h) * (gmean Wml^T) + relu'(h) * (ge Wsl^T)
    const float* wm; const float* ws; int wh_ld; int kredh;   // heads' weights [Hp][wh_ld], kredh = wh_ld (padded A)
    const float* hmask; int ldh; float* dh;                   // last trunk activation / its gradient [B][ldh]
};
__global__ __launch_bounds__(512) void k_sac_actor_bwd(SacActorBwdArgs p)
{
    __shared__ float red[2][4][32][33];
    __shared__ __attribute__((aligned(16))) float gm[32][SF_LDG];
    __shared__ __attribute__((aligned(16))) float gs[32][SF_LDG];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int team = wave >> 2, w4 = wave & 3;
    const int m0 = (int)blockIdx.x * 32, n0 = (int)blockIdx.y * 32;
    const int r = (tid & 255) >> 3, c4 = (tid & 7) * 4;          // epilogue element of threads 0..255
    // ---- operands that depend on nothing this kernel computes, issued now (see k_sac_q_last): the heads' weights of the second tile,
    // the action / sigma / noise values of the tanh-Gaussian backward, the ReLU mask of the output, the entropy coefficient
    f32x4 bh[8];
    dense_small_load_b<true>(team ? p.ws : p.wm, p.wh_ld, n0, p.kredh, w4, lane, bh);   // kredh = the heads' padded width <= 64
    const float la_pre = p.log_alpha[0];
    float a_pre[4] = {0.f, 0.f, 0.f, 0.f}, s_pre[4] = {0.f, 0.f, 0.f, 0.f}, sd_pre[4] = {0.f, 0.f, 0.f, 0.f}, z_pre[4] = {0.f, 0.f, 0.f, 0.f};
    f32x4 mk = {0.f, 0.f, 0.f, 0.f};
    if (tid < 256 && m0 + r < p.B) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int j = p.n0a + c4 + q - p.col0;
            if (j < 0 || j >= p.A) continue;
            const size_t t = (size_t)(m0 + r) * p.ld + j;
            a_pre[q] = p.a[t]; s_pre[q] = p.s[t]; sd_pre[q] = p.sd[t]; z_pre[q] = p.z[(size_t)(m0 + r) * p.A + j];
        }
        mk = *reinterpret_cast<const f32x4*>(p.hmask + (size_t)(m0 + r) * p.ldh + n0 + c4);
    }
    // ---- sum over the critics of d qmin / d a, critic by critic (k_sac_actor_grad's order, starting from 0); two critics per round
    float dq[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i0 = 0; i0 < p.NC; i0 += 2) {
        const int i = i0 + team;
        if (i < p.NC) {
            const float* arow = p.dy0[i] + (size_t)min(m0 + (lane & 31), p.B - 1) * p.ldy0;
            dense_small_tile<true>([&](int k) { return *reinterpret_cast<const f32x4*>(arow + k); }, p.w0[i], p.w0_ld, p.n0a, p.kred0, w4, lane, red[team]);
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            dq[q] += dense_small_sum(red[0], r, c4 + q);
            if (i0 + 1 < p.NC) dq[q] += dense_small_sum(red[1], r, c4 + q);
        }
        __syncthreads();
    }
    // ---- gmean / ge of this block's rows into LDS (A operands of the heads' dX) and, from the first column block, to memory
    for (int e = tid; e < 32 * p.ld; e += 512) { gm[e / p.ld][e % p.ld] = 0.f; gs[e / p.ld][e % p.ld] = 0.f; }
    __syncthreads();
    if (tid < 256) {
        const float alpha = expf(la_pre);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int j = p.n0a + c4 + q - p.col0, b = m0 + r;
            if (j < 0 || j >= p.A || b >= p.B) continue;
            const float a = a_pre[q], s = s_pre[q], sd = sd_pre[q];
            const float dlogp = (2.0f * a) / ((1.0f - a * a) + p.eps);
            const float ga = (alpha * dlogp - dq[q]) / (float)p.B;
            const float gu = ga * (1.0f - a * a);
            const float inr = (s >= p.lo && s <= p.hi) ? 1.0f : 0.0f;
            gm[r][j] = gu;
            gs[r][j] = gu * z_pre[q] * sd * inr * s;
        }
    }
    __syncthreads();
    if (blockIdx.y == 0)
        for (int e = tid; e < 32 * p.ld; e += 512) {
            const int rr = e / p.ld, j = e % p.ld;
            if (m0 + rr < p.B) { p.gmean[(size_t)(m0 + rr) * p.ld + j] = gm[rr][j]; p.ge[(size_t)(m0 + rr) * p.ld + j] = gs[rr][j]; }
        }
    // ---- both heads' input gradient for the 32 trunk features of this column block: team 0 gmean Wml^T, team 1 ge Wsl^T
    const int i32 = lane & 31;
    const float (*ga)[SF_LDG] = team ? gs : gm;
    dense_small_tile_pre([&](int k) { return *reinterpret_cast<const f32x4*>(&ga[i32][k]); }, bh, p.kredh, w4, lane, red[team]);
    __syncthreads();
    if (tid >= 256 || m0 + r >= p.B) return;
    const size_t o = (size_t)(m0 + r) * p.ldh + n0 + c4;
    f32x4 out;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float a1 = dense_small_sum(red[0], r, c4 + q), v2 = dense_small_sum(red[1], r, c4 + q);
        if (!(mk[q] > 0.f)) { a1 = 0.f; v2 = 0.f; }
        out[q] = v2 + a1;            // second launch of the layer-by-layer path: masked v2 += the stored, masked v1
    }
    *reinterpret_cast<f32x4*>(p.dh + o) = out;
}

// ------------------------------------------------------------------------------------------------------------------------
struct SacTdLastArgs {
    int NC;
    const float* hin_t[4]; HeadRef wl_t[4]; float* qt[4];   // target critics: last hidden activation on (next_obs, a'), last layer, output
    int ldh, kred, w_ld, ldq;
    const float* q[4];                                      // Q_i(obs, act) [B][ldq] (k_sac_q_last)
    const float* hin[4]; const float* w_last[4];            // online critics on (obs, act): last hidden activation, last-layer weights
    float* dout[4]; float* dh[4];
    const float* logp; const float* log_alpha; const float* reward; const int8_t* term; float gamma, reward_scale;
    float* tgt; float* part;                                // part: [NC][ceil(B / 32)] block partials of the critics' loss sums
    int B; int loss_kind;
    unsigned* ticket; float* out; float scale; int accumulate;
    int tail_here;                                          // as SacQLastArgs
};
__global__ __launch_bounds__(512) void k_sac_td_last(SacTdLastArgs p)
{
    __shared__ float red[2][4][32][33];
    __shared__ float qtv[4][32];
    __shared__ float dl[4][32];
    __shared__ float lsum[2048];
    __shared__ unsigned s_last;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int team = wave >> 2, w4 = wave & 3;
    const int m0 = (int)blockIdx.x * 32;
    const bool writer = blockIdx.y == 0;      // grid (row blocks, ldh / 64): every y forms the tiles and the row arithmetic, y takes 64 columns of dh
    LastDxRegs dxr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (i < p.NC) last_layer_dx_load(dxr[i], p.hin[i], p.w_last[i], p.w_ld, p.ldh, (int)blockIdx.y * 64, m0, p.B, tid);
    // the row arithmetic's operands and the last workgroup's scalar, issued now (see k_sac_q_last)
    float bias_pre[4], q_pre[4];
    const int b_pre = min(m0 + (tid & 31), p.B - 1);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        bias_pre[i] = i < p.NC ? p.wl_t[i].bias[tid & 31] : 0.f;
        q_pre[i] = (i < p.NC && tid < 64) ? p.q[i][(size_t)b_pre * p.ldq] : 0.f;
    }
    const float la_pre = p.log_alpha[0], out_pre = p.accumulate ? p.out[0] : 0.f;
    float rew_pre = 0.f, term_pre = 0.f, lp_pre = 0.f;
    if (tid < 64) { rew_pre = p.reward[b_pre]; term_pre = (float)p.term[b_pre]; lp_pre = p.logp[b_pre]; }
    for (int j0 = 0; j0 < p.NC; j0 += 2) {
        const int j = j0 + team;
        if (j < p.NC) {
            const float* arow = p.hin_t[j] + (size_t)min(m0 + (lane & 31), p.B - 1) * p.ldh;
            dense_small_tile<false>([&](int k) { return *reinterpret_cast<const f32x4*>(arow + k); }, p.wl_t[j].w, p.w_ld, 0, p.kred, w4, lane, red[team]);
        }
        if (j0 == 0) {   // the prefetched operands (d loss / d h2, the row scalars) land before the first store of the kernel (last_layer_dx_land)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (i < p.NC) { last_layer_dx_land(dxr[i]); asm volatile("" ::"v"(q_pre[i])); }
            asm volatile("" ::"v"(rew_pre), "v"(term_pre), "v"(lp_pre), "v"(la_pre), "v"(out_pre));
        }
        __syncthreads();
        for (int e = tid; e < 2 * 32 * 32; e += 512) {
            const int t = e >> 10, r = (e >> 5) & 31, c = e & 31, jj = j0 + t;
            if (jj >= p.NC || m0 + r >= p.B) continue;
            float v = dense_small_sum(red[t], r, c) + (jj == 0 ? bias_pre[0] : jj == 1 ? bias_pre[1] : jj == 2 ? bias_pre[2] : bias_pre[3]);   // (c == tid & 31)
            if (p.wl_t[jj].relu) v = v > 0.f ? v : 0.f;
            if (writer) p.qt[jj][(size_t)(m0 + r) * p.ldq + c] = v;
            if (c == 0) qtv[jj][r] = v;
        }
        __syncthreads();
    }
    // k_sac_critic_td's row arithmetic; y == 0 also leaves the block partials of the loss sums
    if (tid < 64) {
        const bool ok = tid < 32 && m0 + tid < p.B;
        const int b = min(m0 + tid, p.B - 1);
        const float alpha = expf(la_pre);
        float qm = qtv[0][tid & 31];
        for (int i = 1; i < p.NC; ++i) qm = fminf(qm, qtv[i][tid & 31]);
        const float tgt = sac_td_target(p.reward_scale, rew_pre, term_pre, p.gamma, qm, alpha, lp_pre);
        if (ok && writer) p.tgt[b] = tgt;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i >= p.NC) break;
            float l, g;
            sac_td_loss(q_pre[i], tgt, p.loss_kind, l, g);
            const float dv = g / (float)p.B;
            if (ok) { dl[i][tid] = dv; if (writer) p.dout[i][(size_t)b * p.ldq] = dv; }
            if (writer) {
                const float pi_ = butterfly32(ok ? l : 0.f);
                if (tid == 0) st_agent(p.part + (size_t)i * ((p.B + 31) / 32) + blockIdx.x, pi_);
            }
        }
    }
    // the partials are out: ticket first (its barrier also publishes `dl`), its round trip runs under the gradient stores
    if (p.tail_here) ticket_take(p.ticket, gridDim.x * gridDim.y, &s_last);
    else __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (i < p.NC) last_layer_dx_store(dxr[i], dl[i], p.dh[i], p.ldh, (int)blockIdx.y * 64, m0, p.B, tid);
    if (!p.tail_here || !ticket_last(&s_last)) return;
    const int nb = (p.B + 31) / 32;
    float total = out_pre;
    for (int i = 0; i < p.NC; ++i) total = add_scaled(total, sum_partials(p.part + (size_t)i * nb, nb, lsum), p.scale);
    if (tid == 0) p.out[0] = total;
}


// ------------------------------------------------------------------------------------------------------------------------
// The batch-wide parts of k_sac_q_last (EntCoef::update + the actor loss) and k_sac_td_last (the critics' loss sums) as ONE MORE WORKGROUP of the
// launch that follows them on the queue - the critics' layer-1 input gradient, which reads nothing they write.  Inside their own kernels these parts
// are a ticket (agent-scope partials acknowledged, an atomic's round trip) and then one workgroup's 2-3 dependent round trips while the queue waits for
// the kernel to end: 3.2 of k_sac_q_last's 13.9 us (LAB.md 5, round 3 stamps).  Behind a kernel boundary the partials are ordinary memory, and the
// sums run in the shadow of 512 tile workgroups.  Same partials, same order (block partials one by one in block order), same scalar arithmetic.
struct SacTailArgs {
    int kind;                             // 0: none; 1: k_sac_q_last's part; 2: k_sac_td_last's part
    const float* part; int nb, NC, B;     // kind 1: [3][nb] (log_p + target, log_p, qmin); kind 2: [NC][nb] loss partials
    const float* log_alpha; float* out; float scale; int accumulate;
    int auto_alpha; float* log_alpha_rw; float* al_m; float* al_v; AdamScalars s;
    const unsigned* poison; unsigned long long* applied; unsigned long long step;
};
constexpr int SAC_TAIL_LDS = 4 * 32 * 33;   // floats: the tile workgroups' `red`
__device__ __forceinline__ void sac_tail(const SacTailArgs& t, float* lds)
{
    const int tid = threadIdx.x, nsum = t.kind == 1 ? 3 : t.NC;
    for (int k = tid; k < nsum * t.nb; k += blockDim.x) lds[k] = t.part[k];
    const float out_pre = t.accumulate ? t.out[0] : 0.f;
    float la = 0.f, am = 0.f, av = 0.f;
    if (t.kind == 1) { la = t.log_alpha[0]; if (t.auto_alpha) { am = t.al_m[0]; av = t.al_v[0]; } }
    __syncthreads();
    auto sum = [&](int i) { float v = 0.f; for (int k = 0; k < t.nb; ++k) v += lds[i * t.nb + k]; return v; };   // sum_partials' order
    if (t.kind == 1) {
        float log_alpha = la;
        if (t.auto_alpha) {
            const float g = -(sum(0) / (float)t.B);
            const float mm = am * t.s.b1 + g * t.s.omb1;
            const float vv = av * t.s.b2 + t.s.omb2 * g * g;
            const float denom = __fsqrt_rn(vv) / t.s.sqrt_bc2 + t.s.eps;
            log_alpha = log_alpha + t.s.neg_step * mm / denom;
            if (tid == 0 && !(t.poison && *t.poison)) { t.log_alpha_rw[0] = log_alpha; t.al_m[0] = mm; t.al_v[0] = vv; if (t.applied) *t.applied = t.step; }
        }
        const float alpha = expf(log_alpha);
        const float s_logp = sum(1), s_qm = sum(2);
        if (tid == 0) t.out[0] = actor_loss_sum(out_pre, alpha, s_logp, s_qm, t.scale);
    } else {
        float total = out_pre;
        for (int i = 0; i < t.NC; ++i) total = add_scaled(total, sum(i), t.scale);
        if (tid == 0) t.out[0] = total;
    }
}
// k_dense_small<true> + the tail: grid.x = tiles + 1, the extra workgroup (of instance 0) runs sac_tail
__global__ __launch_bounds__(256) void k_dense_small_dx_tail(DenseArgsZ dz, SacTailArgs t)
{
    __shared__ float red[4][32][33];
    if (blockIdx.x + 1 == gridDim.x) {
        if (blockIdx.z == 0) sac_tail(t, &red[0][0][0]);
        return;
    }
    dense_small_body<true>(dz.a[blockIdx.z], red);
}

// ------------------------------------------------------------------------------------------------------------------------
// The actor's trunk (dense_chain.hpp, one h1 tile per workgroup) AND k_sac_heads_action in one launch: the eight workgroups of a row block store their
// h1 tiles with agent scope and take a ticket; the last one to arrive forms both heads' tiles for the 32 rows (k_sac_heads_action's tiles: same k-slices,
// MFMA order and sum; h1 comes back through the coherent level - the other seven may sit on other XCDs) and does the rows' action / log-probability
// arithmetic (sac_heads_row).  One launch less on the main queue's chain per update (the critic phase's actor pass) and one on the side queue - and
// 3 % SLOWER in the step (7 645 against 7 880 opt-steps/s): eight agent-scope tile stores acknowledged, a ticket and an uncached reload of h1 cost more than the
// launch they replace (as round 3 found for the weight gradients).  Same bits; kept behind BDR_SAC_HEADS_FUSE=1 with its test, not the default.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_sac_pi_chain_heads(Chain2Args a, SacHeadsActionArgs p, unsigned* tickets)
{
    __shared__ __attribute__((aligned(16))) float hs[32 * C2_LD];
    __shared__ float red[4][32][33];
    __shared__ unsigned s_last;
    static_assert(32 * C2_LD >= 4 * 32 * 33, "the second head's slices reuse the h0 block");
    dense_chain2_body<1, 2, true>(a, hs, red);
    const int NCG = a.n1 / 32, rb = (int)blockIdx.x / NCG, m0 = rb * 32;
    if (!last_workgroup(tickets + rb, (unsigned)NCG, &s_last)) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
    float (*red_s)[32][33] = reinterpret_cast<float (*)[32][33]>(hs);
    // operands: the lane's row of h1, k-slice `wave` (sc1: served by the coherent level), both heads' weights of that slice, the rows' noise and the biases
    const float* arow = p.h + (size_t)min(m0 + i, p.B - 1) * p.ldh + wave * (C2_N0 / 4) + 4 * h;
    f32x4 av[8], bm[8], bs[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float* pa = arow + 8 * j; asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(av[j]) : "v"(pa) : "memory"); }
    dense_small_load_b<false>(p.hm.w, p.w_ld, 0, C2_N0, wave, lane, bm);
    dense_small_load_b<false>(p.hs.w, p.w_ld, 0, C2_N0, wave, lane, bs);
    const float bias_m = p.hm.bias[i], bias_s = p.hs.bias[i];
    float zrow[8];
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) zrow[rr] = lane < p.A ? p.z[(size_t)min(m0 + wave * 8 + rr, p.B - 1) * p.A + lane] : 0.f;
    // (the wait carries the asm loads' destinations as in/out operands: to the compiler an asm's output is there when the statement is)
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(av[0]), "+v"(av[1]), "+v"(av[2]), "+v"(av[3]), "+v"(av[4]), "+v"(av[5]), "+v"(av[6]), "+v"(av[7]) :: "memory");
    {
        const f32x16 am = chain_mfma<8>(av, bm), as = chain_mfma<8>(av, bs);   // dense_small_tile's slots (c * 2 + u) and order
#pragma unroll
        for (int r = 0; r < 16; ++r) { red[wave][(r & 3) + 8 * (r >> 2) + 4 * h][i] = am[r]; red_s[wave][(r & 3) + 8 * (r >> 2) + 4 * h][i] = as[r]; }
    }
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
        const int r = wave * 8 + rr, row = m0 + r;
        if (row >= p.B) break;
        sac_heads_row(p, red, red_s, r, row, lane, bias_m, bias_s, zrow[rr]);
    }
}
}  // namespace
