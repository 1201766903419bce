// Two dense layers of an Mlp in ONE launch for the launch-bound agents (SAC: mlp/base.rs:13-41 as called by sac/base.rs:73-105):
//   h0 = act(x W0 + b0),  h1 = act(h0 W1 + b1)      x [M][k0], W0 [k0][n0], W1 [n0][n1], up to four networks per launch (blockIdx.z)
// The rows of an Mlp are independent between its layers: a workgroup takes 32 batch rows, forms ALL of h0 for them (k0 = 64, the padded
// observation width, n0 = 256: eight 32 x 32 tiles of 8 MFMAs per k-slice, two slices when the input has <= 32 columns), keeps it in LDS and forms its share of h1 from there.  No
// workgroup waits for another one, nothing depends on placement; what goes away is one kernel boundary (>= 2.4 us) and the round trip
// of h0 through memory between two dependent launches of ~8-12 us each.
//   Every 32 x 32 tile is formed with the k-slices, the MFMA order and the four-way sum of dense_small_tile / dense_small_sum
//   (dense.hpp): same bits as the layer-by-layer launches of k_dense_small, whatever the path (tests/test_gpu_sac.py).
//   TPW = 4: the workgroup owns four h1 tiles, one per wave; a wave runs the four k-slices one after the other into registers and adds
//            them in the fixed order - no LDS round trip, 128 MFMAs back to back (1024 x 256 x 256 x 4 networks: 256 workgroups).
//   TPW = 1: the workgroup owns one h1 tile, wave w takes k-slice w, the slices meet in LDS (k_dense_small's form) - 8 x the
//            workgroups for a single network, each recomputing h0 (32 MFMAs per wave once the padding slices are left out) for 32 more.
// h0 is stored once per row block (the column groups share its rows): the backward reads it (ReLU mask, dW operand).
#pragma once

namespace bdr {

struct Chain2Net { const float* x; int ldx; const float* w0; const float* b0; const float* w1; const float* b1; float* h0; float* h1; };
#ifdef C2_STAMPS   // tools/probes/chain2_probe.hip only: shader-clock stamps of workgroup 0, wave 0
#define C2_STAMP(k) do { if (a.stamps && blockIdx.x == 0 && blockIdx.z == 0 && threadIdx.x == 0) a.stamps[k] = clock64(); } while (0)
#else
#define C2_STAMP(k) do { } while (0)
#endif
struct Chain2Args {
    Chain2Net n[4];
    int M, n1, relu0, relu1;                  // layer 0: [C2_K0][C2_N0], layer 1: [C2_N0][n1]; n1 % (32 TPW) == 0
    unsigned* sig_flag; unsigned sig_epoch;   // optional: start_signal (igemm.hpp)
    // optional: the inputs x are written by another queue, which publishes wait_epoch in *wait_flag when they are complete (queue_flags.hpp).  The wait
    // is the kernel's first act - one cached load's round trip - instead of a one-wave k_flag_wait packet in front of the launch (+0.9 % of the SAC step, LAB.md 5)
    const unsigned* wait_flag; unsigned wait_epoch; unsigned long long wait_limit; unsigned* wait_err; unsigned wait_code;
#ifdef C2_STAMPS
    long long* stamps;
#endif
};

constexpr int C2_K0 = 64, C2_N0 = 256;   // the widths this kernel is built for (padded observation / hidden width of the SAC networks); others: two launches
constexpr int C2_LD = C2_N0 + 4;         // row stride (floats) of h0 in LDS: the 16 lanes of a ds_read_b128 phase land in 64 different banks

// B operand of k-slice s (of four over KRED) of a 32 x 32 tile: the lane's KRED / 32 k-quads as dense_small_tile<false> loads them (rows
// k = s KRED / 4 + 8 j + 4 h + e of the lane's column).  Buffer loads: ONE 32-bit lane offset (column + the lane half's four rows) for every load of
// the kernel and the row in the scalar offset - with flat loads the 64-bit addresses of ~170 loads in flight do not fit the register file beside them.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t chain_rsrc(const float* p, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, bytes, 0x00020000);
}
template <int KRED>
__device__ __forceinline__ void chain_load_b(__amdgpu_buffer_rsrc_t rs, int voff, int row_bytes, int s, f32x4 (&bv)[KRED / 32])
{
#pragma unroll
    for (int j = 0; j < KRED / 32; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e)
            bv[j][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, (s * (KRED / 4) + 8 * j + e) * row_bytes, 0));
}
// the slice's MFMAs in dense_small_tile's order (slot j = c * 2 + u, then q)
template <int NJ>
__device__ __forceinline__ f32x16 chain_mfma(const f32x4 (&av)[NJ], const f32x4 (&bv)[NJ])
{
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j][q], bv[j][q], acc, 0, 0, 0);
    return acc;
}

// "these loaded registers have landed": the compiler waits for them HERE (vmcnt), not at their first use behind a store - gfx9 counts loads and stores
// in one counter and acknowledges them out of order, so a first use behind a store is a wait for that store
__device__ __forceinline__ void chain_land(const f32x4 (&v)[8])
{
    asm volatile("" ::"v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]));
}

// KZ0: k-slices of layer 0 that hold input columns (2: logical width <= 32, the SAC observations / observation + action rows; else 4).  A slice of
// padding is +0 for every element (0 x 0 products): its MFMAs are left out and the sum keeps its "+ 0.f" so that a -0 partial sum becomes the same +0.
// The kernel's body.  hs: [32 * C2_LD] floats, red: [4][32][33] (TPW = 1 only).  AGENT_H1 (TPW = 1): h1 is stored with agent scope and the function returns in
// every thread - for a caller that hands the row block's h1 to the last of its workgroups (sac_fused.hpp k_sac_pi_chain_heads).
template <int TPW, int KZ0, bool AGENT_H1 = false>
__device__ __forceinline__ void dense_chain2_body(const Chain2Args& a, float* hs, float (*red)[32][33])
{
    static_assert(TPW == 1 || TPW == 4, "one tile per workgroup (a wave per k-slice) or four (a wave per tile)");
    static_assert(KZ0 == 2 || KZ0 == 4, "two or four k-slices of layer 0");
    static_assert(!AGENT_H1 || TPW == 1, "the hand-over form is the one-tile form");
    constexpr int J0 = C2_K0 / 32, J1 = C2_N0 / 32;   // k-quad slots per slice and lane half: 2 (layer 0), 8 (layer 1)
    start_signal(a.sig_flag, a.sig_epoch);
    C2_STAMP(0);
    if (a.wait_flag) {   // every workgroup, before its first load of x
        if (threadIdx.x == 0) {
            // first look with ordinary loads: the caches were invalidated when this kernel started and the flag only grows, so a cached value can be
            // too OLD, never too new - and 256 workgroups reading one word with agent scope (uncached, one channel) cost 6 us (measured)
            const unsigned f = *a.wait_flag, e = a.wait_err ? *a.wait_err : 0u;
            if ((int)(f - a.wait_epoch) < 0 && e == 0u) {   // not there yet: k_flag_wait's loop (time limit, poison word)
                const unsigned long long t0 = wall_clock64();
                while ((int)(__hip_atomic_load(a.wait_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - a.wait_epoch) < 0) {   // wrap-safe
                    __builtin_amdgcn_s_sleep(2);
                    if (a.wait_err && __hip_atomic_load(a.wait_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;   // another wait failed first
                    if (wall_clock64() - t0 > a.wait_limit) {
                        if (a.wait_err) __hip_atomic_store(a.wait_err, a.wait_code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                }
            }
        }
        // (no x line can be in a cache of this workgroup's XCD: the kernel's start invalidated them, and no workgroup reads x before its own wait)
        __syncthreads();
    }
    const Chain2Net& nt = a.n[blockIdx.z];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
    const int NCG = a.n1 / (32 * TPW);
    const int m0 = ((int)blockIdx.x / NCG) * 32, cg = (int)blockIdx.x % NCG;
    const float* arow = nt.x + (size_t)min(m0 + i, a.M - 1) * nt.ldx;   // rows >= M alias the last row (never stored)

    // ---- the operands come from memory in ONE stretch of loads, in the order of their use, layer 1's weights riding behind layer 0's MFMAs: the step
    //      is a chain of launches that are each a few dependent round trips long, and a wave has 64 loads in flight at most
    f32x4 xa[KZ0][J0];        // layer 0, A: the lane's row, slice s
    f32x4 wb[2][KZ0][J0];     // layer 0, B: tiles wave, wave + 4
    constexpr int S1 = TPW == 1 ? 1 : 4;
    f32x4 wl[S1][J1];         // layer 1, B: the wave's k-slice (TPW = 1) or all four slices of its tile (TPW = 4)
    const int n1_0 = TPW == 1 ? cg * 32 : (cg * 4 + wave) * 32;
    const __amdgpu_buffer_rsrc_t rs0 = chain_rsrc(nt.w0, C2_K0 * C2_N0 * 4), rs1 = chain_rsrc(nt.w1, (unsigned)(C2_N0 * a.n1 * 4));
    const int rb1 = a.n1 * 4, vo1 = 4 * h * rb1 + (n1_0 + i) * 4;   // lane offset into W1: the lane half's first row, the lane's column
#pragma unroll
    for (int s = 0; s < KZ0; ++s)
#pragma unroll
        for (int j = 0; j < J0; ++j) xa[s][j] = *reinterpret_cast<const f32x4*>(arow + s * (C2_K0 / 4) + 8 * j + 4 * h);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int s = 0; s < KZ0; ++s) chain_load_b<C2_K0>(rs0, (4 * h * C2_N0 + (wave + 4 * t) * 32 + i) * 4, C2_N0 * 4, s, wb[t][s]);
    const float b0v[2] = {nt.b0[wave * 32 + i], nt.b0[(wave + 4) * 32 + i]};
    chain_load_b<C2_N0>(rs1, vo1, rb1, TPW == 1 ? wave : 0, wl[0]);
    f32x4 e1 = {0.f, 0.f, 0.f, 0.f};
    float bias1 = 0.f;
    if constexpr (TPW == 1) e1 = *reinterpret_cast<const f32x4*>(nt.b1 + n1_0 + (tid & 7) * 4);
    else bias1 = nt.b1[n1_0 + i];
    __builtin_amdgcn_sched_barrier(0);   // (the scheduler would otherwise sink every load to its use: one round trip per slice)
    C2_STAMP(1);

    // ---- layer 0: every tile of the row block (wave w: tiles w, w + 4), the slices of a tile in registers, added in dense_small_sum's order -> LDS
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        if constexpr (TPW == 4) {   // the next slices of layer 1's weights, issued before this tile's MFMAs
            if (t == 0) chain_load_b<C2_N0>(rs1, vo1, rb1, 1, wl[1]);
            else { chain_load_b<C2_N0>(rs1, vo1, rb1, 2, wl[2]); chain_load_b<C2_N0>(rs1, vo1, rb1, 3, wl[3]); }
            __builtin_amdgcn_sched_barrier(0);
        }
        f32x16 v = chain_mfma<J0>(xa[0], wb[t][0]);
#pragma unroll
        for (int s = 1; s < KZ0; ++s) {
            const f32x16 p = chain_mfma<J0>(xa[s], wb[t][s]);
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = v[r] + p[r];
        }
#pragma unroll
        for (int s = KZ0; s < 4; ++s)
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = v[r] + 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float y = v[r] + b0v[t];
            if (a.relu0) y = y > 0.f ? y : 0.f;
            hs[((r & 3) + 8 * (r >> 2) + 4 * h) * C2_LD + (wave + 4 * t) * 32 + i] = y;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    C2_STAMP(2);
    __syncthreads();
    C2_STAMP(3);
    // h0 to memory (the backward reads it), in the shadow of layer 1's MFMAs: the column groups of a row block share it, group cg stores the rows cg,
    // cg + NCG, ...  Every loaded register has landed by now (they were issued a layer ago) and is declared so: the loads' counter is empty when the
    // stores enter it, and no MFMA below waits for a store.
#pragma unroll
    for (int s = 0; s < S1; ++s) chain_land(wl[s]);
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const int e = tid + 256 * p, row = cg + (e >> 6) * NCG, c = (e & 63) * 4;
        if (row < 32 && m0 + row < a.M) *reinterpret_cast<f32x4*>(nt.h0 + (size_t)(m0 + row) * C2_N0 + c) = *reinterpret_cast<const f32x4*>(&hs[row * C2_LD + c]);
    }
    // ---- layer 1 from LDS
    const float* hrow = &hs[i * C2_LD];
    f32x16 t1;
    if constexpr (TPW == 1) {
        f32x4 av[J1];
#pragma unroll
        for (int j = 0; j < J1; ++j) av[j] = *reinterpret_cast<const f32x4*>(hrow + wave * (C2_N0 / 4) + 8 * j + 4 * h);
        t1 = chain_mfma<J1>(av, wl[0]);
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wave][(r & 3) + 8 * (r >> 2) + 4 * h][i] = t1[r];
    } else {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            f32x4 av[J1];
#pragma unroll
            for (int j = 0; j < J1; ++j) av[j] = *reinterpret_cast<const f32x4*>(hrow + s * (C2_N0 / 4) + 8 * j + 4 * h);
            const f32x16 p = chain_mfma<J1>(av, wl[s]);
            if (s == 0) t1 = p;
            else {
#pragma unroll
                for (int r = 0; r < 16; ++r) t1[r] = t1[r] + p[r];
            }
        }
    }
    C2_STAMP(4);
    C2_STAMP(5);
    if constexpr (TPW == 1) {
        __syncthreads();
        const int r = tid >> 3, c4 = (tid & 7) * 4, m = m0 + r;
        if (m < a.M) {
            f32x4 v;
#pragma unroll
            for (int q = 0; q < 4; ++q) { v[q] = dense_small_sum(red, r, c4 + q) + e1[q]; if (a.relu1) v[q] = v[q] > 0.f ? v[q] : 0.f; }
            float* o = nt.h1 + (size_t)m * a.n1 + n1_0 + c4;
            if constexpr (AGENT_H1) {
#pragma unroll
                for (int q = 0; q < 4; ++q) st_agent(o + q, v[q]);   // read by the row block's last workgroup, possibly on another XCD
            } else *reinterpret_cast<f32x4*>(o) = v;
        }
    } else {
        float* o = nt.h1 + (size_t)(m0 + 4 * h) * a.n1 + n1_0 + i;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            t1[r] = t1[r] + bias1;
            if (a.relu1) t1[r] = t1[r] > 0.f ? t1[r] : 0.f;
        }
        if (m0 + 32 <= a.M) {   // (one test for the block: a predicate per store puts a vmcnt(0) between the stores)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[(size_t)((r & 3) + 8 * (r >> 2)) * a.n1] = t1[r];
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (m0 + 4 * h + (r & 3) + 8 * (r >> 2) < a.M) o[(size_t)((r & 3) + 8 * (r >> 2)) * a.n1] = t1[r];
        }
        C2_STAMP(6);
    }
}

template <int TPW, int KZ0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(TPW == 1 ? 2 : 1, TPW == 1 ? 2 : 1))) void k_dense_chain2(Chain2Args a)
{
    __shared__ __attribute__((aligned(16))) float hs[32 * C2_LD];
    __shared__ float red[TPW == 1 ? 4 : 1][32][33];
    dense_chain2_body<TPW, KZ0>(a, hs, red);
}

// can layers l0 -> l1 go through k_dense_chain2?
inline bool dense_chain2_ok(const DenseLayer& l0, const DenseLayer& l1)
{
    return l0.Kp == C2_K0 && l0.Np == C2_N0 && l1.Kp == l0.Np && l1.Np % 32 == 0;
}
// tpw: 0 = by the size of the launch, 1 / 4 = forced (4 needs l1.Np % 128 == 0)
struct ChainWait;
inline Chain2Args dense_chain2_args(const DenseLayer& l0, const DenseLayer& l1, int nz, const float* const* params_base, const DenseSrc* x, float* const* h0,
                                    float* const* h1, int M, unsigned* sig_flag, unsigned sig_epoch, const ChainWait* wait);
struct ChainWait { const unsigned* flag = nullptr; unsigned epoch = 0; unsigned long long limit = 0; unsigned* err = nullptr; unsigned code = 0; };
inline Chain2Args dense_chain2_args(const DenseLayer& l0, const DenseLayer& l1, int nz, const float* const* params_base, const DenseSrc* x, float* const* h0,
                                    float* const* h1, int M, unsigned* sig_flag, unsigned sig_epoch, const ChainWait* wait)
{
    Chain2Args c{};
    for (int z = 0; z < nz; ++z)
        c.n[z] = Chain2Net{x[z].p, x[z].ld, params_base[z] + l0.w, params_base[z] + l0.b, params_base[z] + l1.w, params_base[z] + l1.b, h0[z], h1[z]};
    c.M = M; c.n1 = l1.Np; c.relu0 = l0.relu; c.relu1 = l1.relu;
    c.sig_flag = sig_flag; c.sig_epoch = sig_epoch;
    if (wait && wait->flag) { c.wait_flag = wait->flag; c.wait_epoch = wait->epoch; c.wait_limit = wait->limit; c.wait_err = wait->err; c.wait_code = wait->code; }
    return c;
}
inline int32_t dense_chain2_z(hipStream_t st, const DenseLayer& l0, const DenseLayer& l1, int nz, const float* const* params_base, const DenseSrc* x,
                              float* const* h0, float* const* h1, int M, int tpw = 0, unsigned* sig_flag = nullptr, unsigned sig_epoch = 0,
                              const ChainWait* wait = nullptr)
{
    const Chain2Args c = dense_chain2_args(l0, l1, nz, params_base, x, h0, h1, M, sig_flag, sig_epoch, wait);
    const int rb = (M + 31) / 32;
    const bool can4 = l1.Np % 128 == 0;
    static const int min4 = [] { const char* e = getenv("BDR_CHAIN_MIN4"); return e ? atoi(e) : 192; }();   // (tuning switch)
    const bool four = tpw == 4 ? can4 : tpw == 1 ? false : (can4 && rb * (l1.Np / 128) * nz >= min4);
    const bool narrow = l0.in <= 32;   // input columns in the first two k-slices only
    const dim3 g4(rb * (l1.Np / 128), 1, nz), g1(rb * (l1.Np / 32), 1, nz);
    if (four) BDR_HIP(narrow ? step_launch(st, false, k_dense_chain2<4, 2>, g4, dim3(256), c) : step_launch(st, false, k_dense_chain2<4, 4>, g4, dim3(256), c));
    else BDR_HIP(narrow ? step_launch(st, false, k_dense_chain2<1, 2>, g1, dim3(256), c) : step_launch(st, false, k_dense_chain2<1, 4>, g1, dim3(256), c));
    return BDR_OK;
}

}   // namespace bdr
