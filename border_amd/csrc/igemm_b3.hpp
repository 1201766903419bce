// GEMM on the bf16 matrix cores with SPLIT f32 operands ("3 x bf16"), for the layers that ARE matrix-bound: IQN's
// [B * N quantiles][3136] x [3136][512] forward and its input gradient at config C4 (M = 32 768 rows, 0.74-0.79 of the FP32-MFMA
// peak with the exact kernel: csrc/iqn.hip).  The B = 256 conv layers of the DQN step are operand-delivery bound and gain nothing
// from it (tools/probes/bp_probe.hip, LAB.md 5); they stay on the FP32 MFMA.
//
// Every f32 operand x is split exactly into three bf16 terms x = t0 + t1 + t2 (8 + 8 + 8 significant bits, round-to-nearest
// split: split3_rn below), so a product a*b is the sum of nine exact bf16 x bf16 products.  TERMS = 9 keeps all of them: the MFMA then
// only rounds in its f32 accumulation, the error class of the FP32 MFMA / an fmaf chain.  TERMS = 6 drops the three
// products below 2^-24 |a||b| (mid*lo, lo*mid, lo*lo): measured 4e-6 relative on the C4 layer against the exact kernel
// (the parity bar is 1e-4 on quantile values; tests/test_gpu_iqn.py holds both kernels to it and to each other).  v_mfma_f32_32x32x16_bf16 retires 16 k in 32 cycles, the FP32
// v_mfma_f32_32x32x2_f32 2 k in 64: 9 terms cost 288 matrix cycles per 16 k instead of 512, 6 terms 192.
//
// Same contraction, policies and epilogue as k_igemm (igemm.hpp).  Differences:
//  * both operands are k-major in LDS as three bf16 planes ([rows][32 k], row stride 80 B: conflict-free ds_read_b128
//    fragments of 8 k);
//  * A (activations / gradients, f32 in HBM) is split by the staging threads on its way into LDS;
//  * B (weights) is read from pre-split bf16 planes in HBM (policy hook b_chunk), k-major: the forward wants [n][k] (the
//    transpose of W[k][n]), the input gradient W's own [k][n] rows.  The planes are re-split from the f32 weights at the start
//    of every update (k_split_planes: ~10 us against a 5 ms step), so no parameter writer has to keep them fresh;
//  * an A operand that is a Hadamard product (ADenseHad: IQN's merge m = relu(phi) * psi(x)[b]) is multiplied before the split.
#pragma once
#include "igemm.hpp"

namespace bdr {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));   // plain vector types keep the staging registers out of scratch
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));

constexpr int B3_LDR = 40;   // (probes: padded row stride of a bf16 plane tile in u16 units, 32 k + 8 pad = 80 B)
// k_igemm_b3's plane tiles: rows of 32 k = 64 B, unpadded, the four 16-byte chunks of row r stored at chunk ^ ((r >> 2) & 3).  A fragment
// read (16 lanes = 16 rows, one chunk index) then covers all 64 banks once, and the staging writes (4 rows x 64 B per 16 / 32 lanes)
// are 256 contiguous bytes.  The padded 80-byte rows before it were conflict-free for the reads only: rows r and r + 3 overlap modulo 256 B
// for the writes - SQ_LDS_BANK_CONFLICT was a third of SQ_LDS_IDX_ACTIVE in the C4 kernels.
constexpr int B3_ROW = 32;   // u16 per row
#ifndef BDR_B3_XCD_MAP
#define BDR_B3_XCD_MAP 1
#endif
constexpr bool b3_xcd_map = BDR_B3_XCD_MAP != 0;
__device__ __forceinline__ int b3_off(int row, int chunk) { return row * B3_ROW + ((chunk ^ ((row >> 2) & 3)) << 3); }

// ---- the operand split.  x = t0 + t1 + t2 EXACTLY, every term a bf16, every term the ROUND-TO-NEAREST-EVEN bf16 of what the terms before
// it left over (v_cvt_pk_bf16_f32).  Exactness: an f32 has 24 significant bits; after an 8-bit term rounded to nearest the residual is at most
// half an ulp of that term, so it spans <= 16 bit positions, the next residual <= 8, which the third term holds exactly; each subtraction
// is exact (the result is representable).  Rounds 4-5 split by TRUNCATION (mask the low 16 bits): also exact, but every residual then
// carries the sign of x, so the three products a 6-term kernel drops (t1*u2, t2*u1, t2*u2) all push the same way; with the nearest split the
// residuals are half the size and of either sign - the dropped part is ~10x smaller at the maximum and zero-mean (What's
// weak 2; tests/test_gpu_dqn.py states the resulting per-layer budget).  The nearest split is also the cheaper one here: one packed
// conversion per pair instead of two masks and a pack.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t bf16_pair_rn(float a, float b)   // {bf16(a) in bits 0-15, bf16(b) in bits 16-31}
{
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{a, b}, bf16x2_t));
}
// one float -> its three terms (bf16 bit patterns)
__device__ __forceinline__ void split3_rn(float x, uint16_t (&v)[3])
{
    const uint32_t h = bf16_pair_rn(x, 0.f) & 0xffffu;
    const float r1 = x - __uint_as_float(h << 16);               // exact
    const uint32_t m = bf16_pair_rn(r1, 0.f) & 0xffffu;
    const float r2 = r1 - __uint_as_float(m << 16);              // exact, <= 8 significant bits
    const uint32_t l = bf16_pair_rn(r2, 0.f) & 0xffffu;          // exact
    v[0] = (uint16_t)h; v[1] = (uint16_t)m; v[2] = (uint16_t)l;
}
// four floats -> packed bf16 pairs: p[plane] = {pair(x0,x1), pair(x2,x3)}
__device__ __forceinline__ void split3_f32x4(const f32x4& x, u32x2_t (&p)[3])
{
    float r[4] = {x[0], x[1], x[2], x[3]};
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
        const uint32_t a = bf16_pair_rn(r[0], r[1]), b = bf16_pair_rn(r[2], r[3]);
        p[pl] = u32x2_t{a, b};
        if (pl < 2) {
            r[0] -= __uint_as_float(a << 16); r[1] -= __uint_as_float(a & 0xffff0000u);
            r[2] -= __uint_as_float(b << 16); r[3] -= __uint_as_float(b & 0xffff0000u);
        }
    }
}

// weights -> three bf16 planes in BOTH orders from one read of the f32 matrix src[R][C] (row stride ld): nat[plane][R][C] and
// tr[plane][C][R] (either may be null).  32 x 32 tiles through LDS so that both writes are contiguous runs.
static __global__ __launch_bounds__(256) void k_split_planes2(const float* __restrict__ src, int ld, uint16_t* __restrict__ nat, uint16_t* __restrict__ tr, int R, int C)
{
    __shared__ uint16_t t[3][32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 8 rows per pass
    const size_t n = (size_t)R * C;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = r0 + ty + 8 * j, c = c0 + tx;
        const float x = (r < R && c < C) ? src[(size_t)r * ld + c] : 0.f;
        uint16_t v[3];
        split3_rn(x, v);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            t[pl][ty + 8 * j][tx] = v[pl];
            if (nat && r < R && c < C) nat[pl * n + (size_t)r * C + c] = v[pl];
        }
    }
    __syncthreads();
    if (!tr) return;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = c0 + ty + 8 * j, r = r0 + tx;   // transposed: consecutive threads walk r
        if (c < C && r < R)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) tr[pl * n + (size_t)c * R + r] = t[pl][tx][ty + 8 * j];
    }
}

// (probe form) weights -> three bf16 planes, optionally transposed: src [R][C] f32 -> dst[plane][R][C] (T = 0) or dst[plane][C][R] (T = 1)
static __global__ __launch_bounds__(256) void k_split_planes(const float* __restrict__ src, uint16_t* __restrict__ dst, int R, int C, int T)
{
    const size_t n = (size_t)R * C;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int r = (int)(i / C), c = (int)(i % C);
    const float x = src[i];
    uint16_t v[3];
    split3_rn(x, v);
    const size_t o = T ? (size_t)c * R + r : i;
    dst[o] = v[0]; dst[n + o] = v[1]; dst[2 * n + o] = v[2];
}

// P: as for k_igemm, plus
//   b_chunk(args, z, y, plane, kt, n, kq) -> const uint4*   the 8 bf16 (k = kt*32 + kq*8 ..) of B column n (global column index)
//   P::MAXW   (optional, default 2) waves per SIMD the kernel is compiled for: 1 = 512 VGPRs for 128 x 128 tiles, 2 = 256
// policies with a row-group epilogue declare `static constexpr bool GROUP_EPI = true` (+ group_scale / group_elem_store / group_store)
template <class P, class = void> struct b3_group_epi : std::false_type {};
template <class P> struct b3_group_epi<P, std::enable_if_t<P::GROUP_EPI>> : std::true_type {};

template <class P, class = void> struct b3_maxw { static constexpr int value = 2; };
template <class P> struct b3_maxw<P, std::void_t<decltype(P::MAXW)>> { static constexpr int value = P::MAXW; };
template <class P, class = void> struct b3_minw { static constexpr int value = 1; };   // P::MINW (optional): waves per SIMD the registers must leave room for
template <class P> struct b3_minw<P, std::void_t<decltype(P::MINW)>> { static constexpr int value = P::MINW; };
template <class P, int TERMS = 9>
__global__ __launch_bounds__(64 * P::WM * P::WN) __attribute__((amdgpu_waves_per_eu(b3_minw<P>::value, b3_maxw<P>::value))) void k_igemm_b3(typename P::Args args)
{
    using A = typename P::A;
    if constexpr (has_start_signal<typename P::Args>::value) start_signal(args.sig_flag, args.sig_epoch);   // (igemm.hpp: cross-queue progress flag)
    static_assert(A::VEC == 4, "f32 A operands");
    static_assert(TERMS == 6 || TERMS == 9, "6 or 9 partial products");
    constexpr int NW = P::WM * P::WN, NT = 64 * NW;
    constexpr int BM = P::WM * P::TM * 32, BN = P::WN * P::TN * 32;
    constexpr int APR = BK / 4;                               // f32x4 loads per A row
    static_assert(NT % APR == 0, "a thread keeps its k-quad across passes");
    constexpr int A_ELEMS = BM * APR, A_PASSES = (A_ELEMS + NT - 1) / NT;
    constexpr int B_CH = BN * 4, B_PASSES = (B_CH + NT - 1) / NT;   // 16-byte chunks per plane
    constexpr int PLANE_A = BM * B3_ROW, PLANE_B = BN * B3_ROW;      // u16
    constexpr int STAGE = 3 * (PLANE_A + PLANE_B);                   // u16
    __shared__ __attribute__((aligned(16))) uint16_t smem[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / P::WN, wn = wave % P::WN;
    const int NC = P::N(args), NT_N = (NC + BN - 1) / BN;   // a ragged last column tile computes clamped columns and stores none of them
    // the NT_N column tiles of a row tile read the same A rows: keep them on one XCD (workgroup b runs on XCD b % 8), back to back in that
    // XCD's dispatch order, so that its L2 serves all but the first read (linear order: four XCDs each fetch the rows from HBM / MALL)
    int mt, nt;
    {
        const int MT = (int)gridDim.x / NT_N;
        if ((MT & 7) == 0 && b3_xcd_map) { const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3; mt = (q / NT_N) * 8 + xcd; nt = q % NT_N; }
        else { mt = blockIdx.x / NT_N; nt = blockIdx.x % NT_N; }
    }
    const int m0 = mt * BM, n0 = nt * BN;
    const int z = blockIdx.z, y = blockIdx.y;
    const int M = P::M(args);

    const int a_q = tid % APR, a_r = tid / APR;
    typename A::Row rows[A_PASSES];
#pragma unroll
    for (int p = 0; p < A_PASSES; ++p) {
        int mr;
        const bool ok = vrow_of<P>(args, y, m0 + p * (NT / APR) + a_r, mr);   // (position-class policies: the row map depends on blockIdx.y)
        rows[p] = A::row(P::a_src(args, z), ok ? mr : M, M);
    }
    int kt0, kt1;
    P::kt_range(args, y, kt0, kt1);
    const int nkt = kt1 - kt0;
    auto tile = [&](int it) { return kt0 + min(it, nkt - 1); };

    f32x4 ra[2][A_PASSES];
    constexpr bool HAD = a_has_had<A>::value;
    f32x4 rh[2][HAD ? A_PASSES : 1];   // second factor of a Hadamard A operand
    u32x4_t rb[2][B_PASSES][3];
    // The staging work of a k-tile is cut into per-pass slices (one 16-byte A chunk or one 3-plane B chunk per thread each) so that
    // the k loop can place them BETWEEN groups of MFMAs: a wave issues in order, and a block of 24 back-to-back MFMAs keeps it stalled
    // on the matrix pipe for ~700 cycles before the split's VALU work behind it could start (first version: 36 % of the bf16 peak).
    auto prefetch_a = [&](auto set, int kt, int p) {
        constexpr int S = decltype(set)::value;
        if (A_ELEMS % NT == 0 || tid + p * NT < A_ELEMS) {
            A::load(rows[p], kt, a_q, &ra[S][p]);
            if constexpr (HAD) rh[S][p] = A::load_had(rows[p], kt, a_q);
        }
    };
    auto prefetch_b = [&](auto set, int kt, int p) {
        constexpr int S = decltype(set)::value;
        const int e = tid + p * NT;
        if (B_CH % NT != 0 && e >= B_CH) return;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) rb[S][p][pl] = *reinterpret_cast<const u32x4_t*>(P::b_chunk(args, z, y, pl, kt, min(n0 + e / 4, NC - 1), e % 4));
    };
    auto commit_a = [&](auto set, int stage, int p) {
        constexpr int S = decltype(set)::value;
        uint16_t* As = smem + stage * STAGE;
        if (A_ELEMS % NT != 0 && tid + p * NT >= A_ELEMS) return;
        u32x2_t sp[3];
        f32x4 av = ra[S][p];
        if constexpr (HAD) av *= rh[S][p];
        split3_f32x4(av, sp);
        const int o = b3_off(p * (NT / APR) + a_r, a_q >> 1) + (a_q & 1) * 4;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<u32x2_t*>(&As[pl * PLANE_A + o]) = sp[pl];
    };
    auto commit_b = [&](auto set, int stage, int p) {
        constexpr int S = decltype(set)::value;
        uint16_t* Bs = smem + stage * STAGE + 3 * PLANE_A;
        const int e = tid + p * NT;
        if (B_CH % NT != 0 && e >= B_CH) return;
        const int o = b3_off(e / 4, e % 4);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<u32x4_t*>(&Bs[pl * PLANE_B + o]) = rb[S][p][pl];
    };
    auto prefetch = [&](auto set, int kt) {
#pragma unroll
        for (int p = 0; p < A_PASSES; ++p) prefetch_a(set, kt, p);
#pragma unroll
        for (int p = 0; p < B_PASSES; ++p) prefetch_b(set, kt, p);
    };
    auto commit = [&](auto set, int stage) {
#pragma unroll
        for (int p = 0; p < A_PASSES; ++p) commit_a(set, stage, p);
#pragma unroll
        for (int p = 0; p < B_PASSES; ++p) commit_b(set, stage, p);
    };
    using Set0 = std::integral_constant<int, 0>;
    using Set1 = std::integral_constant<int, 1>;

    f32x16 acc[P::TM][P::TN];
#pragma unroll
    for (int tm = 0; tm < P::TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < P::TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

    if (nkt > 0) {
        prefetch(Set0{}, tile(0));
        prefetch(Set1{}, tile(1));
        commit(Set0{}, 0);
        prefetch(Set0{}, tile(2));
    }

    const int j = lane & 31, h = lane >> 5;
    typename P::Epi epi = P::epi(args, z, y);
    int mrow[P::TM][16];
    unsigned okmask[P::TM];
    float aux[P::TM][P::TN][16];
#pragma unroll
    for (int tm = 0; tm < P::TM; ++tm) {
        okmask[tm] = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int mv = m0 + (wm * P::TM + tm) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (vrow_of<P>(args, y, mv, mrow[tm][r])) okmask[tm] |= 1u << r;
            else mrow[tm][r] = 0;
#pragma unroll
            for (int tn = 0; tn < P::TN; ++tn)
                aux[tm][tn][r] = P::epi_load(epi, mrow[tm][r], min(n0 + (wn * P::TN + tn) * 32 + j, NC - 1));
        }
    }
    __syncthreads();

    int cur = 0;
    // Slices of a k-tile's staging work: the commit of tile it+1 (B passes, then A passes - each A pass followed at once by its load of
    // tile it+3 into the registers it has just freed) into the idle stage rides on the MFMA groups of the first k-step, the B loads of
    // tile it+3 on those of the second; one
    // barrier per k-tile, BETWEEN the two k-steps: by then every wave has committed tile it+1 and has read all of tile it (the
    // fragments of a k-step are fetched from LDS one k-step ahead), so the second k-step can already fetch tile it+1's first fragments
    // and the next step may overwrite this stage.
    constexpr int C_SLICES = A_PASSES + B_PASSES, PER = (C_SLICES + TERMS - 1) / TERMS;
    bf16x8_t fa[2][P::TM][3], fb[2][P::TN][3];
    auto load_frag = [&](auto buf, int stage, int s) {
        constexpr int F = decltype(buf)::value;
        const uint16_t* As = smem + stage * STAGE;
        const uint16_t* Bs = As + 3 * PLANE_A;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
            for (int tm = 0; tm < P::TM; ++tm)
                fa[F][tm][pl] = *reinterpret_cast<const bf16x8_t*>(&As[pl * PLANE_A + b3_off((wm * P::TM + tm) * 32 + j, s * 2 + h)]);
#pragma unroll
            for (int tn = 0; tn < P::TN; ++tn)
                fb[F][tn][pl] = *reinterpret_cast<const bf16x8_t*>(&Bs[pl * PLANE_B + b3_off((wn * P::TN + tn) * 32 + j, s * 2 + h)]);
        }
    };
    // partial products, smallest first: (a plane, b plane)
    auto mfma_group = [&](auto buf, int t) {
        constexpr int F = decltype(buf)::value;
        constexpr int ORD9[9][2] = {{2, 2}, {2, 1}, {1, 2}, {2, 0}, {0, 2}, {1, 1}, {1, 0}, {0, 1}, {0, 0}};
#pragma unroll
        for (int tm = 0; tm < P::TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < P::TN; ++tn)
                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[F][tm][ORD9[t][0]], fb[F][tn][ORD9[t][1]], acc[tm][tn], 0, 0, 0);
    };
    using Buf0 = std::integral_constant<int, 0>;
    using Buf1 = std::integral_constant<int, 1>;
    if (nkt > 0) load_frag(Buf0{}, 0, 0);   // (stage 0 was committed before the prologue's barrier)
    auto step = [&](auto set, int it) {   // set holds tile it+1; refilled with tile it+3
        const int k3 = tile(it + 3);
        load_frag(Buf1{}, cur, 1);
#pragma unroll
        for (int t = 9 - TERMS; t < 9; ++t) {
            mfma_group(Buf0{}, t);
            const int g = t - (9 - TERMS);
#pragma unroll
            for (int u = 0; u < PER; ++u) {
                const int q = g * PER + u;   // (compile-time after unrolling)
                // the B chunks (weight planes: L2 hits) are committed first, the A rows (streamed from HBM, their loads issued first) last:
                // the loads that take longest get the longest time to arrive (+1.9 % on the C4 step)
                if (q < B_PASSES) commit_b(set, cur ^ 1, q);
                else if (q < C_SLICES) {
                    commit_a(set, cur ^ 1, q - B_PASSES);
                    prefetch_a(set, k3, q - B_PASSES);   // the registers are free again: tile it+3's rows are requested half a step earlier (+1.3 %)
                }
            }
            // within the group: one MFMA, then a run of the slice's VALU work, ... (a wave issues in order: four MFMAs back to back stall it
            // on the matrix pipe and the split behind them starts only when the last one has issued; +1 % on the C4 step)
#pragma unroll
            for (int i = 0; i < P::TM * P::TN; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 12, 0);
            }
            __builtin_amdgcn_sched_barrier(0);   // keep the slices between the MFMA groups (hipcc otherwise regroups them behind the block)
        }
        __syncthreads();
        load_frag(Buf0{}, cur ^ 1, 0);
#pragma unroll
        for (int t = 9 - TERMS; t < 9; ++t) {
            mfma_group(Buf1{}, t);
            const int g = t - (9 - TERMS);
#pragma unroll
            for (int u = 0; u < PER; ++u) {
                const int q = g * PER + u;
                if (q < B_PASSES) prefetch_b(set, k3, q);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        cur ^= 1;
    };
    // (no conditional second step inside the loop: with it, the path "first step, then straight back to the header" makes the set the
    //  header consumes the most recently loaded one as far as the compiler's s_waitcnt insertion can tell, and every iteration drained
    //  the loads issued half a step earlier - vmcnt(12 .. 0) at the header instead of vmcnt(28 .. 16))
    int it = 0;
    for (; it + 1 < nkt; it += 2) {
        step(Set1{}, it);
        step(Set0{}, it + 1);
    }
    if (it < nkt) step(Set1{}, it);

    if constexpr (b3_group_epi<P>::value) {
        // Row-group epilogue: the wave's TM * 32 consecutive rows are ONE group (host-checked: group size = TM * 32, M a multiple of the
        // tile); besides the element-wise store, a policy sums v * aux over the group's rows per column and stores one value per
        // (group, column) - IQN's merge backward (d psi[b] = sum_n dm[b, n] phi[b, n]) without a second pass over dm.
        const int g = (m0 + wm * P::TM * 32) / (P::TM * 32);
#pragma unroll
        for (int tn = 0; tn < P::TN; ++tn) {
            const int n = n0 + (wn * P::TN + tn) * 32 + j;
            if (n >= NC) continue;
            const float gs = P::group_scale(epi, g, n);
            float s = 0.f;
#pragma unroll
            for (int tm = 0; tm < P::TM; ++tm)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    s = fmaf(acc[tm][tn][r], aux[tm][tn][r], s);
                    P::group_elem_store(epi, mrow[tm][r], n, acc[tm][tn][r], aux[tm][tn][r], gs);
                }
            s += __shfl_xor(s, 32);
            if (h == 0) P::group_store(epi, g, n, s, gs);
        }
    } else {
#pragma unroll
    for (int tm = 0; tm < P::TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < P::TN; ++tn) {
            const int n = n0 + (wn * P::TN + tn) * 32 + j;
#pragma unroll
            for (int r = 0; r < 16; ++r) asm volatile("" ::"v"(aux[tm][tn][r]));
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if ((okmask[tm] >> r & 1) && n < NC) P::store(epi, mrow[tm][r], n, acc[tm][tn][r], aux[tm][tn][r]);
        }
    }
}

// stop: event completed by this kernel's own dispatch packet (as launch_igemm)
template <class P, int TERMS = 9>
inline hipError_t launch_igemm_b3(hipStream_t st, dim3 grid, const typename P::Args& args, hipEvent_t stop = nullptr)
{
    hipExtLaunchKernelGGL((k_igemm_b3<P, TERMS>), grid, dim3(64 * P::WM * P::WN), 0, st, nullptr, stop, 0, args);
    return hipGetLastError();
}

}  // namespace bdr
