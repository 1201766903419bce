// Probe only (round 4): the weight gradient of IQN's merge layer with BOTH operands split into three bf16 terms (reduction over the rows).
// Correct (tests/test_gpu_iqn.py passed with it wired in) but SLOWER than the FP32-MFMA kernel at C4: 1.40 ms vs 1.15 ms - two operands to
// split per k-tile is as much VALU work (about 1 300 cycles per k-tile and wave) as the six bf16 MFMAs per product save on the matrix
// pipe (1 536 cycles), and the transposed staging needs 36 dword loads per thread and k-tile.  The product keeps the FP32 kernel for this GEMM.
// Include after border_amd/csrc/igemm_b3.hpp (namespace bdr).
#pragma once
#include "igemm_b3.hpp"

namespace bdr {

// ------------------------------------------------------------------------------------------------
// k_igemm_red_b3: G[ko][n] = sum_{m in chunk} X(m,ko) * Y(m,n)  (a weight gradient: both operands are f32 activations / gradients,
// the contraction runs over the ROWS m) on the bf16 matrix cores with both operands split into three bf16 terms.
//   The MFMA wants, per lane, 8 consecutive contraction elements of its row: the LDS tiles are [ko][m] and [n][m] (m contiguous:
//   exactly the plane layout of k_igemm_b3), so the staging transposes.  A thread owns one column (ko, or n) and four quads of
//   four consecutive rows m: four dword loads per quad (a wave reads 256 contiguous bytes per row), the optional Hadamard factor
//   once per quad (had[m / had_group][ko]; had_group % 4 == 0), one exact 3-way split, one ds_write_b64 per plane.
//   Tile 128 x 128 (2 x 2 waves, 64 x 64 each), 32 rows of m per k-tile; per k-tile 16 staging slices (8 commits, 8 prefetches) ride
//   between the 12 MFMA groups like in k_igemm_b3.  grid: (ko tiles * n tiles) * chunks workgroups, 1-D, chunk c on XCD c % 8
//   (chunks % 8 == 0): all tiles of a row chunk read the same X / Y rows.
//   part[chunk][Kp * Np + Np]: the tile's sums, and the column sums of Y (bias gradient) from the ko-tile-0 workgroups.
// ------------------------------------------------------------------------------------------------
struct RedB3Args {
    const float* x; int x_ld;                       // [M][Kp]
    const float* had; int had_ld, had_group;        // optional second factor of X: x[m][k] * had[m / had_group][k] (nullptr: none)
    const float* y;                                  // [M][Np]
    float* part; size_t part_stride;                 // [chunks][Kp * Np + Np]
    int M, Kp, Np, chunks;
};
template <int TERMS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_igemm_red_b3(RedB3Args a)
{
    static_assert(TERMS == 6 || TERMS == 9, "6 or 9 partial products");
    constexpr int TM = 2, TN = 2, BMK = 128, BN = 128;               // ko rows x n columns of the tile
    constexpr int PLANE = 128 * B3_LDR;                               // u16 per plane and operand
    constexpr int STAGE = 6 * PLANE;
    __shared__ __attribute__((aligned(16))) uint16_t smem[2 * STAGE];   // 120 KB
    __shared__ float sb[2][128];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int KT = (a.Kp + BMK - 1) / BMK, NT = a.Np / BN, TILES = KT * NT;
    int tile, chunk;
    if ((a.chunks & 7) == 0) { const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3; chunk = (q / TILES) * 8 + xcd; tile = q % TILES; }
    else { chunk = blockIdx.x / TILES; tile = blockIdx.x % TILES; }
    const int kot = tile / NT, ko0 = kot * BMK, n0 = (tile % NT) * BN;
    const int n_mt = (a.M + 31) / 32, per = (n_mt + a.chunks - 1) / a.chunks;
    const int mt0 = chunk * per, mt1 = min(n_mt, mt0 + per), nkt = max(mt1 - mt0, 0);
    auto tile_m = [&](int it) { return (mt0 + min(it, max(nkt - 1, 0))) * 32; };

    // staging roles: column c = tid % 128 of the X tile (ko0 + c) and of the Y tile (n0 + c); quads mq, mq + 2, mq + 4, mq + 6 (mq = tid / 128)
    const int c = tid & 127, mq = tid >> 7;
    const bool x_ok = ko0 + c < a.Kp;
    const float* xc = a.x + (x_ok ? ko0 + c : 0);
    const float* hc = a.had ? a.had + (x_ok ? ko0 + c : 0) : nullptr;
    const float* yc = a.y + n0 + c;
    f32x4 rx[2][4], ry[2][4];
    float rh[2][4];
    float bsum = 0.f;
    const bool count_bias = kot == 0;
    auto prefetch_x = [&](auto set, int m0, int q) {
        constexpr int S = decltype(set)::value;
        const int m = m0 + 4 * (mq + 2 * q);
#pragma unroll
        for (int j = 0; j < 4; ++j) rx[S][q][j] = (x_ok && m + j < a.M) ? xc[(size_t)(m + j) * a.x_ld] : 0.f;
        if (hc) rh[S][q] = m < a.M ? hc[(size_t)(m / a.had_group) * a.had_ld] : 0.f;
    };
    auto prefetch_y = [&](auto set, int m0, int q) {
        constexpr int S = decltype(set)::value;
        const int m = m0 + 4 * (mq + 2 * q);
#pragma unroll
        for (int j = 0; j < 4; ++j) ry[S][q][j] = m + j < a.M ? yc[(size_t)(m + j) * a.Np] : 0.f;
    };
    auto commit_x = [&](auto set, int stage, int q) {
        constexpr int S = decltype(set)::value;
        f32x4 v = rx[S][q];
        if (hc) v *= rh[S][q];
        u32x2_t sp[3];
        split3_f32x4(v, sp);
        uint16_t* Xs = smem + stage * STAGE;
        const int o = c * B3_LDR + 4 * (mq + 2 * q);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<u32x2_t*>(&Xs[pl * PLANE + o]) = sp[pl];
    };
    auto commit_y = [&](auto set, int stage, int q, bool fresh) {
        constexpr int S = decltype(set)::value;
        const f32x4 v = ry[S][q];
        if (count_bias && fresh) bsum += (v[0] + v[1]) + (v[2] + v[3]);
        u32x2_t sp[3];
        split3_f32x4(v, sp);
        uint16_t* Ys = smem + stage * STAGE + 3 * PLANE;
        const int o = c * B3_LDR + 4 * (mq + 2 * q);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<u32x2_t*>(&Ys[pl * PLANE + o]) = sp[pl];
    };
    using Set0 = std::integral_constant<int, 0>;
    using Set1 = std::integral_constant<int, 1>;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

    if (nkt > 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { prefetch_x(Set0{}, tile_m(0), q); prefetch_y(Set0{}, tile_m(0), q); }
#pragma unroll
        for (int q = 0; q < 4; ++q) { prefetch_x(Set1{}, tile_m(1), q); prefetch_y(Set1{}, tile_m(1), q); }
#pragma unroll
        for (int q = 0; q < 4; ++q) { commit_x(Set0{}, 0, q); commit_y(Set0{}, 0, q, true); }
#pragma unroll
        for (int q = 0; q < 4; ++q) { prefetch_x(Set0{}, tile_m(2), q); prefetch_y(Set0{}, tile_m(2), q); }
    }
    __syncthreads();

    const int j = lane & 31, h = lane >> 5;
    bf16x8_t fa[2][TM][3], fb[2][TN][3];
    auto load_frag = [&](auto buf, int stage, int s) {
        constexpr int F = decltype(buf)::value;
        const uint16_t* Xs = smem + stage * STAGE;
        const uint16_t* Ys = Xs + 3 * PLANE;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
                fa[F][tm][pl] = *reinterpret_cast<const bf16x8_t*>(&Xs[pl * PLANE + ((wm * TM + tm) * 32 + j) * B3_LDR + s * 16 + h * 8]);
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
                fb[F][tn][pl] = *reinterpret_cast<const bf16x8_t*>(&Ys[pl * PLANE + ((wn * TN + tn) * 32 + j) * B3_LDR + s * 16 + h * 8]);
        }
    };
    auto mfma_group = [&](auto buf, int t) {
        constexpr int F = decltype(buf)::value;
        constexpr int ORD9[9][2] = {{2, 2}, {2, 1}, {1, 2}, {2, 0}, {0, 2}, {1, 1}, {1, 0}, {0, 1}, {0, 0}};
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[F][tm][ORD9[t][0]], fb[F][tn][ORD9[t][1]], acc[tm][tn], 0, 0, 0);
    };
    using Buf0 = std::integral_constant<int, 0>;
    using Buf1 = std::integral_constant<int, 1>;
    if (nkt > 0) load_frag(Buf0{}, 0, 0);
    int cur = 0;
    constexpr int PER = (8 + TERMS - 1) / TERMS;   // 8 slices per k-step over TERMS groups
    auto step = [&](auto set, int it) {   // set holds tile it+1; refilled with tile it+3
        const int m3 = tile_m(it + 3);
        const bool fresh = it + 1 < nkt;   // the clamped tail re-stages the last tile: not counted twice in the bias sums
        load_frag(Buf1{}, cur, 1);
#pragma unroll
        for (int t = 9 - TERMS; t < 9; ++t) {
            mfma_group(Buf0{}, t);
            const int g = t - (9 - TERMS);
#pragma unroll
            for (int u = 0; u < PER; ++u) {
                const int q = g * PER + u;
                if (q < 4) commit_x(set, cur ^ 1, q);
                else if (q < 8) commit_y(set, cur ^ 1, q - 4, fresh);
            }
#pragma unroll
            for (int i = 0; i < TM * TN; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 20, 0); }
            __builtin_amdgcn_sched_barrier(0);
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
                if (q < 4) prefetch_x(set, m3, q);
                else if (q < 8) prefetch_y(set, m3, q - 4);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        cur ^= 1;
    };
    for (int it = 0; it < nkt; it += 2) {
        step(Set1{}, it);
        if (it + 1 < nkt) step(Set0{}, it + 1);
    }

    float* part = a.part + (size_t)chunk * a.part_stride;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ko = ko0 + (wm * TM + tm) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const int n = n0 + (wn * TN + tn) * 32 + j;
                if (ko < a.Kp) part[(size_t)ko * a.Np + n] = acc[tm][tn][r];
            }
    if (count_bias) {   // column sums of Y over this chunk's rows: the two quad groups of a column, in a fixed order
        sb[mq][c] = bsum;
        __syncthreads();
        if (tid < 128) part[(size_t)a.Kp * a.Np + n0 + tid] = sb[0][tid] + sb[1][tid];
    }
}

}  // namespace bdr
