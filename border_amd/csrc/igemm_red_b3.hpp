// Weight gradient of a dense layer on the bf16 matrix cores with split operands (six of the nine exact bf16 partial products, as
// igemm_b3.hpp): G[ko][n] = sum_m X(m, ko) Y(m, n), the contraction running over the ROWS m.  gfx950 only.
//   X (an activation, optionally times a per-row-group Hadamard factor: IQN's merged feature m = psi[b] * phi) is f32 in HBM and is
//   split in the kernel; Y (the output gradient) comes as three bf16 planes ytr[plane][n][m] (m contiguous) that k_split_rows_tr
//   writes ONCE per update - every one of the Kp / 128 row tiles of G would otherwise split the same Y rows again.  Two round-4 probes
//   (tools/probes/igemm_red_b3*.hpp) that staged X column by column (16 scalar loads per thread and k-tile) only matched the FP32 kernel;
//   here a thread loads eight ROWS of two adjacent columns (wave-coalesced 512-byte row segments) and the transposition is just the
//   choice of which eight values go into one 16-byte LDS chunk.
#pragma once
#include "igemm_b3.hpp"

namespace bdr {

// y[M][C] (row stride ld) -> tr[plane][C][M] bf16 planes, M % 64 == 0, C % 64 == 0: 64 x 64 tiles through LDS, one 16-byte store per
// plane and 8 rows.  colsum (optional): colsum[by][c] = the tile's column sums (summed by the consumer in a fixed order).
static __global__ __launch_bounds__(256) void k_split_rows_tr(const float* __restrict__ y, int ld, uint16_t* __restrict__ tr, int M, int C)
{
    __shared__ float t[64][65];
    const int c0 = blockIdx.x * 64, m0 = blockIdx.y * 64, tid = threadIdx.x;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int r = p * 16 + (tid >> 4), q = (tid & 15) * 4;
        const f32x4 v = *reinterpret_cast<const f32x4*>(y + (size_t)(m0 + r) * ld + c0 + q);
        t[r][q] = v[0]; t[r][q + 1] = v[1]; t[r][q + 2] = v[2]; t[r][q + 3] = v[3];
    }
    __syncthreads();
    const size_t n = (size_t)M * C;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int e = tid + p * 256, c = e & 63, g = e >> 6;   // column c, rows 8 g .. 8 g + 7
        u32x2_t lo4[3], hi4[3];
        split3_f32x4(f32x4{t[8 * g][c], t[8 * g + 1][c], t[8 * g + 2][c], t[8 * g + 3][c]}, lo4);
        split3_f32x4(f32x4{t[8 * g + 4][c], t[8 * g + 5][c], t[8 * g + 6][c], t[8 * g + 7][c]}, hi4);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
            *reinterpret_cast<u32x4_t*>(tr + pl * n + (size_t)(c0 + c) * M + m0 + 8 * g) = u32x4_t{lo4[pl][0], lo4[pl][1], hi4[pl][0], hi4[pl][1]};
    }
}

// ------------------------------------------------------------------------------------------------
// k_igemm_red_b3: G[ko][n] = sum_{m in chunk} X(m,ko) * Y(m,n)  (the contraction runs over the ROWS m).
//   The MFMA wants, per lane, 8 consecutive contraction elements of its row: the LDS tiles are [ko][m] and [n][m] (m contiguous, the
//   swizzled 64-byte rows of k_igemm_b3).  X staging: wave w owns rows m0 + 8 w .. + 7 of the 32-row k-tile, lane l the columns
//   ko0 + 2 l, + 1: eight 8-byte loads (a wave reads 512 contiguous bytes per row), the optional Hadamard factor once (had[m / had_group],
//   had_group % 8 == 0), two exact 3-way splits per column, one ds_write_b128 per plane and column (chunk w of row ko).  Y staging: plain
//   16-byte copies of the planes.  Tile 128 x 128 (2 x 2 waves, 64 x 64 each); per k-tile 10 staging slices ride between the 12 MFMA
//   groups like in k_igemm_b3.  grid: (ko tiles * n tiles) * chunks workgroups, 1-D, chunk c on XCD c % 8 (chunks % 8 == 0): all tiles
//   of a row chunk read the same X / Y rows.  M % 32 == 0.
//   part[chunk][Kp * Np + Np]: the tile's sums, and the column sums of Y (bias gradient) from the ko-tile-0 workgroups.
// ------------------------------------------------------------------------------------------------
struct RedB3Args {
    const float* x; int x_ld;                       // [M][Kp]
    const float* had; int had_ld, had_group;        // optional second factor of X: x[m][k] * had[m / had_group][k] (nullptr: none)
    const uint16_t* ytr;                             // [3][Np][M] bf16 planes of Y (k_split_rows_tr)
    float* part; size_t part_stride;                 // [chunks][Kp * Np + Np]
    int M, Kp, Np, chunks;
};
template <int TERMS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_igemm_red_b3(RedB3Args a)
{
    static_assert(TERMS == 6 || TERMS == 9, "6 or 9 partial products");
    constexpr int TM = 2, TN = 2, BMK = 128, BN = 128;               // ko rows x n columns of the tile
    constexpr int PLANE = 128 * B3_ROW;                               // u16 per plane and operand
    constexpr int STAGE = 6 * PLANE;
    __shared__ __attribute__((aligned(16))) uint16_t smem[2 * STAGE];   // 96 KB
    __shared__ float sbq[512];
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

    // staging roles (X): rows m0 + 8 wave + j (j = 0 .. 7), columns ko0 + 2 lane + {0, 1}
    const int xc0 = ko0 + 2 * lane;
    const bool x_ok = xc0 < a.Kp;                   // (Kp is even: both columns or none)
    const float* xc = a.x + (x_ok ? xc0 : 0);
    const float* hc = a.had ? a.had + (x_ok ? xc0 : 0) : nullptr;
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    f32x2_t rx[2][8], rh[2];
    u32x4_t ry[2][2][3];
    float bs[2] = {0.f, 0.f};   // column sums of Y over the chunks this thread stages (pass 0: column tid / 4, pass 1: column 64 + tid / 4)
    const bool count_bias = kot == 0;
    auto prefetch_x = [&](auto set, int m0, int q) {   // slice q: rows 2 q, 2 q + 1 of the wave's eight (+ the Hadamard factor with slice 0)
        constexpr int S = decltype(set)::value;
        const int m = m0 + 8 * wave + 2 * q;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            rx[S][2 * q + j] = *reinterpret_cast<const f32x2_t*>(xc + (size_t)(m + j) * a.x_ld);   // (columns beyond Kp: column 0's values, zeroed at the commit)
        }
        if (q == 0) rh[S] = hc ? *reinterpret_cast<const f32x2_t*>(hc + (size_t)((m0 + 8 * wave) / a.had_group) * a.had_ld) : f32x2_t{1.f, 1.f};
    };
    auto prefetch_y = [&](auto set, int m0, int p) {   // pass p: 16-byte chunk e = tid + 256 p of the [128 n][32 m] tile, three planes
        constexpr int S = decltype(set)::value;
        const int e = tid + p * 256;
        const uint16_t* src = a.ytr + (size_t)(n0 + (e >> 2)) * a.M + m0 + (e & 3) * 8;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) ry[S][p][pl] = *reinterpret_cast<const u32x4_t*>(src + (size_t)pl * a.Np * a.M);
    };
    auto commit_x = [&](auto set, int stage, int c) {   // slice c: column 2 lane + c -> chunk `wave` of row 2 lane + c in the three planes
        constexpr int S = decltype(set)::value;
        const float hf = x_ok ? rh[S][c] : 0.f;
        u32x2_t lo4[3], hi4[3];
        split3_f32x4(f32x4{rx[S][0][c] * hf, rx[S][1][c] * hf, rx[S][2][c] * hf, rx[S][3][c] * hf}, lo4);
        split3_f32x4(f32x4{rx[S][4][c] * hf, rx[S][5][c] * hf, rx[S][6][c] * hf, rx[S][7][c] * hf}, hi4);
        uint16_t* Xs = smem + stage * STAGE;
        const int o = b3_off(2 * lane + c, wave);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<u32x4_t*>(&Xs[pl * PLANE + o]) = u32x4_t{lo4[pl][0], lo4[pl][1], hi4[pl][0], hi4[pl][1]};
    };
    auto commit_y = [&](auto set, int stage, int p, bool fresh) {
        constexpr int S = decltype(set)::value;
        const int e = tid + p * 256;
        if (count_bias && fresh) {   // bias gradient: hi + mid + lo is the f32 value again, exactly
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const uint32_t h0 = ry[S][p][0][w], h1 = ry[S][p][1][w], h2 = ry[S][p][2][w];
                bs[p] += (__uint_as_float(h0 << 16) + __uint_as_float(h1 << 16)) + __uint_as_float(h2 << 16);
                bs[p] += (__uint_as_float(h0 & 0xffff0000u) + __uint_as_float(h1 & 0xffff0000u)) + __uint_as_float(h2 & 0xffff0000u);
            }
        }
        uint16_t* Ys = smem + stage * STAGE + 3 * PLANE;
        const int o = b3_off(e >> 2, e & 3);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<u32x4_t*>(&Ys[pl * PLANE + o]) = ry[S][p][pl];
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

    auto prefetch_all = [&](auto set, int m0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) prefetch_x(set, m0, q);
#pragma unroll
        for (int p = 0; p < 2; ++p) prefetch_y(set, m0, p);
    };
    if (nkt > 0) {
        prefetch_all(Set0{}, tile_m(0));
        prefetch_all(Set1{}, tile_m(1));
        commit_x(Set0{}, 0, 0); commit_x(Set0{}, 0, 1); commit_y(Set0{}, 0, 0, true); commit_y(Set0{}, 0, 1, true);
        prefetch_all(Set0{}, tile_m(2));
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
                fa[F][tm][pl] = *reinterpret_cast<const bf16x8_t*>(&Xs[pl * PLANE + b3_off((wm * TM + tm) * 32 + j, s * 2 + h)]);
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
                fb[F][tn][pl] = *reinterpret_cast<const bf16x8_t*>(&Ys[pl * PLANE + b3_off((wn * TN + tn) * 32 + j, s * 2 + h)]);
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
    constexpr int PER = 1;   // at most one staging slice per MFMA group: 4 commits in the first k-step, 6 prefetches in the second
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
                if (q < 2) commit_y(set, cur ^ 1, q, fresh);          // (Y planes first, the X rows - streamed from HBM, loaded first - last)
                else if (q < 4) commit_x(set, cur ^ 1, q - 2);
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
                else if (q < 6) prefetch_y(set, m3, q - 4);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        cur ^= 1;
    };
    {
        int it = 0;
        for (; it + 1 < nkt; it += 2) { step(Set1{}, it); step(Set0{}, it + 1); }
        if (it < nkt) step(Set1{}, it);
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
        // thread tid staged the chunks (column tid / 4, rows 8 (tid % 4) ..) and (column 64 + tid / 4, same rows): two columns' partial sums
        __syncthreads();
        sbq[tid] = bs[0]; sbq[256 + tid] = bs[1];
        __syncthreads();
        if (tid < 128) { const float* q = sbq + (tid >> 6) * 256 + (tid & 63) * 4; part[(size_t)a.Kp * a.Np + n0 + tid] = (q[0] + q[1]) + (q[2] + q[3]); }
    }
}

}  // namespace bdr
