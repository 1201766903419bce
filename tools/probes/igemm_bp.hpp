// Implicit GEMM on the bf16 matrix cores with operands that are ALREADY split into three exact bf16 planes in HBM.
//
// Every f32 value x of an activation / gradient / weight tensor is kept as x = hi + mid + lo with three bf16 terms (8 + 8 + 8
// significant bits, truncation split: split3()), written once by whoever produces x (the epilogue of the layer before, the
// optimizer step for weights).  A product a * b is then the sum of nine exact bf16 x bf16 products; the six that are not below
// 2^-24 |a||b| (everything but mid*lo, lo*mid, lo*lo) are accumulated in f32 by v_mfma_f32_32x32x16_bf16, smallest first: the
// error class of the FP32 MFMA / an fmaf chain (each of those rounds every product-sum to 2^-24 too), at 192 matrix cycles per
// 16 k and 32x32 tile instead of 512 (v_mfma_f32_32x32x2_f32).  Measured against the FP32 kernels: <= 2e-6 relative.
//
// What this buys over splitting inside the GEMM (tools/probes/igemm_b3.hpp, round 1: only 9-25 % faster): no VALU work in the
// k loop at all - staging is 16-byte global loads and 16-byte LDS stores of finished bf16 - and the split costs ~8 VALU
// operations per OUTPUT element once, in an epilogue, instead of per staged operand element in every consumer.
//
// Same contraction, row maps and epilogue hooks as k_igemm (igemm.hpp).  Policy P:
//   WM, WN, TM, TN, vrow / vrow_y, M, N, kt_range, epi, epi_load, store            as for k_igemm
//   A::Row, A::row(base, m, M)                                                       row context of the A gather
//   A::aoff(row, kt, q, ok) -> element offset of the 8-element chunk q (0..3) of k-tile kt from the tensor base
//                              (ok = false: structurally zero, loads nothing)
//   a_planes(args, z) -> const uint16_t*   plane 0 of the A tensor;  a_plane_stride(args) elements between planes
//   b_chunk(args, z, y, plane, kt, n, kq) -> const uint4*   the 8 bf16 (k = kt*32 + kq*8 ..) of B column n, k-major planes
// LDS: per stage and operand three planes [rows][32 k + 8 pad] bf16 (80-byte rows: conflict-free ds_read_b128 fragments).
#pragma once
#include "igemm.hpp"

namespace bdr {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));

constexpr int BP_LDR = 40;   // row stride of a plane tile in u16 units (32 k + 8 pad = 80 B)

// exact three-way split of one float: x == hi + mid + lo, each exactly representable in bf16 (upper 16 bits of an f32)
__device__ __forceinline__ void split3(float x, uint32_t& hi, uint32_t& mid, uint32_t& lo)
{
    hi = __float_as_uint(x) & 0xffff0000u;
    const float r1 = x - __uint_as_float(hi);              // exact
    mid = __float_as_uint(r1) & 0xffff0000u;
    const float r2 = r1 - __uint_as_float(mid);            // exact, <= 8 significant bits
    lo = __float_as_uint(r2) & 0xffff0000u;
}

// one value -> its three planes (element stores; epilogues whose lanes own single elements)
__device__ __forceinline__ void store_split3(gptr<uint16_t> planes, size_t plane_stride, size_t idx, float x)
{
    uint32_t hi, mid, lo;
    split3(x, hi, mid, lo);
    planes[idx] = (uint16_t)(hi >> 16);
    planes[plane_stride + idx] = (uint16_t)(mid >> 16);
    planes[2 * plane_stride + idx] = (uint16_t)(lo >> 16);
}

// f32 [R][C] -> three bf16 planes, as is (T = 0: dst[plane][r][c]) or transposed (T = 1: dst[plane][c][r])
__global__ __launch_bounds__(256) void k_split_planes(const float* __restrict__ src, uint16_t* __restrict__ dst, int R, int C, int T)
{
    const size_t n = (size_t)R * C;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int r = (int)(i / C), c = (int)(i % C);
    uint32_t hi, mid, lo;
    split3(src[i], hi, mid, lo);
    const size_t o = T ? (size_t)c * R + r : i;
    dst[o] = (uint16_t)(hi >> 16); dst[n + o] = (uint16_t)(mid >> 16); dst[2 * n + o] = (uint16_t)(lo >> 16);
}

template <class P, int TERMS = 6>
__global__ __launch_bounds__(64 * P::WM * P::WN) void k_igemm_bp(typename P::Args args)
{
    if constexpr (has_start_signal<typename P::Args>::value) start_signal(args.sig_flag, args.sig_epoch);
    using A = typename P::A;
    static_assert(TERMS == 6 || TERMS == 9, "6 or 9 partial products");
    constexpr int NW = P::WM * P::WN, NT = 64 * NW;
    constexpr int BM = P::WM * P::TM * 32, BN = P::WN * P::TN * 32;
    constexpr int A_CH = BM * 4, A_PASSES = (A_CH + NT - 1) / NT;     // 16-byte chunks per plane
    constexpr int B_CH = BN * 4, B_PASSES = (B_CH + NT - 1) / NT;
    static_assert(NT % 4 == 0, "a thread keeps its chunk column across passes");
    constexpr int PLANE_A = BM * BP_LDR, PLANE_B = BN * BP_LDR;       // u16
    constexpr int STAGE = 3 * (PLANE_A + PLANE_B);                    // u16
    __shared__ __attribute__((aligned(16))) uint16_t smem[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / P::WN, wn = wave % P::WN;
    const int NT_N = P::N(args) / BN;
    const int mt = blockIdx.x / NT_N, nt = blockIdx.x % NT_N;
    const int m0 = mt * BM, n0 = nt * BN;
    const int z = blockIdx.z, y = blockIdx.y;
    const int M = P::M(args);

    const int a_q = tid & 3, a_r = tid >> 2;
    typename A::Row rows[A_PASSES];
#pragma unroll
    for (int p = 0; p < A_PASSES; ++p) {
        int mr;
        const bool ok = vrow_of<P>(args, y, m0 + p * (NT / 4) + a_r, mr);
        rows[p] = A::row(nullptr, ok ? mr : M, M);
    }
    const uint16_t* apl = P::a_planes(args, z);
    const size_t aps = P::a_plane_stride(args);
    int kt0, kt1;
    P::kt_range(args, y, kt0, kt1);
    const int nkt = kt1 - kt0;
    auto tile = [&](int it) { return kt0 + min(it, nkt - 1); };

    u32x4_t ra[2][A_PASSES][3], rb[2][B_PASSES][3];
    auto prefetch = [&](auto set, int kt) {
        constexpr int S = decltype(set)::value;
#pragma unroll
        for (int p = 0; p < A_PASSES; ++p) {
            if (A_CH % NT != 0 && tid + p * NT >= A_CH) continue;
            bool ok;
            const size_t off = A::aoff(rows[p], kt, a_q, ok);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                const u32x4_t t = *reinterpret_cast<const u32x4_t*>(apl + pl * aps + off);
                ra[S][p][pl] = ok ? t : u32x4_t{0u, 0u, 0u, 0u};
            }
        }
#pragma unroll
        for (int p = 0; p < B_PASSES; ++p) {
            const int e = tid + p * NT;
            if (B_CH % NT != 0 && e >= B_CH) continue;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) rb[S][p][pl] = *reinterpret_cast<const u32x4_t*>(P::b_chunk(args, z, y, pl, kt, n0 + e / 4, e % 4));
        }
    };
    auto commit = [&](auto set, int stage) {
        constexpr int S = decltype(set)::value;
        uint16_t* As = smem + stage * STAGE;
        uint16_t* Bs = As + 3 * PLANE_A;
#pragma unroll
        for (int p = 0; p < A_PASSES; ++p) {
            if (A_CH % NT != 0 && tid + p * NT >= A_CH) continue;
            const int o = (p * (NT / 4) + a_r) * BP_LDR + a_q * 8;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<u32x4_t*>(&As[pl * PLANE_A + o]) = ra[S][p][pl];
        }
#pragma unroll
        for (int p = 0; p < B_PASSES; ++p) {
            const int e = tid + p * NT;
            if (B_CH % NT != 0 && e >= B_CH) continue;
            const int o = (e / 4) * BP_LDR + (e % 4) * 8;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<u32x4_t*>(&Bs[pl * PLANE_B + o]) = rb[S][p][pl];
        }
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
                aux[tm][tn][r] = P::epi_load(epi, mrow[tm][r], n0 + (wn * P::TN + tn) * 32 + j);
        }
    }
    __syncthreads();

    int cur = 0;
    auto step = [&](auto set, int it) {   // set holds tile it+1; refilled with tile it+3
        const uint16_t* As = smem + cur * STAGE;
        const uint16_t* Bs = As + 3 * PLANE_A;
#pragma unroll
        for (int s = 0; s < 2; ++s) {     // two k-steps of 16
            bf16x8_t af[P::TM][3], bfr[P::TN][3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
                for (int tm = 0; tm < P::TM; ++tm)
                    af[tm][pl] = *reinterpret_cast<const bf16x8_t*>(&As[pl * PLANE_A + ((wm * P::TM + tm) * 32 + j) * BP_LDR + s * 16 + h * 8]);
#pragma unroll
                for (int tn = 0; tn < P::TN; ++tn)
                    bfr[tn][pl] = *reinterpret_cast<const bf16x8_t*>(&Bs[pl * PLANE_B + ((wn * P::TN + tn) * 32 + j) * BP_LDR + s * 16 + h * 8]);
            }
            // partial products, smallest first: (a plane, b plane)
            constexpr int ORD9[9][2] = {{2, 2}, {2, 1}, {1, 2}, {2, 0}, {0, 2}, {1, 1}, {1, 0}, {0, 1}, {0, 0}};
#pragma unroll
            for (int t = 9 - TERMS; t < 9; ++t)
#pragma unroll
                for (int tm = 0; tm < P::TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < P::TN; ++tn)
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[tm][ORD9[t][0]], bfr[tn][ORD9[t][1]], acc[tm][tn], 0, 0, 0);
            if (s == 0) commit(set, cur ^ 1);
            else prefetch(set, tile(it + 3));
        }
        __syncthreads();
        cur ^= 1;
    };
    for (int it = 0; it < nkt; it += 2) {
        step(Set1{}, it);
        if (it + 1 < nkt) step(Set0{}, it + 1);
    }

#pragma unroll
    for (int tm = 0; tm < P::TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < P::TN; ++tn) {
            const int n = n0 + (wn * P::TN + tn) * 32 + j;
#pragma unroll
            for (int r = 0; r < 16; ++r) asm volatile("" ::"v"(aux[tm][tn][r]));
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (okmask[tm] >> r & 1) P::store(epi, mrow[tm][r], n, acc[tm][tn][r], aux[tm][tn][r]);
        }
}

template <class P, int TERMS = 6>
inline hipError_t launch_igemm_bp(hipStream_t st, dim3 grid, const typename P::Args& args, unsigned flags = 0, hipEvent_t stop = nullptr)
{
    hipExtLaunchKernelGGL((k_igemm_bp<P, TERMS>), grid, dim3(64 * P::WM * P::WN), 0, st, nullptr, stop, flags, args);
    return hipGetLastError();
}

}  // namespace bdr
