// Direct-to-LDS variant of k_igemm (global_load_lds_dwordx4 staging, XOR-swizzled tiles, mid-tile barrier
// with cross-tile fragment prefetch).  EXPERIMENTAL: bit-identical results to k_igemm but not faster on
// gfx950 at these shapes (the k loop is bound by LDS instruction issue and the per-k-tile barrier, not by
// the register staging it removes; see LAB.md section 6), so the product does not use it.  Only
// tools/probes/ includes this header.
#pragma once
#include "igemm.hpp"

namespace bdr {

// ------------------------------------------------------------------------------------------------
// k_igemm_dl: same contraction and policy interface as k_igemm, but the tiles go from global memory
// straight into LDS (global_load_lds_dwordx4: no VGPR staging, no ds_write, no vmcnt stall in front of
// the matrix pipe) through a ring of STAGES stages: the DMA for k-tile it+STAGES-1 is issued while
// k-tile it is being multiplied.
//
// LDS layout.  One wave-wide DMA instruction deposits 64 x 16 B at consecutive addresses (lane l at
// base + 16*l), so a row of a k-contiguous tile (A, and B of the dX kernels) is exactly 8 chunks = 128 B
// and cannot be padded.  Bank conflicts of the ds_read_b128 fragment reads are avoided by an XOR swizzle
// instead: chunk c of row r is stored at slot c ^ (r & 7) - the lane that owns LDS slot s of row r simply
// FETCHES global chunk s ^ (r & 7).  The n-contiguous B tile ([32][BN]) is read with ds_read_b32 along n
// and needs neither padding nor swizzle.
// ------------------------------------------------------------------------------------------------
#define BDR_GP(p) ((const __attribute__((address_space(1))) void*)(p))
#define BDR_LP(p) ((__attribute__((address_space(3))) void*)(p))

template <int N>
__device__ __forceinline__ void wait_vmcnt_barrier()
{
    static_assert(N >= 0 && N < 64, "vmcnt is 6 bits on gfx9");
    // memory clobber: no LDS access may move across.  lgkmcnt(0): my fragment reads of the stage that the
    // next iteration's DMA overwrites have returned.
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}

// One k-group (8 deep) of MFMA operands of a wave: A rows as f32x4 (4 consecutive k per lane), B either
// the same (k-major tile) or 4 scalars along n.
template <int TM, int TN>
struct Frag { f32x4 a[TM]; f32x4 b[TN]; };

template <int TM, int TN, int BN_, bool B_KMAJOR>
__device__ __forceinline__ void load_frag(Frag<TM, TN>& f, const float* __restrict__ As, const float* __restrict__ Bs, int arow0,
                                          int bcol0, int lane, int u)
{
    const int i = lane & 31, h = lane >> 5;
    const int slot = ((2 * u + h) ^ (i & 7)) * 4;   // arow0 / bcol0 / tm*32 are multiples of 8: row & 7 == i & 7
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) f.a[tm] = *reinterpret_cast<const f32x4*>(&As[(arow0 + tm * 32 + i) * BK + slot]);
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        if constexpr (B_KMAJOR) f.b[tn] = *reinterpret_cast<const f32x4*>(&Bs[(bcol0 + tn * 32 + i) * BK + slot]);
        else {
#pragma unroll
            for (int s = 0; s < 4; ++s) f.b[tn][s] = Bs[(8 * u + 4 * h + s) * BN_ + bcol0 + tn * 32 + i];
        }
    }
}
template <int TM, int TN>
__device__ __forceinline__ void mfma_group(const Frag<TM, TN>& f, f32x16 (&acc)[TM][TN])
{
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[tm][s], f.b[tn][s], acc[tm][tn], 0, 0, 0);
}

template <class P, int STAGES = 3>
__global__ __launch_bounds__(64 * P::WM * P::WN) void k_igemm_dl(typename P::Args args)
{
    IGEMM_TP(0);
    using A = typename P::A;
    static_assert(A::VEC == 4, "f32 operands only");
    constexpr int NW = P::WM * P::WN;
    constexpr int BM = P::WM * P::TM * 32, BN = P::WN * P::TN * 32;
    constexpr int A_INS = BM / 8, B_INS = BN / 8;          // 1 KB DMA instructions per k-tile
    static_assert(A_INS % NW == 0 && B_INS % NW == 0, "every wave issues the same number of DMA loads");
    constexpr int A_PW = A_INS / NW, B_PW = B_INS / NW, LOADS = A_PW + B_PW;
    constexpr int STAGE = BM * BK + BK * BN;               // floats; B is [32][BN] or [BN][32]: same size
    static_assert(STAGES >= 3 && (STAGES - 2) * LOADS < 64, "vmcnt range");
    __shared__ __attribute__((aligned(1024))) float smem[STAGES * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / P::WN, wn = wave % P::WN;
    const int NT_N = P::N(args) / BN;
    const int mt = blockIdx.x / NT_N, nt = blockIdx.x % NT_N;
    const int m0 = mt * BM, n0 = nt * BN;
    const int z = blockIdx.z, y = blockIdx.y;
    const int M = P::M(args);

    // DMA instruction q = p*NW + wave covers A rows 8q..8q+7: lane -> (row 8q + lane/8, LDS slot lane%8)
    typename A::Row rows[A_PW];
    int a_chunk[A_PW];
#pragma unroll
    for (int p = 0; p < A_PW; ++p) {
        const int r = (p * NW + wave) * 8 + (lane >> 3);
        int mr;
        const bool ok = P::vrow(args, m0 + r, mr);
        rows[p] = A::row(P::a_src(args, z), ok ? mr : M, M);
        a_chunk[p] = (lane & 7) ^ (r & 7);
    }
    const float* w = P::w(args, z, y);
    int kt0, kt1;
    P::kt_range(args, y, kt0, kt1);
    const int nkt = kt1 - kt0;
    auto tile = [&](int it) { return kt0 + min(it, nkt - 1); };   // clamped: the tail re-stages the last tile (never read)

    auto dma = [&](int kt, int stage) {
        float* As = smem + stage * STAGE;
        float* Bs = As + BM * BK;
#pragma unroll
        for (int p = 0; p < A_PW; ++p)
            __builtin_amdgcn_global_load_lds(BDR_GP(A::chunk(rows[p], kt, a_chunk[p])), BDR_LP(As + (p * NW + wave) * 256), 16, 0, 0);
#pragma unroll
        for (int p = 0; p < B_PW; ++p) {
            const int q = p * NW + wave;
            const float* src;
            if constexpr (!P::B_TR) {   // [32 k][BN n]: instruction q covers k rows (64 / (BN/4)) q ..
                const int e = q * 64 + lane;
                const int kr = e / (BN / 4), n4 = e % (BN / 4);
                src = w + (size_t)(kt * BK + kr) * P::N(args) + n0 + n4 * 4;
            } else {                    // [BN n'][32 k'] swizzled like A
                const int TPT = P::KP(args) / BK;
                const int tap = kt / TPT, c0 = (kt % TPT) * BK;
                const int np = q * 8 + (lane >> 3);
                src = w + ((size_t)P::tap_index(y, tap) * P::N(args) + n0 + np) * P::KP(args) + c0 + (((lane & 7) ^ (np & 7)) * 4);
            }
            __builtin_amdgcn_global_load_lds(BDR_GP(src), BDR_LP(Bs + q * 256), 16, 0, 0);
        }
    };

    // epilogue operands first (older than every DMA load, so the vmcnt arithmetic below only counts tiles)
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
            if (P::vrow(args, mv, mrow[tm][r])) okmask[tm] |= 1u << r;
            else mrow[tm][r] = 0;
#pragma unroll
            for (int tn = 0; tn < P::TN; ++tn)
                aux[tm][tn][r] = P::epi_load(epi, mrow[tm][r], n0 + (wn * P::TN + tn) * 32 + j);
        }
    }

    f32x16 acc[P::TM][P::TN];
#pragma unroll
    for (int tm = 0; tm < P::TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < P::TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s) dma(tile(s), s);
    wait_vmcnt_barrier<(STAGES - 2) * LOADS>();   // tile 0 has landed (everyone's share)
    IGEMM_TP(1);

    // Per k-tile: 4 k-groups, fragments double-buffered in registers.  The workgroup barrier sits in the
    // MIDDLE of the k-tile: by then tile it+1 has landed, so the first fragments of tile it+1 are fetched
    // while the last MFMAs of tile it run and the matrix pipe never waits for LDS across the tile boundary.
    const int ar0 = wm * P::TM * 32, bc0 = wn * P::TN * 32;
    Frag<P::TM, P::TN> f0, f1;
    load_frag<P::TM, P::TN, BN, P::B_TR>(f0, smem, smem + BM * BK, ar0, bc0, lane, 0);
    int cur = 0;
    // sched_barrier: a wave issues in order and a dependent MFMA blocks at issue until its predecessor
    // retires, so the LDS reads of group u+1 must be IN FRONT of the MFMAs of group u in program order
    // (the scheduler would otherwise sink them to just before their use, behind four blocking MFMAs).
#define BDR_PIN() __builtin_amdgcn_sched_barrier(0)
    for (int it = 0; it < nkt; ++it) {
        const float* As = smem + cur * STAGE;
        const float* Bs = As + BM * BK;
        const int nx = cur + 1 == STAGES ? 0 : cur + 1;
        const int fr = cur == 0 ? STAGES - 1 : cur - 1;     // stage of tile it-1 (free after the barrier)
        load_frag<P::TM, P::TN, BN, P::B_TR>(f1, As, Bs, ar0, bc0, lane, 1);
        BDR_PIN();
        mfma_group(f0, acc);
        BDR_PIN();
        load_frag<P::TM, P::TN, BN, P::B_TR>(f0, As, Bs, ar0, bc0, lane, 2);
        BDR_PIN();
        mfma_group(f1, acc);
        BDR_PIN();
        wait_vmcnt_barrier<(STAGES - 3) * LOADS>();         // tile it+1 landed everywhere; tile it-1's stage is free
        load_frag<P::TM, P::TN, BN, P::B_TR>(f1, As, Bs, ar0, bc0, lane, 3);
        BDR_PIN();
        mfma_group(f0, acc);
        dma(tile(it + STAGES - 1), fr);
        BDR_PIN();
        const float* An = smem + nx * STAGE;
        load_frag<P::TM, P::TN, BN, P::B_TR>(f0, An, An + BM * BK, ar0, bc0, lane, 0);   // tile it+1, first group
        BDR_PIN();
        mfma_group(f1, acc);
        BDR_PIN();
        cur = nx;
    }
    IGEMM_TP(2);

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
    IGEMM_TP(3);
}

template <class P, int STAGES = 3>
inline hipError_t launch_igemm_dl(hipStream_t st, dim3 grid, const typename P::Args& args)
{
    hipLaunchKernelGGL((k_igemm_dl<P, STAGES>), grid, dim3(64 * P::WM * P::WN), 0, st, args);
    return hipGetLastError();
}


}  // namespace bdr
