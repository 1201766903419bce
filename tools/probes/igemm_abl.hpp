// Instrumented copy of the product's k_igemm (border_amd/csrc/igemm.hpp) for the timing probes: per-workgroup phase stamps
// (IGEMM_TRACE), per-phase cycles inside the k loop (IGEMM_TRACE2) and the IGEMM_ABL ablations (results are WRONG when non-zero).
// Lives in namespace bdr_abl and takes the product's policies unchanged; the product header carries none of this.
// Snapshot of the round-4 kernel body - re-copy when the product kernel changes.
#pragma once
#include "igemm.hpp"

namespace bdr_abl {
using namespace bdr;

#ifndef IGEMM_ABL    // tools/probes only: timing ablations (results are wrong when non-zero)
#define IGEMM_ABL 0  // 1: no global prefetch in the loop, 2: no LDS commit, 4: no barrier, 8: no LDS fragment reads
#endif

// between(u) is called after the MFMAs of k-group u have been issued: the staging work of the next
// tiles is sliced into those gaps so that it executes in the shadow of the (dependent, 64-cycle)
// MFMAs instead of in front of them.
// B_KMAJOR: the B tile is stored like the A tile ([n][LDA], k contiguous: transposed-weight operands of
// the dX kernels are k-contiguous in memory) and a lane fetches its 4 k-values with one ds_read_b128.
template <int TM, int TN, int LDB, bool B_KMAJOR, class F>
__device__ __forceinline__ void mfma_ktile(const float* __restrict__ As, const float* __restrict__ Bs, int arow0,
                                           int bcol0, int lane, f32x16 (&acc)[TM][TN], F&& between)
{
    const int i = lane & 31, h = lane >> 5;
#pragma unroll
    for (int u = 0; u < BK / 8; ++u) {
        f32x4 a[TM];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
            a[tm] = (IGEMM_ABL & 8) ? f32x4{1.f * lane, 2.f, 3.f, 4.f}
                                    : *reinterpret_cast<const f32x4*>(&As[(arow0 + tm * 32 + i) * LDA + 8 * u + 4 * h]);
        f32x4 bq[TN];
        if constexpr (B_KMAJOR) {
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
                bq[tn] = *reinterpret_cast<const f32x4*>(&Bs[(bcol0 + tn * 32 + i) * LDA + 8 * u + 4 * h]);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            float b[TN];
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                if constexpr (B_KMAJOR) b[tn] = bq[tn][s];
                else b[tn] = (IGEMM_ABL & 8) ? 0.5f * lane : Bs[(8 * u + 4 * h + s) * LDB + bcol0 + tn * 32 + i];
            }
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm][s], b[tn], acc[tm][tn], 0, 0, 0);
        }
        between(u);
    }
}

#ifdef IGEMM_TRACE   // tools/probes only: per-workgroup phase timestamps (100 MHz wall clock)
__device__ unsigned long long* g_igemm_trace;
#define IGEMM_TP(slot) do { if (threadIdx.x == 0 && bdr_abl::g_igemm_trace) { \
    unsigned long long* t_ = bdr_abl::g_igemm_trace + ((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8; \
    t_[slot] = wall_clock64(); t_[4 + (slot)] = clock64(); } } while (0)
#else
#define IGEMM_TP(slot) do { } while (0)
#endif
#ifdef IGEMM_TRACE2  // tools/probes only: cycles of one wave inside the k loop, per phase (s_memtime)
__device__ unsigned long long* g_igemm_phase;
#define IGEMM_PH(var) const long long var = clock64()
#else
#define IGEMM_PH(var) do { } while (0)
#endif

template <class P, int TEAMS = 1>
__global__ __launch_bounds__(64 * P::WM * P::WN * TEAMS) void k_igemm(typename P::Args args)
{
    IGEMM_TP(0);
    if constexpr (has_start_signal<typename P::Args>::value) start_signal(args.sig_flag, args.sig_epoch);
    using A = typename P::A;
    constexpr int NW = P::WM * P::WN, NT = 64 * NW;          // waves / threads per team
    constexpr int BM = P::WM * P::TM * 32, BN = P::WN * P::TN * 32;
    constexpr int LDB = BN;
    constexpr int APR = BK / A::VEC;                         // staging loads per A row
    static_assert(NT % APR == 0, "thread count must keep the k-quad of a thread fixed across passes");
    constexpr int A_ELEMS = BM * APR;
    constexpr int A_PASSES = (A_ELEMS + NT - 1) / NT;
    constexpr int AV = A::VEC / 4;
    constexpr int B_ELEMS = BK * BN / 4;
    constexpr int B_VECS = (B_ELEMS + NT - 1) / NT;          // f32x4 per thread for the B tile
    static_assert(TEAMS == 1 || TEAMS == 2, "one or two teams");

    // two LDS stages per team: tile t+1 is written while tile t feeds the matrix pipe (one barrier per k-tile)
    constexpr int STAGE = BM * LDA + (P::B_TR ? BN * LDA : BK * LDB);
    static_assert(TEAMS == 1 || 2 * STAGE >= BM * BN, "team reduction buffer must fit one team's stages");
    __shared__ __attribute__((aligned(16))) float smem_all[2 * STAGE * TEAMS];

    const int team = TEAMS == 1 ? 0 : (int)(threadIdx.x / NT);
    float* smem = smem_all + team * 2 * STAGE;
    const int tid = threadIdx.x % NT, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / P::WN, wn = wave % P::WN;
    const int NT_N = P::N(args) / BN;
    const int M = P::M(args);
    // block -> (m-tile, n-tile, instance) map.  Workgroup b runs on XCD b % 8 and every XCD has its own L2, so which tiles
    // share an XCD decides how often an operand crosses the fabric (PMC: l1 forward moved 71.6 MB for 26.5 MB of operands
    // and results with the plain map, every XCD streaming all of A).
    int mt, nt, z = blockIdx.z, y = blockIdx.y;
    if constexpr (xmap_of<P>::value == 3) {
        // split-K slice = XCD: XCD j multiplies k-slice j of every output tile, so each XCD's L2 sees 1/8 of A's columns and 1/8
        // of B's rows exactly once (PMC, round 3: the (instance, n-tile pair) map fetched 38.6 MB for 19.3 MB of operands - every A
        // element crossed the fabric four times).  grid: (8 * m-tiles * n-tiles, 1, instances)
        y = blockIdx.x & 7;
        const int j = blockIdx.x >> 3;
        nt = j % NT_N; mt = j / NT_N;
    } else if constexpr (xmap_of<P>::value == 1) {
        // contiguous runs of tiles per XCD, m fastest: the m-tiles of one n-tile (same B columns) meet in one L2
        const int MT = (M + BM - 1) / BM, per = gridDim.x >> 3;   // (flat rows: RPIP == 0)
        const int t = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
        if (t >= MT * NT_N) return;
        nt = t / MT; mt = t % MT;
    } else if constexpr (xmap_of<P>::value == 2) {
        // 8 n-tiles, instances in pairs: XCD = (instance parity, pair of n-tiles); each B element is read by one XCD,
        // each A element by four.  grid: (16 * m-tiles, splits, instances / 2)
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        z = blockIdx.z * 2 + (xcd >> 2);
        nt = (xcd & 3) * 2 + (j & 1); mt = j >> 1;
    } else {
        mt = blockIdx.x / NT_N; nt = blockIdx.x % NT_N;
    }
    const int m0 = mt * BM, n0 = nt * BN;                    // m0: virtual row

    // per-thread staging coordinates: element e = tid + p*NT of the A tile -> (row e / APR, k-quad e % APR)
    const int a_q = tid % APR, a_r = tid / APR;
    typename A::Row rows[A_PASSES];
#pragma unroll
    for (int p = 0; p < A_PASSES; ++p) {
        int mr;
        const bool ok = vrow_of<P>(args, y, m0 + p * (NT / APR) + a_r, mr);
        rows[p] = A::row(P::a_src(args, z), ok ? mr : M, M);   // invalid rows alias row 0 and are never stored
    }
    const float* w = P::w(args, z, y);

    int kt0, kt1;
    P::kt_range(args, y, kt0, kt1);
    // team g owns k-tiles kt0+g, kt0+g+TEAMS, ...; both teams run the same number of iterations so the
    // workgroup barriers match (the shorter team idles through its last one)
    const int iters = (kt1 - kt0 + TEAMS - 1) / TEAMS;
    const int my_n = kt0 + team < kt1 ? (kt1 - kt0 - team + TEAMS - 1) / TEAMS : 0;
    auto tile = [&](int it) { return kt0 + team + min(it, max(my_n - 1, 0)) * TEAMS; };   // clamped to my last tile

    // Two register sets: global loads are issued ~2 k-tiles before they are committed to LDS (one k-tile
    // of MFMAs is ~0.5 us; an L2/MALL round trip under load is longer than that).
    f32x4 ra[2][A_PASSES][AV];
    f32x4 rb[2][B_VECS];
    constexpr bool HAD = a_has_had<A>::value;
    f32x4 rh[2][HAD ? A_PASSES : 1];   // second factor of a Hadamard A operand (ADenseHad)
    auto prefetch_a = [&](auto set, int kt) {
        constexpr int S = decltype(set)::value;
#pragma unroll
        for (int p = 0; p < A_PASSES; ++p)
            if (A_ELEMS % NT == 0 || tid + p * NT < A_ELEMS) {
                A::load(rows[p], kt, a_q, ra[S][p]);
                if constexpr (HAD) rh[S][p] = A::load_had(rows[p], kt, a_q);
            }
    };
    auto prefetch_b = [&](auto set, int kt) {
        constexpr int S = decltype(set)::value;
#pragma unroll
        for (int v = 0; v < B_VECS; ++v) {
            const int e = tid + v * NT;
            if (B_ELEMS % NT != 0 && e >= B_ELEMS) continue;
            if constexpr (!P::B_TR) {
                const int kr = e / (BN / 4), n4 = e % (BN / 4);
                rb[S][v] = *reinterpret_cast<const f32x4*>(w + (size_t)(kt * BK + kr) * P::N(args) + n0 + n4 * 4);
            } else {
                // k-tile kt = (tap, c0); element (k'=c0+kq*4.., n') = w[(tap*NP + n0+n')*KP + c0 + kq*4]
                const int TPT = P::KP(args) / BK;
                const int tap = kt / TPT, c0 = (kt % TPT) * BK;
                const int kq = e % 8, np = e / 8;
                size_t wrow;
                if constexpr (has_b_row<P>::value) wrow = (size_t)P::b_row(y, tap, n0 + np);
                else wrow = (size_t)P::tap_index(y, tap) * P::N(args) + n0 + np;
                rb[S][v] = *reinterpret_cast<const f32x4*>(w + wrow * P::KP(args) + c0 + kq * 4);
            }
        }
    };
    auto commit_a = [&](auto set, int stage) {
        constexpr int S = decltype(set)::value;
        float* As = smem + stage * STAGE;
#pragma unroll
        for (int p = 0; p < A_PASSES; ++p) {
            if (A_ELEMS % NT != 0 && tid + p * NT >= A_ELEMS) continue;
#pragma unroll
            for (int j = 0; j < AV; ++j) {
                f32x4 v = ra[S][p][j];
                if constexpr (HAD) v *= rh[S][p];
                *reinterpret_cast<f32x4*>(&As[(p * (NT / APR) + a_r) * LDA + a_q * A::VEC + j * 4]) = v;
            }
        }
    };
    auto commit_b = [&](auto set, int stage) {
        constexpr int S = decltype(set)::value;
        float* Bs = smem + stage * STAGE + BM * LDA;
#pragma unroll
        for (int v = 0; v < B_VECS; ++v) {
            const int e = tid + v * NT;
            if (B_ELEMS % NT != 0 && e >= B_ELEMS) continue;
            if constexpr (!P::B_TR) {
                const int kr = e / (BN / 4), n4 = e % (BN / 4);
                *reinterpret_cast<f32x4*>(&Bs[kr * LDB + n4 * 4]) = rb[S][v];
            } else {
                const int kq = e % 8, np = e / 8;
                *reinterpret_cast<f32x4*>(&Bs[np * LDA + kq * 4]) = rb[S][v];   // [n'][k'] like the A tile
            }
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

    // Software pipeline, branch-free body (tile indices are clamped, so the tail re-stages the last
    // tile into the idle stage: harmless).  While stage `cur` (tile it) feeds the matrix pipe, register
    // set (it+1)&1 holds tile it+1 and the other set tile it+2.
    if (my_n > 0) {
        prefetch_a(Set0{}, tile(0)); prefetch_b(Set0{}, tile(0));
        prefetch_a(Set1{}, tile(1)); prefetch_b(Set1{}, tile(1));
        commit_a(Set0{}, 0); commit_b(Set0{}, 0);
        prefetch_a(Set0{}, tile(2)); prefetch_b(Set0{}, tile(2));
    }

    // Epilogue operands (row map, bias / ReLU mask) are fetched here, so the k loop hides their latency and
    // the 16 stores per tile go out back to back.  (vmcnt counts stores as well as loads on gfx9: a
    // load -> wait -> store sequence per element would serialise 16 memory round trips per wave.)
    const int j = lane & 31, h = lane >> 5;
    typename P::Epi epi = P::epi(args, z, y);   // output / bias / mask pointers, pinned in SGPRs for the epilogue
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
    IGEMM_TP(1);
    int cur = 0;
#ifdef IGEMM_TRACE2
    long long ph[6] = {0, 0, 0, 0, 0, 0};
#endif
    auto step = [&](auto set, int it) {   // set = (it+1)&1: holds tile it+1; refilled with tile it+3
        IGEMM_PH(t_a);
#ifdef IGEMM_TRACE2
        long long t_b = t_a, t_c = t_a, t_d = t_a;
#endif
        if (it < my_n) {                  // team-uniform
            const float* As = smem + cur * STAGE;
            const int k3 = tile(it + 3);
            mfma_ktile<P::TM, P::TN, LDB, P::B_TR>(As, As + BM * LDA, wm * P::TM * 32, wn * P::TN * 32, lane, acc, [&](int u) {
#ifdef IGEMM_TRACE2
                if (u == 0) t_b = clock64();
                if (u == 3) t_d = clock64();
#endif
                if (u == 0) { if (!(IGEMM_ABL & 2)) { commit_a(set, cur ^ 1); commit_b(set, cur ^ 1); } }   // tile it+1 -> idle stage
                else if (u == 1) { if (!(IGEMM_ABL & 1)) prefetch_a(set, k3); }                       // tile it+3 global loads
                else if (u == 2) { if (!(IGEMM_ABL & 1)) prefetch_b(set, k3); }
#ifdef IGEMM_TRACE2
                if (u == 0) t_c = clock64();
#endif
            });
        }
        IGEMM_PH(t_e);
        if (!(IGEMM_ABL & 4)) __syncthreads();
#ifdef IGEMM_TRACE2
        const long long t_f = clock64();
        ph[0] += t_b - t_a; ph[1] += t_c - t_b; ph[2] += t_d - t_c; ph[3] += t_e - t_d; ph[4] += t_f - t_e; ph[5] += 1;
#endif
        cur ^= 1;
    };
    for (int it = 0; it < iters; it += 2) {
        step(Set1{}, it);
        if (it + 1 < iters) step(Set0{}, it + 1);
    }
    IGEMM_TP(2);
#ifdef IGEMM_TRACE2
    if (threadIdx.x == 0 && g_igemm_phase) {
        unsigned long long* o = g_igemm_phase + ((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8;
        for (int q = 0; q < 6; ++q) o[q] = (unsigned long long)ph[q];
    }
#endif
    if constexpr (TEAMS == 2) {   // acc(team 0) += acc(team 1), through team 1's (now idle) stages
        constexpr int PER_WAVE = P::TM * P::TN * 16 * 64;
        static_assert(2 * STAGE >= NW * PER_WAVE, "team reduction buffer");
        float* red = smem_all + 2 * STAGE;
        if (team == 1) {
#pragma unroll
            for (int tm = 0; tm < P::TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < P::TN; ++tn)
#pragma unroll
                    for (int r = 0; r < 16; ++r) red[wave * PER_WAVE + ((tm * P::TN + tn) * 16 + r) * 64 + lane] = acc[tm][tn][r];
        }
        __syncthreads();
        if (team == 1) return;
#pragma unroll
        for (int tm = 0; tm < P::TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < P::TN; ++tn)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[tm][tn][r] += red[wave * PER_WAVE + ((tm * P::TN + tn) * 16 + r) * 64 + lane];
    }

#pragma unroll
    for (int tm = 0; tm < P::TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < P::TN; ++tn) {
            const int n = n0 + (wn * P::TN + tn) * 32 + j;
            // land the operand loads here, in straight-line code: otherwise every conditional store block
            // gets its own vmcnt(0), which also waits for the previous block's store
#pragma unroll
            for (int r = 0; r < 16; ++r) asm volatile("" ::"v"(aux[tm][tn][r]));
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (okmask[tm] >> r & 1) P::store(epi, mrow[tm][r], n, acc[tm][tn][r], aux[tm][tn][r]);
        }
    IGEMM_TP(3);
}


template <class P, int TEAMS>
inline hipError_t launch_igemm(hipStream_t st, dim3 grid, const typename P::Args& args)
{
    hipLaunchKernelGGL((k_igemm<P, TEAMS>), grid, dim3(64 * P::WM * P::WN * TEAMS), 0, st, args);
    return hipGetLastError();
}

}  // namespace bdr_abl
