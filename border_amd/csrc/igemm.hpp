// FP32-MFMA implicit-GEMM building blocks for gfx950 (v_mfma_f32_32x32x2_f32: exact f32, bitwise a
// k-ordered fmaf chain, 157 TFLOP/s chip peak).
//
// Two kernel shapes cover every contraction of the Nature-CNN step:
//   k_igemm      C[m][n]  = sum_k A(m,k) * B(k,n)      forward convs / linears and the
//                                                       input-gradient (transposed conv as gather)
//   k_igemm_red  G[k][n]  = sum_m A(m,k) * Y(m,n)      weight gradients (reduction over rows,
//                                                       split across workgroups, partials summed
//                                                       by k_reduce_partials)
// A(m,k) is never materialised: policies compute im2col addresses from compile-time geometry.
// Activations are NHWC ([B][H][W][C] == the row-major GEMM C matrix), weights are [K][N] with
// K ordered (kh,kw,cin) (conv1: (cin,kh,kw), matching its NCHW u8 input).
//
// MFMA operand maps (cdna_hip_programming.md section 3):
//   A: lane l holds A[i=l&31][k=l>>5];  B: lane l holds B[k=l>>5][j=l&31];
//   D: acc[r] = D[(r&3) + 8*(r>>2) + 4*(l>>5)][l&31].
// The k-slot an element lands in is free as long as A and B agree, so a lane reads 4 consecutive
// k of its row with one ds_read_b128 (half h takes k = 8u+4h+s) and feeds 4 MFMAs.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>
#include <utility>

#include <hip/hip_ext.h>

namespace bdr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 32;        // k-tile
constexpr int LDA = BK + 4;   // A tile row stride (floats): conflict-free ds_read_b128, 16B aligned

template <int IH_, int IW_, int CIN_, int KH_, int KW_, int S_, int OH_, int OW_, int COUT_>
struct Geom {
    static constexpr int IH = IH_, IW = IW_, CIN = CIN_, KH = KH_, KW = KW_, S = S_, OH = OH_, OW = OW_,
                         COUT = COUT_;
    static constexpr int K = KH * KW * CIN;
};
using GeomC1 = Geom<84, 84, 4, 8, 8, 4, 20, 20, 32>;
using GeomC2 = Geom<20, 20, 32, 4, 4, 2, 9, 9, 64>;
using GeomC3 = Geom<9, 9, 64, 3, 3, 1, 7, 7, 64>;
using GeomL1 = Geom<1, 1, 3136, 1, 1, 1, 1, 1, 512>;

// 16 bytes of zeros in global memory: what structurally-zero im2col taps load in the direct-to-LDS path
__device__ __attribute__((aligned(16))) static const float g_zero_page[4] = {0.f, 0.f, 0.f, 0.f};
__device__ __forceinline__ const float* zero_page() { return g_zero_page; }

// ------------------------------------------------------------------------------------------------
// A-operand policies.  Interface:
//   VEC            k elements one thread stages per load (4: one f32x4; 8: 8 bytes of u8 pixels)
//   Row            per-thread row context, row(x, m, M)
//   load(row, kt, q, v[VEC/4])   q = which VEC-group of the 32-wide k-tile
// ------------------------------------------------------------------------------------------------

// NHWC f32 input, forward patches: m=(b,oh,ow), k=(kh,kw,c).
template <class G>
struct AFwd {
    static constexpr int VEC = 4;
    static constexpr int NKT = G::K / BK;
    struct Row { const float* p; bool ok; };
    __device__ static Row row(const float* x, int m, int M)
    {
        Row r;
        r.ok = m < M;
        const int mm = r.ok ? m : 0;
        const int b = mm / (G::OH * G::OW), rem = mm % (G::OH * G::OW);
        const int oh = rem / G::OW, ow = rem % G::OW;
        r.p = x + ((size_t)(b * G::IH + oh * G::S) * G::IW + ow * G::S) * G::CIN;
        return r;
    }
    __device__ static void load(const Row& r, int kt, int q, f32x4* v)
    {
        constexpr int TPT = G::CIN / BK;  // k-tiles per tap
        const int tap = kt / TPT, c0 = (kt % TPT) * BK;
        const int kh = tap / G::KW, kw = tap % G::KW;
        const float* p = r.p + (kh * G::IW + kw) * G::CIN + c0 + q * 4;
        v[0] = *reinterpret_cast<const f32x4*>(p);   // rows >= M alias row 0 (never stored / zero Y row)
    }
    // address of the 16-byte chunk q of k-tile kt (direct-to-LDS staging)
    __device__ static const float* chunk(const Row& r, int kt, int q)
    {
        constexpr int TPT = G::CIN / BK;
        const int tap = kt / TPT, c0 = (kt % TPT) * BK;
        const int kh = tap / G::KW, kw = tap % G::KW;
        return r.p + (kh * G::IW + kw) * G::CIN + c0 + q * 4;
    }
};

// Input gradient of a stride-1 conv as a gather: rows m'=(b,ih,iw) over the conv INPUT grid,
// k'=(kh,kw,cout) over dY [B][OH][OW][COUT]; out-of-range taps read zero.
template <class G>
struct ADxS1 {
    static constexpr int VEC = 4;
    static constexpr int NKT = G::KH * G::KW * G::COUT / BK;
    struct Row { const float* dy; int b, ih, iw; bool ok; };
    __device__ static Row row(const float* dy, int m, int M)
    {
        Row r;
        r.ok = m < M; r.dy = dy;
        const int mm = r.ok ? m : 0;
        r.b = mm / (G::IH * G::IW);
        const int rem = mm % (G::IH * G::IW);
        r.ih = rem / G::IW; r.iw = rem % G::IW;
        return r;
    }
    __device__ static void load(const Row& r, int kt, int q, f32x4* v)
    {
        constexpr int TPT = G::COUT / BK;
        const int tap = kt / TPT, c0 = (kt % TPT) * BK;
        const int oh = r.ih - tap / G::KW, ow = r.iw - tap % G::KW;
        const bool ok = oh >= 0 && oh < G::OH && ow >= 0 && ow < G::OW;
        const int ohc = min(max(oh, 0), G::OH - 1), owc = min(max(ow, 0), G::OW - 1);   // branch-free: clamp + select
        const float* p = r.dy + ((size_t)(r.b * G::OH + ohc) * G::OW + owc) * G::COUT + c0 + q * 4;
        const f32x4 t = *reinterpret_cast<const f32x4*>(p);
        v[0] = ok ? t : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __device__ static const float* chunk(const Row& r, int kt, int q)   // out-of-range taps read the zero page
    {
        constexpr int TPT = G::COUT / BK;
        const int tap = kt / TPT, c0 = (kt % TPT) * BK;
        const int oh = r.ih - tap / G::KW, ow = r.iw - tap % G::KW;
        const bool ok = oh >= 0 && oh < G::OH && ow >= 0 && ow < G::OW;
        const float* p = r.dy + ((size_t)(r.b * G::OH + oh) * G::OW + ow) * G::COUT + c0 + q * 4;
        return ok ? p : zero_page();
    }
};

// The same gather for tiles whose rows all sit at ONE input position (ih, iw): only the taps that reach a valid output are
// walked (loop index t -> the t-th valid tap of the row's position), so border positions do 1-6 taps instead of 9 - 40 %
// of the stride-1 3x3 transposed conv's products over the full tap set are with zero padding.
template <class G>
struct ADxS1Pos {
    static constexpr int VEC = 4;
    static constexpr int NKT = G::KH * G::KW * G::COUT / BK;
    struct Row { const float* base; int ih, iw, kh0, kw0, nw; };
    __device__ static void taps(int i, int n_out, int k, int& k0, int& n) { k0 = max(0, i - (n_out - 1)); n = min(k - 1, i) - k0 + 1; }
    __device__ static int div_small(int t, int d) { return d == 1 ? t : (d == 2 ? t >> 1 : (t * 11) >> 5); }   // t < 16, d <= 3
    __device__ static Row row(const float* dy, int m, int M)
    {
        Row r;
        const int mm = m < M ? m : 0;
        const int b = mm / (G::IH * G::IW), rem = mm % (G::IH * G::IW);
        r.ih = rem / G::IW; r.iw = rem % G::IW;
        int nh;
        taps(r.ih, G::OH, G::KH, r.kh0, nh);
        taps(r.iw, G::OW, G::KW, r.kw0, r.nw);
        r.base = dy + (size_t)b * G::OH * G::OW * G::COUT;
        return r;
    }
    __device__ static const float* chunk(const Row& r, int kt, int q)
    {
        constexpr int TPT = G::COUT / BK;
        const int t = kt / TPT, c0 = (kt % TPT) * BK;
        const int th = div_small(t, r.nw);
        const int kh = r.kh0 + th, kw = r.kw0 + (t - th * r.nw);
        // clamped: rows of padding images and the clamped tail iterations of a shorter team stay inside the tensor
        const int oh = min(max(r.ih - kh, 0), G::OH - 1), ow = min(max(r.iw - kw, 0), G::OW - 1);
        return r.base + (size_t)(oh * G::OW + ow) * G::COUT + c0 + q * 4;
    }
    __device__ static void load(const Row& r, int kt, int q, f32x4* v) { v[0] = *reinterpret_cast<const f32x4*>(chunk(r, kt, q)); }
};

// Input gradient of a stride-2, 4x4 conv: the input grid splits into 4 parity classes (ih%2, iw%2);
// class (ph,pw) only sees taps kh in {ph, ph+2}, kw in {pw, pw+2}, so K' = 4*COUT per class and no
// MFMA work is spent on structurally-zero taps.  rows m' = (b, ih/2, iw/2) within a class.
template <class G>
struct ADxS2 {
    static constexpr int VEC = 4;
    static constexpr int NKT = 4 * G::COUT / BK;
    static constexpr int HH = G::IH / 2, WH = G::IW / 2;
    struct Row { const float* dy; int b, ihh, iwh; bool ok; };
    __device__ static Row row(const float* dy, int m, int M)
    {
        static_assert(G::S == 2 && G::KH == 4 && G::KW == 4 && G::IH % 2 == 0 && G::IW % 2 == 0, "c2 geometry");
        Row r;
        r.ok = m < M; r.dy = dy;
        const int mm = r.ok ? m : 0;
        r.b = mm / (HH * WH);
        const int rem = mm % (HH * WH);
        r.ihh = rem / WH; r.iwh = rem % WH;
        return r;
    }
    // tap t=(a,b2): kh = ph+2a, kw = pw+2*b2, oh = ihh-a, ow = iwh-b2
    __device__ static void load(const Row& r, int kt, int q, f32x4* v)
    {
        constexpr int TPT = G::COUT / BK;
        const int tap = kt / TPT, c0 = (kt % TPT) * BK;
        const int oh = r.ihh - (tap >> 1), ow = r.iwh - (tap & 1);
        const bool ok = oh >= 0 && oh < G::OH && ow >= 0 && ow < G::OW;
        const int ohc = min(max(oh, 0), G::OH - 1), owc = min(max(ow, 0), G::OW - 1);
        const float* p = r.dy + ((size_t)(r.b * G::OH + ohc) * G::OW + owc) * G::COUT + c0 + q * 4;
        const f32x4 t = *reinterpret_cast<const f32x4*>(p);
        v[0] = ok ? t : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __device__ static const float* chunk(const Row& r, int kt, int q)
    {
        constexpr int TPT = G::COUT / BK;
        const int tap = kt / TPT, c0 = (kt % TPT) * BK;
        const int oh = r.ihh - (tap >> 1), ow = r.iwh - (tap & 1);
        const bool ok = oh >= 0 && oh < G::OH && ow >= 0 && ow < G::OW;
        const float* p = r.dy + ((size_t)(r.b * G::OH + oh) * G::OW + ow) * G::COUT + c0 + q * 4;
        return ok ? p : zero_page();
    }
};

// The stride-2 gather for tiles whose rows all sit at ONE half-resolution position (ihh, iwh) (cnn_layers.hpp DxC2MPosP): only the taps
// (a, b2) that reach a valid output are walked - 2 x 2 in the interior, 1 x 2 / 2 x 1 on the first and last row / column, 1 at the corners:
// 324 tap-rows per image instead of 400 (19 % of the flat row tiling's products are with zero padding).
template <class G>
struct ADxS2Pos {
    static constexpr int VEC = 4;
    static constexpr int NKT = 4 * G::COUT / BK;
    static constexpr int HH = G::IH / 2, WH = G::IW / 2;
    struct Row { const float* base; int ihh, iwh, a0, b0, nb; };
    // valid taps along one axis at half-resolution coordinate i: a in {0, 1} with 0 <= i - a <= n_out - 1
    __device__ static void taps(int i, int n_out, int& k0, int& n) { k0 = max(0, i - (n_out - 1)); n = min(1, i) - k0 + 1; }
    __device__ static Row row(const float* dy, int m, int M)
    {
        static_assert(G::S == 2 && G::KH == 4 && G::KW == 4 && G::IH % 2 == 0 && G::IW % 2 == 0, "c2 geometry");
        Row r;
        const int mm = m < M ? m : 0;
        const int b = mm / (HH * WH), rem = mm % (HH * WH);
        r.ihh = rem / WH; r.iwh = rem % WH;
        int na;
        taps(r.ihh, G::OH, r.a0, na);
        taps(r.iwh, G::OW, r.b0, r.nb);
        r.base = dy + (size_t)b * G::OH * G::OW * G::COUT;
        return r;
    }
    __device__ static const float* chunk(const Row& r, int kt, int q)
    {
        constexpr int TPT = G::COUT / BK;
        const int t = kt / TPT, c0 = (kt % TPT) * BK;
        const int th = r.nb == 2 ? t >> 1 : t;
        const int a = r.a0 + th, b2 = r.b0 + (t - th * r.nb);
        // clamped: rows of padding images and the clamped tail iterations of a shorter team stay inside the tensor
        const int oh = min(max(r.ihh - a, 0), G::OH - 1), ow = min(max(r.iwh - b2, 0), G::OW - 1);
        return r.base + (size_t)(oh * G::OW + ow) * G::COUT + c0 + q * 4;
    }
    __device__ static void load(const Row& r, int kt, int q, f32x4* v) { v[0] = *reinterpret_cast<const f32x4*>(chunk(r, kt, q)); }
};

// Dense row-major f32 matrix with a runtime leading dimension (MLP layers of the Mlp / SAC / IQN nets).
struct DenseSrc { const float* p; int ld; };
struct ADense {
    static constexpr int VEC = 4;
    struct Row { const float* p; bool ok; };
    __device__ static Row row(DenseSrc s, int m, int M)
    {
        Row r;
        r.ok = m < M;
        r.p = s.p + (size_t)(r.ok ? m : 0) * s.ld;
        return r;
    }
    __device__ static void load(const Row& r, int kt, int q, f32x4* v)
    {
        v[0] = *reinterpret_cast<const f32x4*>(r.p + kt * BK + q * 4);
    }
    __device__ static const float* chunk(const Row& r, int kt, int q) { return r.p + kt * BK + q * 4; }
};

// Dense rows multiplied element-wise by a row of a second matrix on their way into LDS: A(m,k) = x[m][k] * h[m / group][k]
// (IQN's merge m = relu(phi) * psi(x)[b]: the product is an operand of two big GEMMs and is never written out).  The kernels
// load both vectors in the prefetch and multiply when the tile is committed to LDS (policies with HAD, see a_has_had).
struct DenseSrcHad { const float* p; int ld; const float* had; int had_ld, had_group; };
struct ADenseHad {
    static constexpr int VEC = 4;
    static constexpr bool HAD = true;
    struct Row { const float* p; const float* h; bool ok; };
    __device__ static Row row(DenseSrcHad s, int m, int M)
    {
        Row r;
        r.ok = m < M;
        const int mm = r.ok ? m : 0;
        r.p = s.p + (size_t)mm * s.ld;
        r.h = s.had + (size_t)(mm / s.had_group) * s.had_ld;
        return r;
    }
    __device__ static void load(const Row& r, int kt, int q, f32x4* v) { v[0] = *reinterpret_cast<const f32x4*>(r.p + kt * BK + q * 4); }
    __device__ static f32x4 load_had(const Row& r, int kt, int q) { return *reinterpret_cast<const f32x4*>(r.h + kt * BK + q * 4); }
};
template <class A, class = void> struct a_has_had : std::false_type {};
template <class A> struct a_has_had<A, std::void_t<decltype(A::HAD)>> : std::bool_constant<A::HAD> {};

// Keeps a wave-uniform pointer in an SGPR pair from here on (otherwise the compiler re-loads kernel-argument
// pointers inside every conditional store block of the epilogue: 16 scalar-load round trips per tile).
// The pinned pointer is typed as global address space so that accesses stay global_load / global_store
// (an opaque generic pointer would turn them into flat_* instructions, which also tick lgkmcnt).
template <class T>
using gptr = __attribute__((address_space(1))) T*;
template <class T>
__device__ __forceinline__ gptr<T> pin_sgpr(T* p)
{
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    asm volatile("" : "+s"(lo), "+s"(hi));
    return (gptr<T>)(((uint64_t)hi << 32) | lo);
}

// Row maps of k_igemm.  Flat: virtual row == real row.  Per-image: every image's RPI rows are padded to
// RPIP (a multiple of the tile height) so that tiles never straddle images and a B-image batch is
// B * RPIP / BM equal workgroups.
__device__ __forceinline__ bool vrow_flat(int M, int mv, int& mr) { mr = mv; return mv < M; }
template <int RPI, int RPIP>
__device__ __forceinline__ bool vrow_img(int M, int mv, int& mr)
{
    const int b = mv / RPIP, r = mv % RPIP;
    mr = b * RPI + r;
    return r < RPI && mr < M;
}
// policies opt into an XCD-aware block map with `static constexpr int XMAP` (see k_igemm)
template <class P, class = void> struct xmap_of { static constexpr int value = 0; };
template <class P> struct xmap_of<P, std::void_t<decltype(P::XMAP)>> { static constexpr int value = P::XMAP; };

// transposed-weight policies may map (y, tap, column) to a weight row themselves (`b_row`): the default is row tap_index(y, tap) * N + n
template <class P, class = void> struct has_b_row : std::false_type {};
template <class P> struct has_b_row<P, std::void_t<decltype(P::b_row(0, 0, 0))>> : std::true_type {};

// m-tiles of a launch: RPIP == 0 -> flat
template <class P>
inline int m_tiles(int M)
{
    constexpr int BM = P::WM * P::TM * 32;
    if constexpr (P::RPIP == 0) return (M + BM - 1) / BM;
    else return (M / P::RPI) * (P::RPIP / BM);
}

// ------------------------------------------------------------------------------------------------
// The MFMA core shared by both kernels: one k-tile (32 deep) from LDS.
//   As: [rows][LDA] f32, k contiguous.  Bs: [32][LDB] f32, n contiguous.
// ------------------------------------------------------------------------------------------------
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
            a[tm] = *reinterpret_cast<const f32x4*>(&As[(arow0 + tm * 32 + i) * LDA + 8 * u + 4 * h]);
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
                else b[tn] = Bs[(8 * u + 4 * h + s) * LDB + bcol0 + tn * 32 + i];
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

// ------------------------------------------------------------------------------------------------
// k_igemm: C = A(m,k) * B(k,n) with a policy P:
//   P::A (A-operand policy), P::WM, P::WN, P::TM, P::TN   (4 waves: WM*WN == 4)
//   P::B_TR    false: B(k,n) = w[k*ldw + n]  (vector loads along n)
//              true : B(k',n') = w[(tap*NP + n')*KP + c]  (transposed weights for dX; vector along k')
//   P::Args    kernel arguments;  device hooks:
//     a_src(args,z) -> input pointer;  M(args);  w(args,z,cls) -> weight pointer
//     kt_range(args, &kt0, &kt1)  (split-K over blockIdx.y when P::SPLITK)
//     epi(args, z, cls/split) -> Epi   (epilogue pointers, read from the kernel arguments once and pinned)
//     epi_load(epi, m, n) -> aux       (bias / mask operand of the epilogue, loaded before the k loop)
//     store(epi, m, n, value, aux)
// grid: x = m-tiles * n-tiles (n fastest), y = split or parity class, z = problem instance.
// ------------------------------------------------------------------------------------------------
// TEAMS = 2 runs two independent 4-wave teams in one 512-thread workgroup: team g stages and multiplies
// the k-tiles kt = g (mod 2) in its own LDS stages and the two accumulators are summed through LDS
// at the end.  Kernels whose grids give only ~1 workgroup per CU get 2 waves per SIMD this way (the
// staging of one team runs in the shadow of the other team's MFMAs) without more, smaller tiles.
// Virtual rows: a policy may pad every image's rows to a multiple of the tile height
// (P::vrow(args, mv, m_real) -> valid), so that one workgroup == one image and a B-image batch maps
// onto the 256 CUs without the round-robin tail of a flat row tiling (81 conv2 rows -> 96, 49 -> 64).
// Cross-queue progress flags (see DqnCnn::update_critic, schedule 3).  A kernel whose Args carry `sig_flag` / `sig_epoch`
// publishes "everything queued before me on my stream is complete" the moment its first workgroup starts: the dispatch
// packet's barrier bit has already waited for the predecessors and their end-of-kernel release, so a consumer on another
// queue that sees flag >= epoch (k_gate) and then starts a kernel (acquire) reads complete data - without any event /
// barrier packet in the producer's queue.
template <class A, class = void> struct has_start_signal : std::false_type {};
template <class A> struct has_start_signal<A, std::void_t<decltype(std::declval<A>().sig_flag)>> : std::true_type {};
__device__ inline void start_signal(unsigned* flag, unsigned epoch)
{
    if (flag && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0)
        __hip_atomic_store(flag, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// policies whose row map depends on blockIdx.y (position-class tiles) provide vrow_y(args, y, mv, mr)
template <class P, class = void> struct has_vrow_y : std::false_type {};
template <class P>
struct has_vrow_y<P, std::void_t<decltype(P::vrow_y(std::declval<const typename P::Args&>(), 0, 0, std::declval<int&>()))>> : std::true_type {};
template <class P>
__device__ inline bool vrow_of(const typename P::Args& a, int y, int mv, int& mr)
{
    if constexpr (has_vrow_y<P>::value) return P::vrow_y(a, y, mv, mr);
    else return P::vrow(a, mv, mr);
}

template <class P, int TEAMS = 1>
__global__ __launch_bounds__(64 * P::WM * P::WN * TEAMS) void k_igemm(typename P::Args args)
{
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
#ifdef BDR_IGEMM_ABL   // (tools/probes/dx_abl.hip) bit 0: no k loop, bit 1: no stores, bit 2: no epilogue operand loads
    constexpr int ABL = BDR_IGEMM_ABL;
#else
    constexpr int ABL = 0;
#endif
    const int iters = (ABL & 1) ? 0 : (kt1 - kt0 + TEAMS - 1) / TEAMS;
    const int my_n = (ABL & 1) ? 0 : (kt0 + team < kt1 ? (kt1 - kt0 - team + TEAMS - 1) / TEAMS : 0);
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
                aux[tm][tn][r] = (ABL & 4) ? 1.f : P::epi_load(epi, mrow[tm][r], n0 + (wn * P::TN + tn) * 32 + j);
        }
    }
    __syncthreads();
    int cur = 0;
    auto step = [&](auto set, int it) {   // set = (it+1)&1: holds tile it+1; refilled with tile it+3
        if (TEAMS == 1 || it < my_n) {    // team-uniform; one team: my_n == iters, and a branch here costs the compiler its knowledge of which loads are outstanding
            const float* As = smem + cur * STAGE;
            const int k3 = tile(it + 3);
            mfma_ktile<P::TM, P::TN, LDB, P::B_TR>(As, As + BM * LDA, wm * P::TM * 32, wn * P::TN * 32, lane, acc, [&](int u) {
                if (u == 0) { commit_a(set, cur ^ 1); commit_b(set, cur ^ 1); }   // tile it+1 -> idle stage
                else if (u == 1) prefetch_a(set, k3);                             // tile it+3 global loads
                else if (u == 2) prefetch_b(set, k3);
            });
        }
        __syncthreads();
        cur ^= 1;
    };
    // (pairs, then the odd tail: a conditional second step inside the loop gives the compiler a path "first step, straight back to the
    //  header" on which the set the header consumes is the most recently loaded one, and its s_waitcnt insertion then drains the loads of
    //  the previous step at every header - seen as vmcnt(3 .. 0) instead of vmcnt(7 .. 4) in the l1 forward kernel)
    {
        int it = 0;
        for (; it + 1 < iters; it += 2) {
            step(Set1{}, it);
            step(Set0{}, it + 1);
        }
        if (it < iters) step(Set1{}, it);
    }
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
                if ((okmask[tm] >> r & 1) && !((ABL & 2) && acc[tm][tn][r] != 12345.f)) P::store(epi, mrow[tm][r], n, acc[tm][tn][r], aux[tm][tn][r]);
        }
}


// flags: 0, or hipExtAnyOrderLaunch (no barrier bit on the dispatch packet: the kernel may start while the packets queued
// before it on the same stream are still running).  stop: event completed by this kernel's own dispatch packet (no separate
// marker packet in the queue, unlike hipEventRecord)
template <class P, int TEAMS>
inline hipError_t launch_igemm(hipStream_t st, dim3 grid, const typename P::Args& args, unsigned flags = 0, hipEvent_t stop = nullptr)
{
    hipExtLaunchKernelGGL((k_igemm<P, TEAMS>), grid, dim3(64 * P::WM * P::WN * TEAMS), 0, st, nullptr, stop, flags, args);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// k_igemm_red: G[k][n] = sum_{m in chunk} A(m,k) * Y(m,n).
//   P::A, P::WM, P::WN, P::TM, P::TN; KO_T = WM*TM*32 output rows, N_T = WN*TN*32 columns.
//   P::K(args) (total output rows), P::N(args) (columns, == N_T * n-tiles); compile-time
//   constants for the conv policies, runtime (padded to 32/64) for the dense policies
//   hooks: a_src(args), y_src(args), M(args), part(args, chunk) -> float* partial [K*N + N]
// grid: 1-D (see the XCD-aware map in the kernel).  The bias gradient (column sums of Y) is accumulated
// by the ko-tile-0 workgroups from the very Y tiles they stage.
// LDS: As[32 rows][KO_T (+4)] (read with lane = output row: consecutive floats), Ys[32][N_T].
// ------------------------------------------------------------------------------------------------
// body: workgroup `bx` of `gx` (the launch's 1-D grid, or this GEMM's slice of a grouped launch)
template <class P>
__device__ __forceinline__ void igemm_red_body(const typename P::Args& args, const int bx, const int gx)
{
    using A = typename P::A;
    constexpr int KO_T = P::WM * P::TM * 32, N_T = P::WN * P::TN * 32;
    constexpr int LDAR = KO_T + 4, LDY = N_T;
    constexpr int KSUB = KO_T / BK;                      // 32-wide k groups per tile row
    constexpr int ROWS_PER_PASS = 256 * A::VEC / BK;     // rows per staging pass for one k group
    constexpr int A_PASSES = (32 * KSUB) / ROWS_PER_PASS;  // (row, ksub) pairs / pass rows
    constexpr int AV = A::VEC / 4;
    constexpr int Y_VECS = 32 * N_T / 4 / 256;
    static_assert(P::WM * P::WN == 4, "4 waves");
    static_assert(A_PASSES >= 1, "A tile too small");
    static_assert(Y_VECS >= 1, "Y tile too small");

    constexpr int STAGE = 32 * LDAR + 32 * LDY;
    __shared__ __attribute__((aligned(16))) float smem[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / P::WN, wn = wave % P::WN;
    // XCD-aware block -> (tile, chunk) map.  Workgroup b is dispatched to XCD b % 8 and every XCD
    // has its own L2: all output tiles of one row chunk read the same A/Y rows, so they are given
    // to the same XCD (chunk c lives on XCD c % 8) and each XCD's L2 only ever sees 1/8 of the rows.
    // grid: 1-D, TILES * nchunks workgroups with nchunks % 8 == 0 (else the identity map).
    const int NT_N = P::N(args) / N_T;
    const int TILES = (P::K(args) / KO_T) * NT_N;
    const int nchunks = gx / TILES;
    int tile, chunk;
    if (nchunks % 8 == 0) {
        const int xcd = bx & 7, j = bx >> 3;
        chunk = (j / TILES) * 8 + xcd;
        tile = j % TILES;
    } else if (nchunks == 1 && (gx & 7) == 0) {
        // a single row chunk (l1's weight gradient: 256 rows, 49 x 8 tiles): give every XCD a contiguous run of tiles, n fastest,
        // i.e. ~1/8 of the A columns and all of Y - with the identity map XCD j ran n-tile j of every ko-tile and fetched ALL of
        // A (PMC, round 2: 25.7 MB for 3.7 MB of operands)
        chunk = 0;
        tile = (bx & 7) * (gx >> 3) + (bx >> 3);
    } else {
        chunk = bx / TILES;
        tile = bx % TILES;
    }
    const int kot = tile / NT_N, nt = tile % NT_N;
    const int ko0 = kot * KO_T, n0 = nt * N_T;
    const int M = P::M(args);
    const int n_mt = (M + 31) / 32;
    const int per = (n_mt + nchunks - 1) / nchunks;
    const int mt0 = chunk * per, mt1 = min(n_mt, mt0 + per);

    const int a_q = tid % (BK / A::VEC), a_r = tid / (BK / A::VEC);  // a_r in [0, ROWS_PER_PASS)
    const float* ysrc = P::y_src(args);

    // two register sets (as in k_igemm): the loads of row tile mt + 3 are issued while tile mt is on the matrix pipe and are consumed a
    // full tile later; with one set they were consumed six MFMAs (~400 cycles) after their issue and every tile waited out the rest of
    // the memory latency (vmcnt(1), vmcnt(0) behind the first MFMAs of each tile; the dW kernels had the lowest MFMA-busy of the step)
    f32x4 ra[2][A_PASSES][AV];
    constexpr bool HAD = a_has_had<A>::value;
    f32x4 rh[2][HAD ? A_PASSES : 1];
    f32x4 ry[2][Y_VECS];
    f32x4 bsum[Y_VECS];
#pragma unroll
    for (int v = 0; v < Y_VECS; ++v) bsum[v] = f32x4{0.f, 0.f, 0.f, 0.f};

    // pass p covers (row, ksub) pairs: idx = p*ROWS_PER_PASS + a_r; row = idx % 32, ksub = idx / 32
    auto prefetch_a = [&](auto set, int mt) {
        constexpr int S = decltype(set)::value;
#pragma unroll
        for (int p = 0; p < A_PASSES; ++p) {
            const int idx = p * ROWS_PER_PASS + a_r;
            const int row = idx % 32, ksub = idx / 32;
            typename A::Row r = A::row(P::a_src(args), mt * 32 + row, M);
            A::load(r, ko0 / BK + ksub, a_q, ra[S][p]);
            if constexpr (HAD) rh[S][p] = A::load_had(r, ko0 / BK + ksub, a_q);
        }
    };
    auto prefetch_y = [&](auto set, int mt) {   // rows >= M are clamped here and zeroed at the commit: a select behind the load makes the
                                                // compiler wait for it where the select is scheduled, half a tile after its issue
        constexpr int S = decltype(set)::value;
#pragma unroll
        for (int v = 0; v < Y_VECS; ++v) {
            const int e = tid + v * 256;
            const int row = e / (N_T / 4), n4 = e % (N_T / 4);
            const int m = mt * 32 + row;
            ry[S][v] = *reinterpret_cast<const f32x4*>(ysrc + (size_t)min(m, M - 1) * P::N(args) + n0 + n4 * 4);   // (masked at the commit)
        }
    };
    auto commit_a = [&](auto set, int stage) {
        constexpr int S = decltype(set)::value;
        float* As = smem + stage * STAGE;
#pragma unroll
        for (int p = 0; p < A_PASSES; ++p) {
            const int idx = p * ROWS_PER_PASS + a_r;
            const int row = idx % 32, ksub = idx / 32;
#pragma unroll
            for (int j = 0; j < AV; ++j) {
                f32x4 v = ra[S][p][j];
                if constexpr (HAD) v *= rh[S][p];
                *reinterpret_cast<f32x4*>(&As[row * LDAR + ksub * BK + a_q * A::VEC + j * 4]) = v;
            }
        }
    };
    auto commit_y = [&](auto set, int stage, bool count, int mt) {   // mt: the row tile the set holds
        constexpr int S = decltype(set)::value;
        float* Ys = smem + stage * STAGE + 32 * LDAR;
#pragma unroll
        for (int v = 0; v < Y_VECS; ++v) {
            const int e = tid + v * 256;
            const int row = e / (N_T / 4), n4 = e % (N_T / 4);
            const f32x4 t = mt * 32 + row < M ? ry[S][v] : f32x4{0.f, 0.f, 0.f, 0.f};   // rows >= M contribute zero
            *reinterpret_cast<f32x4*>(&Ys[row * LDY + n4 * 4]) = t;
            if (count) bsum[v] += t;
        }
    };

    f32x16 acc[P::TM][P::TN];
#pragma unroll
    for (int tm = 0; tm < P::TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < P::TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

    const int i = lane & 31, h = lane >> 5;
    // branch-free software pipeline (see k_igemm); the tail re-stages the last tile into the idle
    // stage, which must not be counted twice in the bias sums
    using Set0 = std::integral_constant<int, 0>;
    using Set1 = std::integral_constant<int, 1>;
    auto tile_mt = [&](int k) { return min(mt0 + k, mt1 - 1); };
    if (mt0 < mt1) {
        prefetch_a(Set0{}, mt0); prefetch_y(Set0{}, mt0);
        prefetch_a(Set1{}, tile_mt(1)); prefetch_y(Set1{}, tile_mt(1));
        commit_a(Set0{}, 0); commit_y(Set0{}, 0, true, mt0);
        prefetch_a(Set0{}, tile_mt(2)); prefetch_y(Set0{}, tile_mt(2));
    }
    __syncthreads();
    int cur = 0;
    auto step = [&](auto set, int mt) {   // set holds row tile mt + 1 (committed into the idle stage) and is refilled with tile mt + 3
        const float* As = smem + cur * STAGE;
        const float* Ys = As + 32 * LDAR;
        const int m3 = min(mt + 3, mt1 - 1);
        const bool fresh = mt + 1 < mt1;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int red = 2 * t + h;
            float a[P::TM], b[P::TN];
#pragma unroll
            for (int tm = 0; tm < P::TM; ++tm) a[tm] = As[red * LDAR + (wm * P::TM + tm) * 32 + i];
#pragma unroll
            for (int tn = 0; tn < P::TN; ++tn) b[tn] = Ys[red * LDY + (wn * P::TN + tn) * 32 + i];
#pragma unroll
            for (int tm = 0; tm < P::TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < P::TN; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm], b[tn], acc[tm][tn], 0, 0, 0);
            if (t == 1) commit_a(set, cur ^ 1);               // staging sliced into the MFMA shadows
            else if (t == 5) commit_y(set, cur ^ 1, fresh, min(mt + 1, mt1 - 1));
            else if (t == 8) prefetch_a(set, m3);
            else if (t == 11) prefetch_y(set, m3);
        }
        __syncthreads();
        cur ^= 1;
    };
    {
        int mt = mt0;
        for (; mt + 1 < mt1; mt += 2) {
            step(Set1{}, mt);
            step(Set0{}, mt + 1);
        }
        if (mt < mt1) step(Set1{}, mt);
    }

    float* part = P::part(args, chunk);
#pragma unroll
    for (int tm = 0; tm < P::TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < P::TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ko = ko0 + (wm * P::TM + tm) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const int n = n0 + (wn * P::TN + tn) * 32 + i;
                part[(size_t)ko * P::N(args) + n] = acc[tm][tn][r];
            }

    // bias gradient: column sums of the Y rows this chunk staged (ko-tile 0 only)
    if (kot == 0) {
        float* red = smem;  // reuse: [32 rows][N_T]
#pragma unroll
        for (int v = 0; v < Y_VECS; ++v) {
            const int e = tid + v * 256;
            const int row = e / (N_T / 4), n4 = e % (N_T / 4);
            *reinterpret_cast<f32x4*>(&red[row * N_T + n4 * 4]) = bsum[v];
        }
        __syncthreads();
        if (tid < N_T) {
            float s = 0.f;
#pragma unroll 8
            for (int row = 0; row < 32; ++row) s += red[row * N_T + tid];
            part[(size_t)P::K(args) * P::N(args) + n0 + tid] = s;
        }
    }
}

template <class P>
__global__ __launch_bounds__(256) void k_igemm_red(typename P::Args args)
{
    if constexpr (has_start_signal<typename P::Args>::value) start_signal(args.sig_flag, args.sig_epoch);
    igemm_red_body<P>(args, (int)blockIdx.x, (int)gridDim.x);
}

// Up to NG independent weight-gradient GEMMs of one policy (different shapes, operands and chunk counts) in ONE launch: the
// 1-D grid is the concatenation of their grids, first[g] = first workgroup of GEMM g, first[n] = the grid size.  For the
// launch-bound dense agents, where a step is a chain of 2-5 us kernels and every kernel boundary costs as much as the kernel.
template <class Args, int NG>
struct IgemmRedGroup { Args a[NG]; int first[NG + 1]; int n; };
template <class P, int NG>
__global__ __launch_bounds__(256) void k_igemm_red_group(IgemmRedGroup<typename P::Args, NG> g)
{
    const int b = (int)blockIdx.x;
    int z = 0;
#pragma unroll
    for (int k = 1; k < NG; ++k) z += (k < g.n && b >= g.first[k]) ? 1 : 0;
    igemm_red_body<P>(g.a[z], b - g.first[z], g.first[z + 1] - g.first[z]);
}

}  // namespace bdr
