// Nature-CNN layer policies shared by the DQN and IQN agents (border-tch-agent/src/cnn/base.rs:23-36):
// parameter arena layout, forward / input-gradient / weight-gradient policies for the implicit-GEMM
// kernels of igemm.hpp, the deterministic partial reduction, and the dW chunk plan.
#pragma once
#include <algorithm>

#include "agent_base.hpp"
#include "conv1_dw_bf16.hpp"
#include "igemm.hpp"

namespace {

constexpr int MAXZ = 3;       // network instances per forward launch
#ifndef BDR_L1_XSPLIT
#define BDR_L1_XSPLIT 1   // l1 forward: split-K slice = XCD (FwdL1X); 0: the round-2/3 maps (FwdL1 / FwdL1Z2, split 7)
#endif
#ifndef BDR_L1_SPLIT
#define BDR_L1_SPLIT (BDR_L1_XSPLIT ? 8 : 7)
#endif
constexpr int L1_SPLIT = BDR_L1_SPLIT;   // split-K of the 3136-deep l1 contraction (98 k-tiles)
constexpr float INV255 = 1.0f / 255.0f;

// ---- flat parameter arena (internal layouts; every segment 16-byte aligned) ----------------------
//   W1 [64 * n_stack][32]  k=(c,kh,kw)      b1[32]      (n_stack = 4: [256][32])
//   W2 [512][64]  k=(kh,kw,c)      b2[64]
//   W3 [576][64]  k=(kh,kw,c)      b3[64]
//   W4 [3136][512] k=(h,w,c)       b4[512]      (NHWC flatten of conv3's output)
//   W5 [A][512]  (= the reference's [out][in]: k_head reads a lane's 8 columns of an action as two f32x4)   b5[A]
struct Arena {
    size_t w1, b1, w2, b2, w3, b3, w4, b4, w5, b5, total;  // offsets in floats
    int A;
    int ns;        // AtariCnnConfig::n_stack (cnn/config.rs:14-24): conv1 has 64 * ns rows
    size_t n_w1() const { return (size_t)64 * ns * 32; }
};
Arena make_arena(int A, int ns = 4)
{
    Arena a{};
    size_t o = 0;
    auto seg = [&](size_t n) { size_t r = o; o += (n + 3) / 4 * 4; return r; };
    a.ns = ns;
    a.w1 = seg((size_t)64 * ns * 32); a.b1 = seg(32);
    a.w2 = seg(512 * 64); a.b2 = seg(64);
    a.w3 = seg(576 * 64); a.b3 = seg(64);
    a.w4 = seg((size_t)3136 * 512); a.b4 = seg(512);
    a.w5 = seg((size_t)512 * A); a.b5 = seg(A);
    a.total = o; a.A = A;
    return a;
}

// ================================================================================================
// forward policies
// ================================================================================================
struct FwdArgs {
    const void* x[MAXZ];     // layer input per instance
    const float* w[MAXZ];    // weights [K][N]
    const float* bias[MAXZ];
    float* out[MAXZ];        // [M][N] (or split-K partials [S][M][N])
    int M;
    int nkt_per_split;
};

template <class G, class APolicy, int WM_, int WN_, bool U8SCALE, int RPIP_ = 0, int TM_ = 1, int TN_ = 1>
struct FwdP {
    using A = APolicy;
    using Args = FwdArgs;
    static constexpr int WM = WM_, WN = WN_, TM = TM_, TN = TN_;
    static constexpr int RPI = G::OH * G::OW, RPIP = RPIP_;   // rows per image / padded (0: flat rows)
    static constexpr int NC = G::COUT;
    __device__ static bool vrow(const Args& a, int mv, int& mr)
    {
        if constexpr (RPIP == 0) return vrow_flat(a.M, mv, mr);
        else return vrow_img<RPI, RPIP>(a.M, mv, mr);
    }
    static constexpr bool B_TR = false;
    __device__ static constexpr int N(const Args&) { return NC; }
    __device__ static constexpr int KP(const Args&) { return 32; }
    __device__ static int M(const Args& a) { return a.M; }
    __device__ static auto a_src(const Args& a, int z)
    {
        if constexpr (U8SCALE) return reinterpret_cast<const uint8_t*>(a.x[z]);
        else return reinterpret_cast<const float*>(a.x[z]);
    }
    __device__ static const float* w(const Args& a, int z, int) { return a.w[z]; }
    __device__ static int tap_index(int, int t) { return t; }
    __device__ static void kt_range(const Args&, int, int& k0, int& k1) { k0 = 0; k1 = A::NKT; }
    struct Epi { gptr<const float> bias; gptr<float> out; };
    __device__ static Epi epi(const Args& a, int z, int) { return Epi{pin_sgpr(a.bias[z]), pin_sgpr(a.out[z])}; }
    __device__ static float epi_load(const Epi& e, int, int n) { return e.bias[n]; }
    __device__ static void store(const Epi& e, int m, int n, float v, float bias)
    {
        if constexpr (U8SCALE) v *= INV255;   // cnn/base.rs:26 "/ 255", folded into the epilogue
        v += bias;
        e.out[(size_t)m * NC + n] = v > 0.f ? v : 0.f;   // relu (cnn/base.rs:28,30,32)
    }
};
// (conv1 runs on the bf16 matrix cores with exact operands: conv1_bf16.hpp forward, conv1_dw_bf16.hpp weight gradient)
#ifndef BDR_FWDC2_SHAPE
#define BDR_FWDC2_SHAPE 2, 2, 1, 1
#endif
template <class G, int WM, int WN, int TM, int TN> using FwdConvP = FwdP<G, AFwd<G>, WM, WN, false, 0, TM, TN>;
using FwdC2 = FwdConvP<GeomC2, BDR_FWDC2_SHAPE>;     // 64x64,  M = B*81
using FwdC3 = FwdConvP<GeomC3, 2, 2, 1, 1>;          // 64x64,  M = B*49

// l1: [B][3136] x [3136][512], split-K over blockIdx.y, raw partials (bias/relu in the head kernel)
#ifndef BDR_FWDL1_SHAPE
#define BDR_FWDL1_SHAPE 2, 2, 1, 1
#endif
template <int WM_, int WN_, int TM_, int TN_>
struct FwdL1P {
    using A = AFwd<GeomL1>;
    using Args = FwdArgs;
    static constexpr int WM = WM_, WN = WN_, TM = TM_, TN = TN_;
    static constexpr int RPI = 1, RPIP = 0;
    static constexpr int NC = 512;
    static constexpr bool B_TR = false;
    __device__ static bool vrow(const Args& a, int mv, int& mr) { return vrow_flat(a.M, mv, mr); }
    __device__ static constexpr int N(const Args&) { return NC; }
    __device__ static constexpr int KP(const Args&) { return 32; }
    __device__ static int M(const Args& a) { return a.M; }
    __device__ static const float* a_src(const Args& a, int z) { return reinterpret_cast<const float*>(a.x[z]); }
    __device__ static const float* w(const Args& a, int z, int) { return a.w[z]; }
    __device__ static int tap_index(int, int t) { return t; }
    __device__ static void kt_range(const Args& a, int y, int& k0, int& k1)
    {
        k0 = y * a.nkt_per_split; k1 = min(A::NKT, k0 + a.nkt_per_split);
    }
    struct Epi { gptr<float> out; };
    __device__ static Epi epi(const Args& a, int z, int y) { return Epi{pin_sgpr(a.out[z] + (size_t)y * a.M * NC)}; }
    __device__ static float epi_load(const Epi&, int, int) { return 0.f; }
    __device__ static void store(const Epi& e, int m, int n, float v, float) { e.out[(size_t)m * NC + n] = v; }
};
using FwdL1 = FwdL1P<BDR_FWDL1_SHAPE>;
// even instance counts (the online + target pair): XCD = (instance parity, pair of n-tiles); grid (16 m-tiles, splits, nz/2)
struct FwdL1Z2 : FwdL1 { static constexpr int XMAP = 2; };
// eight k-slices, slice = XCD (98 k-tiles: 12 or 13 per slice); grid (8 * m-tiles * 8 n-tiles, 1, nz); L1_SPLIT == 8
struct FwdL1X : FwdL1 {
    static constexpr int XMAP = 3;
    __device__ static void kt_range(const Args&, int y, int& k0, int& k1) { k0 = (y * A::NKT) >> 3; k1 = ((y + 1) * A::NKT) >> 3; }
};

// ================================================================================================
// input-gradient policies (transposed conv as gather; epilogue applies relu'(previous activation))
// ================================================================================================
struct DxArgs {
    const float* dy;     // gradient w.r.t. this layer's pre-activation output
    const float* w;      // this layer's weights [K][N]
    const float* mask;   // previous layer's post-relu activation (same shape as out)
    float* out;          // gradient w.r.t. the previous layer's pre-activation
    int M;               // rows (per parity class for stride 2)
    unsigned* sig_flag;  // optional progress flag written at kernel start (igemm.hpp start_signal)
    unsigned sig_epoch;
};

// l1: dh0[b][k] = sum_n dh1[b][n] W4[k][n];  treated as 1x1 "conv" with CIN=512 -> N'=3136
#ifndef BDR_DXL1_SHAPE
#define BDR_DXL1_SHAPE 2, 2, 1, 1
#endif
template <int WM_, int WN_, int TM_, int TN_>
struct DxL1P {
    using G = Geom<1, 1, 512, 1, 1, 1, 1, 1, 3136>;
    using A = AFwd<G>;     // dense rows of dh1
    using Args = DxArgs;
    static constexpr int WM = WM_, WN = WN_, TM = TM_, TN = TN_;
    static constexpr int RPI = 1, RPIP = 0;
    static constexpr int NC = 3136;    // N' (columns of the result)
    static constexpr bool B_TR = true;
    static constexpr int XMAP = 1;     // the 4 m-tiles of a column tile share one XCD's L2 (grid.x = 8 * ceil(tiles / 8))
    __device__ static bool vrow(const Args& a, int mv, int& mr) { return vrow_flat(a.M, mv, mr); }
    __device__ static constexpr int N(const Args&) { return NC; }
    __device__ static constexpr int KP(const Args&) { return 512; }   // K' per tap (contiguous in memory)
    __device__ static int M(const Args& a) { return a.M; }
    __device__ static const float* a_src(const Args& a, int) { return a.dy; }
    __device__ static const float* w(const Args& a, int, int) { return a.w; }
    __device__ static int tap_index(int, int t) { return t; }
    __device__ static void kt_range(const Args&, int, int& k0, int& k1) { k0 = 0; k1 = A::NKT; }
    struct Epi { gptr<const float> mask; gptr<float> out; };
    __device__ static Epi epi(const Args& a, int, int) { return Epi{pin_sgpr(a.mask), pin_sgpr(a.out)}; }
    __device__ static float epi_load(const Epi& e, int m, int n) { return e.mask[(size_t)m * NC + n]; }
    __device__ static void store(const Epi& e, int m, int n, float v, float mask)
    {
        e.out[(size_t)m * NC + n] = mask > 0.f ? v : 0.f;
    }
};

using DxL1 = DxL1P<BDR_DXL1_SHAPE>;

// column tiles of a launch
template <class P> constexpr int n_tiles() { return P::NC / (P::WN * P::TN * 32); }

// conv3 (3x3, stride 1): rows over the 9x9 input grid, K' = 9 taps * 64, N' = 64
template <int WM_, int WN_, int RPIP_ = 0, int TM_ = 1, int TN_ = 1>
struct DxC3P {
    using G = GeomC3;
    using A = ADxS1<G>;
    using Args = DxArgs;
    static constexpr int WM = WM_, WN = WN_, TM = TM_, TN = TN_;
    static constexpr int RPI = G::IH * G::IW, RPIP = RPIP_;
    static constexpr int NC = G::CIN;
    static constexpr bool B_TR = true;
    __device__ static bool vrow(const Args& a, int mv, int& mr)
    {
        if constexpr (RPIP == 0) return vrow_flat(a.M, mv, mr);
        else return vrow_img<RPI, RPIP>(a.M, mv, mr);
    }
    __device__ static constexpr int N(const Args&) { return NC; }
    __device__ static constexpr int KP(const Args&) { return G::COUT; }
    __device__ static int M(const Args& a) { return a.M; }
    __device__ static const float* a_src(const Args& a, int) { return a.dy; }
    __device__ static const float* w(const Args& a, int, int) { return a.w; }
    __device__ static int tap_index(int, int t) { return t; }
    __device__ static void kt_range(const Args&, int, int& k0, int& k1) { k0 = 0; k1 = A::NKT; }
    struct Epi { gptr<const float> mask; gptr<float> out; };
    __device__ static Epi epi(const Args& a, int, int) { return Epi{pin_sgpr(a.mask), pin_sgpr(a.out)}; }
    __device__ static float epi_load(const Epi& e, int m, int n) { return e.mask[(size_t)m * NC + n]; }
    __device__ static void store(const Epi& e, int m, int n, float v, float mask)
    {
        e.out[(size_t)m * NC + n] = mask > 0.f ? v : 0.f;
    }
};
using DxC3 = DxC3P<2, 2>;      // 64x64 flat tiles, M = B*81

// Position-class tiles: workgroup (x, y) = 64 images at input position pos(y); its k loop walks only the taps that reach a
// valid output from that position (1, 2, 3, 4, 6 or 9 of them) - 441 tap-tiles per 64 images instead of 729.  y enumerates
// the positions by decreasing tap count so that the long tiles are dispatched first.  grid: (ceil(B / 64), 81, 1).
static __device__ __constant__ unsigned char c3_pos_by_taps[81] = {
    20, 21, 22, 23, 24, 29, 30, 31, 32, 33, 38, 39, 40, 41, 42, 47, 48, 49, 50, 51, 56, 57, 58, 59, 60, 11, 12, 13, 14, 15, 19, 25, 28, 34, 37, 43, 46,
    52, 55, 61, 65, 66, 67, 68, 69, 10, 16, 64, 70, 2, 3, 4, 5, 6, 18, 26, 27, 35, 36, 44, 45, 53, 54, 62, 74, 75, 76, 77, 78, 1, 7, 9, 17, 63, 71, 73,
    79, 0, 8, 72, 80};
#ifndef BDR_DXC3_SHAPE
#define BDR_DXC3_SHAPE 2, 2, 1, 1
#endif
template <int WM_, int WN_, int TM_, int TN_>
struct DxC3PosP {
    using G = GeomC3;
    using A = ADxS1Pos<G>;
    using Args = DxArgs;          // M = B * 81
    static constexpr int WM = WM_, WN = WN_, TM = TM_, TN = TN_;
    static constexpr int RPI = 1, RPIP = 0;
    static constexpr int NC = G::CIN;
    static constexpr bool B_TR = true;
    __device__ static bool vrow(const Args& a, int mv, int& mr) { return vrow_flat(a.M, mv, mr); }   // (unused: vrow_y)
    __device__ static bool vrow_y(const Args& a, int y, int mv, int& mr)
    {
        mr = mv * (G::IH * G::IW) + c3_pos_by_taps[y];     // image mv at this workgroup's position
        return mr < a.M;
    }
    __device__ static constexpr int N(const Args&) { return NC; }
    __device__ static constexpr int KP(const Args&) { return G::COUT; }
    __device__ static int M(const Args& a) { return a.M; }
    __device__ static const float* a_src(const Args& a, int) { return a.dy; }
    __device__ static const float* w(const Args& a, int, int) { return a.w; }
    __device__ static void pos_taps(int y, int& kh0, int& nh, int& kw0, int& nw)
    {
        const int pos = c3_pos_by_taps[y];
        A::taps(pos / G::IW, G::OH, G::KH, kh0, nh);
        A::taps(pos % G::IW, G::OW, G::KW, kw0, nw);
    }
    __device__ static int tap_index(int y, int t)
    {
        int kh0, nh, kw0, nw;
        pos_taps(y, kh0, nh, kw0, nw);
        const int th = A::div_small(min(t, nh * nw - 1), nw);
        return (kh0 + th) * G::KW + kw0 + (min(t, nh * nw - 1) - th * nw);
    }
    __device__ static void kt_range(const Args&, int y, int& k0, int& k1)
    {
        int kh0, nh, kw0, nw;
        pos_taps(y, kh0, nh, kw0, nw);
        k0 = 0; k1 = nh * nw * (G::COUT / BK);
    }
    struct Epi { gptr<const float> mask; gptr<float> out; };
    __device__ static Epi epi(const Args& a, int, int) { return Epi{pin_sgpr(a.mask), pin_sgpr(a.out)}; }
    __device__ static float epi_load(const Epi& e, int m, int n) { return e.mask[(size_t)m * NC + n]; }
    __device__ static void store(const Epi& e, int m, int n, float v, float mask) { e.out[(size_t)m * NC + n] = mask > 0.f ? v : 0.f; }
};

using DxC3Pos = DxC3PosP<BDR_DXC3_SHAPE>;

// conv2 (4x4, stride 2): blockIdx.y = parity class (ph,pw); rows (b, ih/2, iw/2); K' = 4 taps * 64
template <int WM_, int WN_, int RPIP_ = 0, int TM_ = 1, int TN_ = 1>
struct DxC2P {
    using G = GeomC2;
    using A = ADxS2<G>;
    using Args = DxArgs;
    static constexpr int WM = WM_, WN = WN_, TM = TM_, TN = TN_;
    static constexpr int RPI = (G::IH / 2) * (G::IW / 2), RPIP = RPIP_;
    static constexpr int NC = G::CIN;
    static constexpr bool B_TR = true;
    __device__ static bool vrow(const Args& a, int mv, int& mr)
    {
        if constexpr (RPIP == 0) return vrow_flat(a.M, mv, mr);
        else return vrow_img<RPI, RPIP>(a.M, mv, mr);
    }
    __device__ static constexpr int N(const Args&) { return NC; }
    __device__ static constexpr int KP(const Args&) { return G::COUT; }
    __device__ static int M(const Args& a) { return a.M; }
    __device__ static const float* a_src(const Args& a, int) { return a.dy; }
    __device__ static const float* w(const Args& a, int, int) { return a.w; }
    // class y=(ph,pw), tap t=(a,b2) -> (kh,kw) = (ph+2a, pw+2*b2) -> kh*4+kw
    __device__ static int tap_index(int y, int t) { return ((y >> 1) + 2 * (t >> 1)) * 4 + (y & 1) + 2 * (t & 1); }
    __device__ static void kt_range(const Args&, int, int& k0, int& k1) { k0 = 0; k1 = A::NKT; }
    __device__ static size_t out_index(int y, int m, int n)
    {
        constexpr int HH = G::IH / 2, WH = G::IW / 2;
        const int b = m / (HH * WH), rem = m % (HH * WH);
        const int ih = 2 * (rem / WH) + (y >> 1), iw = 2 * (rem % WH) + (y & 1);
        return ((size_t)(b * G::IH + ih) * G::IW + iw) * NC + n;
    }
    struct Epi { gptr<const float> mask; gptr<float> out; int y; };
    __device__ static Epi epi(const Args& a, int, int y) { return Epi{pin_sgpr(a.mask), pin_sgpr(a.out), y}; }
    __device__ static float epi_load(const Epi& e, int m, int n) { return e.mask[out_index(e.y, m, n)]; }
    __device__ static void store(const Epi& e, int m, int n, float v, float mask)
    {
        e.out[out_index(e.y, m, n)] = mask > 0.f ? v : 0.f;
    }
};
using DxC2 = DxC2P<4, 1>;      // 128x32 flat tiles per class, M = B*100

// The four parity classes as ONE GEMM.  Row (b, ih/2, iw/2) of class (ph, pw) reads dY[b][ih/2 - a][iw/2 - b2][:] for tap (a, b2):
// the A operand does not depend on the class at all - only the weight row does (kh = ph + 2a, kw = pw + 2*b2).  So the classes
// are 4 x 32 = 128 COLUMNS of one [B*100][256] x [256][128] product: a dY tile is staged once instead of four times (PMC, round 3:
// the per-class launch fetched 20.1 MB for 5.4 MB of dY) and every A fragment read from LDS feeds both classes of its wave.
// Per output element the k order (tap, cout) is that of DxC2P: bit-identical results.
template <int WM_, int WN_, int TM_, int TN_>
struct DxC2MP {
    using G = GeomC2;
    using A = ADxS2<G>;
    using Args = DxArgs;          // M = B * 100 (rows of ONE class)
    static constexpr int WM = WM_, WN = WN_, TM = TM_, TN = TN_;
    static constexpr int RPI = (G::IH / 2) * (G::IW / 2), RPIP = 0;
    static constexpr int NC = 4 * G::CIN;          // columns n = class * 32 + cin
    static constexpr bool B_TR = true;
    __device__ static bool vrow(const Args& a, int mv, int& mr) { return vrow_flat(a.M, mv, mr); }
    __device__ static constexpr int N(const Args&) { return NC; }
    __device__ static constexpr int KP(const Args&) { return G::COUT; }
    __device__ static int M(const Args& a) { return a.M; }
    __device__ static const float* a_src(const Args& a, int) { return a.dy; }
    __device__ static const float* w(const Args& a, int, int) { return a.w; }
    __device__ static int tap_index(int cls, int t) { return ((cls >> 1) + 2 * (t >> 1)) * 4 + (cls & 1) + 2 * (t & 1); }
    // weight row of column n for tap t: W2 is [(kh,kw,cin)][cout]
    __device__ static int b_row(int, int t, int n) { return tap_index(n >> 5, t) * G::CIN + (n & 31); }
    __device__ static void kt_range(const Args&, int, int& k0, int& k1) { k0 = 0; k1 = A::NKT; }
    __device__ static size_t out_index(int m, int n)
    {
        constexpr int HH = G::IH / 2, WH = G::IW / 2;
        const int cls = n >> 5;
        const int b = m / (HH * WH), rem = m % (HH * WH);
        const int ih = 2 * (rem / WH) + (cls >> 1), iw = 2 * (rem % WH) + (cls & 1);
        return ((size_t)(b * G::IH + ih) * G::IW + iw) * G::CIN + (n & 31);
    }
    struct Epi { gptr<const float> mask; gptr<float> out; };
    __device__ static Epi epi(const Args& a, int, int) { return Epi{pin_sgpr(a.mask), pin_sgpr(a.out)}; }
    __device__ static float epi_load(const Epi& e, int m, int n) { return e.mask[out_index(m, n)]; }
    __device__ static void store(const Epi& e, int m, int n, float v, float mask) { e.out[out_index(m, n)] = mask > 0.f ? v : 0.f; }
};
#ifndef BDR_DXC2M_SHAPE
#define BDR_DXC2M_SHAPE 2, 2, 1, 2     // 64 x 128 tiles: 400 workgroups at B = 256
#endif
using DxC2M = DxC2MP<BDR_DXC2M_SHAPE>;

// The merged-class GEMM with POSITION-CLASS tiles (round 6; conv3's DxC3PosP idea on the stride-2 layer): workgroup (x, y) = 64 images at the
// half-resolution position pos(y) = (ihh, iwh), all four parity classes (128 columns).  Its k loop walks only the taps that reach a valid
// output from that position: 4 in the interior (64 positions), 2 on the border rows / columns (32), 1 at the corners (4) - 2 592 k-tile
// units of 64 x 128 x 32 per 256 images instead of 3 200, and the longest workgroups (8 k-tiles) are exactly 256 = one per CU at B = 256;
// the 144 short ones (4 and 2 k-tiles) run beside them.  The flat row tiling had 400 equal workgroups of 8 k-tiles: 144 CUs held two,
// 112 idled through half of the k loop.  y enumerates the positions by decreasing tap count so that the long tiles are dispatched first.
// Per output element the products are those of DxC2MP minus the ones with a zero-padding operand, in the same order: bit-identical results.
// grid: (ceil(B / 64), 100, 1).
static __device__ __constant__ unsigned char c2_pos_by_taps[100] = {
    11, 12, 13, 14, 15, 16, 17, 18, 21, 22, 23, 24, 25, 26, 27, 28, 31, 32, 33, 34, 35, 36, 37, 38, 41, 42, 43, 44, 45, 46, 47, 48, 51, 52, 53, 54, 55,
    56, 57, 58, 61, 62, 63, 64, 65, 66, 67, 68, 71, 72, 73, 74, 75, 76, 77, 78, 81, 82, 83, 84, 85, 86, 87, 88, 1, 2, 3, 4, 5, 6, 7, 8, 10, 19, 20, 29,
    30, 39, 40, 49, 50, 59, 60, 69, 70, 79, 80, 89, 91, 92, 93, 94, 95, 96, 97, 98, 0, 9, 90, 99};
template <int WM_, int WN_, int TM_, int TN_>
struct DxC2MPosP {
    using G = GeomC2;
    using A = ADxS2Pos<G>;
    using Args = DxArgs;          // M = B * 100 (rows of ONE class)
    static constexpr int WM = WM_, WN = WN_, TM = TM_, TN = TN_;
    static constexpr int RPI = 1, RPIP = 0;
    static constexpr int NC = 4 * G::CIN;          // columns n = class * 32 + cin
    static constexpr bool B_TR = true;
    static constexpr int NPOS = A::HH * A::WH;
    __device__ static bool vrow(const Args& a, int mv, int& mr) { return vrow_flat(a.M, mv, mr); }   // (unused: vrow_y)
    __device__ static bool vrow_y(const Args& a, int y, int mv, int& mr)
    {
        mr = mv * NPOS + c2_pos_by_taps[y];     // image mv at this workgroup's position
        return mr < a.M;
    }
    __device__ static constexpr int N(const Args&) { return NC; }
    __device__ static constexpr int KP(const Args&) { return G::COUT; }
    __device__ static int M(const Args& a) { return a.M; }
    __device__ static const float* a_src(const Args& a, int) { return a.dy; }
    __device__ static const float* w(const Args& a, int, int) { return a.w; }
    __device__ static void pos_taps(int y, int& a0, int& na, int& b0, int& nb)
    {
        const int pos = c2_pos_by_taps[y];
        A::taps(pos / A::WH, G::OH, a0, na);
        A::taps(pos % A::WH, G::OW, b0, nb);
    }
    __device__ static int tap_index(int, int t) { return t; }   // (unused: b_row)
    // weight row of column n = (class, cin) for the t-th VALID tap of position y: W2 is [(kh,kw,cin)][cout], kh = ph + 2a, kw = pw + 2*b2
    __device__ static int b_row(int y, int t, int n)
    {
        int a0, na, b0, nb;
        pos_taps(y, a0, na, b0, nb);
        const int tc = min(t, na * nb - 1);
        const int th = nb == 2 ? tc >> 1 : tc;
        const int a = a0 + th, b2 = b0 + (tc - th * nb), cls = n >> 5;
        return (((cls >> 1) + 2 * a) * 4 + (cls & 1) + 2 * b2) * G::CIN + (n & 31);
    }
    __device__ static void kt_range(const Args&, int y, int& k0, int& k1)
    {
        int a0, na, b0, nb;
        pos_taps(y, a0, na, b0, nb);
        k0 = 0; k1 = na * nb * (G::COUT / BK);
    }
    struct Epi { gptr<const float> mask; gptr<float> out; };
    __device__ static Epi epi(const Args& a, int, int) { return Epi{pin_sgpr(a.mask), pin_sgpr(a.out)}; }
    __device__ static float epi_load(const Epi& e, int m, int n) { return e.mask[DxC2MP<WM_, WN_, TM_, TN_>::out_index(m, n)]; }
    __device__ static void store(const Epi& e, int m, int n, float v, float mask) { e.out[DxC2MP<WM_, WN_, TM_, TN_>::out_index(m, n)] = mask > 0.f ? v : 0.f; }
};
#ifndef BDR_DXC2MPOS_SHAPE
#define BDR_DXC2MPOS_SHAPE 2, 2, 1, 1  // 64 images x 64 columns: 800 workgroups at B = 256 (512 of 8 k-tiles, 256 of 4, 32 of 2); same box, 3 interleaved runs: 5 176-5 189 opt-steps/s against 5 093-5 095 for 64 x 128, 5 182-5 184 for 32 x 128, 5 054-5 061 for 128 x 64 and 5 025-5 037 for the flat row tiles of rounds 4-5
#endif
using DxC2MPos = DxC2MPosP<BDR_DXC2MPOS_SHAPE>;
template <class P> inline dim3 dxc2_pos_grid(int B) { return dim3((unsigned)((B + P::WM * P::TM * 32 - 1) / (P::WM * P::TM * 32)) * n_tiles<P>(), P::NPOS, 1); }

// ================================================================================================
// weight-gradient policies
// ================================================================================================
struct DwArgs {
    const void* x;      // layer input (activation of the previous layer)
    const float* dy;    // [M][N]
    float* part;        // partials: [chunks][K*N + N]
    size_t part_stride; // floats between chunks
    int M;
    unsigned* sig_flag; // optional progress flag written at kernel start (igemm.hpp start_signal)
    unsigned sig_epoch;
};
template <class G, class APolicy, int WM_, int WN_, bool U8>
struct DwP {
    using A = APolicy;
    using Args = DwArgs;
    static constexpr int WM = WM_, WN = WN_, TM = 1, TN = 1;
    __device__ static constexpr int K(const Args&) { return G::K; }
    __device__ static constexpr int N(const Args&) { return G::COUT; }
    __device__ static int M(const Args& a) { return a.M; }
    __device__ static auto a_src(const Args& a)
    {
        if constexpr (U8) return reinterpret_cast<const uint8_t*>(a.x);
        else return reinterpret_cast<const float*>(a.x);
    }
    __device__ static const float* y_src(const Args& a) { return a.dy; }
    __device__ static float* part(const Args& a, int chunk) { return a.part + (size_t)chunk * a.part_stride; }
};
using DwC2 = DwP<GeomC2, AFwd<GeomC2>, 2, 2, false>;     // 64 x 64: 8 ko-tiles
using DwC3 = DwP<GeomC3, AFwd<GeomC3>, 2, 2, false>;     // 64 x 64: 9 ko-tiles
using DwL1 = DwP<GeomL1, AFwd<GeomL1>, 2, 2, false>;     // 64 x 64: 49 x 8 tiles, single chunk

// g[i] = scale * sum_c part[c][i].  64 outputs per workgroup, 4 chunk groups per output (chunk c goes
// to group c % 4), fixed-order combine through LDS => deterministic.
__global__ __launch_bounds__(256) void k_reduce_partials(const float* __restrict__ part, size_t stride, int chunks,
                                                          float* __restrict__ g, int n, int n_weights, float wscale)
{
    __shared__ float red[4][64];
    const int o = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + o;
    float s = 0.f;
    if (i < n) {
        int c = grp;
        for (; c + 12 < chunks; c += 16) {
            const float v0 = part[(size_t)c * stride + i], v1 = part[(size_t)(c + 4) * stride + i];
            const float v2 = part[(size_t)(c + 8) * stride + i], v3 = part[(size_t)(c + 12) * stride + i];
            s += v0; s += v1; s += v2; s += v3;
        }
        for (; c < chunks; c += 4) s += part[(size_t)c * stride + i];
    }
    red[grp][o] = s;
    __syncthreads();
    if (grp == 0 && i < n) {
        const float t = ((red[0][o] + red[1][o]) + red[2][o]) + red[3][o];
        g[i] = i < n_weights ? t * wscale : t;
    }
}

// All three conv layers' partial reductions in ONE launch: 32 outputs x 8 chunk groups per workgroup
// (chunk c -> group c % 8), fixed-order combine => deterministic.
struct ReduceSeg { const float* part; size_t stride; int chunks; float* g; int n, n_weights; float wscale; int wg0; };
struct Reduce3Args { ReduceSeg seg[3]; };
__global__ __launch_bounds__(256) void k_reduce_partials3(Reduce3Args a)
{
    __shared__ float red[8][32];
    const int s_id = (int)blockIdx.x >= a.seg[2].wg0 ? 2 : ((int)blockIdx.x >= a.seg[1].wg0 ? 1 : 0);
    const ReduceSeg& sg = a.seg[s_id];
    const int o = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const int i = ((int)blockIdx.x - sg.wg0) * 32 + o;
    float s = 0.f;
    if (i < sg.n) {
        int c = grp;
        for (; c + 24 < sg.chunks; c += 32) {
            const float v0 = sg.part[(size_t)c * sg.stride + i], v1 = sg.part[(size_t)(c + 8) * sg.stride + i];
            const float v2 = sg.part[(size_t)(c + 16) * sg.stride + i], v3 = sg.part[(size_t)(c + 24) * sg.stride + i];
            s += v0; s += v1; s += v2; s += v3;
        }
        for (; c < sg.chunks; c += 8) s += sg.part[(size_t)c * sg.stride + i];
    }
    red[grp][o] = s;
    __syncthreads();
    if (grp == 0 && i < sg.n) {
        float t = red[0][o];
#pragma unroll
        for (int k = 1; k < 8; ++k) t += red[k][o];
        sg.g[i] = i < sg.n_weights ? t * sg.wscale : t;
    }
}

// chunk counts of the weight-gradient reductions (rows M split across workgroups)
struct DwPlan { int chunks_c1, chunks_c2, chunks_c3; size_t stride_c1, stride_c2, stride_c3, off_c1, off_c2, off_c3, total; };
DwPlan dw_plan(int B, int ns = 4)
{
    DwPlan p{};
    auto mt = [](int M) { return (M + 31) / 32; };
    p.chunks_c1 = std::min(256, B);           // conv1: one partial per workgroup, workgroups stride over the images
    p.chunks_c2 = std::min(64, mt(B * 81));
    p.chunks_c3 = std::min(56, mt(B * 49));
    // diagnostics (A/B of the partial-sum traffic): BDR_DW_CHUNKS="c1,c2,c3" caps the three counts (c2 / c3: multiples of 8 keep the XCD map)
    if (const char* e = getenv("BDR_DW_CHUNKS")) {
        int c1 = 0, c2 = 0, c3 = 0;
        if (sscanf(e, "%d,%d,%d", &c1, &c2, &c3) == 3) {
            if (c1 > 0) p.chunks_c1 = std::min(p.chunks_c1, c1);
            if (c2 > 0) p.chunks_c2 = std::min(p.chunks_c2, c2);
            if (c3 > 0) p.chunks_c3 = std::min(p.chunks_c3, c3);
        }
    }
    p.stride_c1 = (size_t)64 * ns * 32 + 32; p.stride_c2 = 512 * 64 + 64; p.stride_c3 = 576 * 64 + 64;
    p.off_c1 = 0;
    p.off_c2 = p.off_c1 + p.chunks_c1 * p.stride_c1;
    p.off_c3 = p.off_c2 + p.chunks_c2 * p.stride_c2;
    p.total = p.off_c3 + p.chunks_c3 * p.stride_c3;
    return p;
}

}  // namespace
