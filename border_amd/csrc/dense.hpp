// Dense (fully connected) layers for the Mlp-based agents (border-tch-agent/src/mlp/base.rs:13-41,
// mlp/mlp2.rs:7-50): the same FP32-MFMA kernels as the CNN path (igemm.hpp) driven by runtime
// dimensions.  Every dimension is zero-padded to a multiple of 64 inside the library, so a layer is
//   W [Kp][Np] (k-major, like the conv weights)   followed by   bias [Np]
// and an activation / gradient buffer is [B][Np].  Padding rows/columns of W and bias stay exactly
// zero under Adam (their gradients are exactly zero), so padded activations are zero as well.
#pragma once
#include <vector>

#include "agent_base.hpp"
#include "igemm_b3.hpp"
#include "igemm_red_b3.hpp"
#include "dense_k64_b3.hpp"

namespace bdr {

struct DenseLayer {
    int in, out;       // logical sizes
    int Kp, Np;        // padded sizes
    size_t w, b;       // offsets (floats) into the flat arena
    int relu;          // ReLU after this layer
};

struct MlpLayout {
    std::vector<DenseLayer> L;
    size_t total = 0;          // arena floats
    size_t ref_total = 0;      // reference-layout floats
    int in_dim = 0, out_dim = 0;
};

inline int pad64(int x) { return (x + 63) / 64 * 64; }

// Mlp (mlp/base.rs:13-41): in -> units[0] -> ... -> out, ReLU between, optional ReLU on the output.
// `base` lets several nets share one arena.
inline MlpLayout make_mlp(int in_dim, const int* units, int n_units, int out_dim, bool activation_out, size_t base = 0)
{
    MlpLayout m;
    m.in_dim = in_dim; m.out_dim = out_dim;
    size_t o = base;
    int in = in_dim;
    for (int i = 0; i <= n_units; ++i) {
        DenseLayer l;
        l.in = in; l.out = i < n_units ? units[i] : out_dim;
        l.Kp = pad64(l.in); l.Np = pad64(l.out);
        l.w = o; o += (size_t)l.Kp * l.Np;
        l.b = o; o += l.Np;
        l.relu = i < n_units ? 1 : (activation_out ? 1 : 0);
        m.ref_total += (size_t)l.out * l.in + l.out;
        m.L.push_back(l);
        in = l.out;
    }
    m.total = o - base;
    return m;
}

// reference order: ln{i}.weight [out][in], ln{i}.bias [out]   <->   internal W[k][n], b[n] (padded)
inline void mlp_to_internal(const MlpLayout& m, size_t base, const float* ref, float* in)
{
    const float* p = ref;
    for (const auto& l : m.L) {
        for (int o = 0; o < l.out; ++o)
            for (int k = 0; k < l.in; ++k) in[l.w - base + (size_t)k * l.Np + o] = p[(size_t)o * l.in + k];
        p += (size_t)l.out * l.in;
        for (int o = 0; o < l.out; ++o) in[l.b - base + o] = p[o];
        p += l.out;
    }
}
inline void mlp_to_reference(const MlpLayout& m, size_t base, const float* in, float* ref)
{
    float* p = ref;
    for (const auto& l : m.L) {
        for (int o = 0; o < l.out; ++o)
            for (int k = 0; k < l.in; ++k) p[(size_t)o * l.in + k] = in[l.w - base + (size_t)k * l.Np + o];
        p += (size_t)l.out * l.in;
        for (int o = 0; o < l.out; ++o) p[o] = in[l.b - base + o];
        p += l.out;
    }
}

// the library's own initialiser: uniform(+-1/sqrt(fan_in)) (tests inject weights through set_params)
inline void mlp_init_reference(const MlpLayout& m, uint64_t seed, float* ref)
{
    uint64_t s = seed * 0x9E3779B97F4A7C15ull + 0x7654321ull;
    float* p = ref;
    for (const auto& l : m.L) {
        const float bound = 1.0f / std::sqrt((float)l.in);
        const size_t n = (size_t)l.out * l.in + l.out;
        for (size_t i = 0; i < n; ++i) {
            s += 0x9E3779B97F4A7C15ull;
            uint64_t x = s; x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31;
            p[i] = ((float)(x >> 40) * (2.0f / 16777216.0f) - 1.0f) * bound;
        }
        p += n;
    }
}

}  // namespace bdr

namespace {
using namespace bdr;

// ---- kernel policies ---------------------------------------------------------------------------------
struct DenseArgs {
    DenseSrc x;          // A operand rows
    const float* w;      // layer weights [Kp][Np]
    const float* bias;   // fwd only
    float* out; int ldo;
    const float* mask; int ldm;   // dx: ReLU' mask = post-activation of the previous layer (or null)
    int M, ncols, kred, relu;
    int w_ld;            // row stride of W (= Np of the layer)
    int accum;           // dx: out += result (sum of several branches' input gradients)
    // optional second factor of the A operand (DenseFwdHad): A(m,k) = x[m][k] * xhad[m / xhad_group][k]
    const float* xhad; int xhad_ld, xhad_group;
    float* gsum; int gmask;   // DenseDxHad: per-group column sums (row stride xhad_ld) and whether they are masked by xhad > 0
};

struct DenseFwd {
    using A = ADense;
    using Args = DenseArgs;
    static constexpr int WM = 2, WN = 2, TM = 1, TN = 1;
    static constexpr int RPI = 1, RPIP = 0;
    __device__ static bool vrow(const Args& a, int mv, int& mr) { return vrow_flat(a.M, mv, mr); }
    static constexpr bool B_TR = false;
    __device__ static int N(const Args& a) { return a.ncols; }
    __device__ static int KP(const Args&) { return 32; }
    __device__ static int M(const Args& a) { return a.M; }
    __device__ static DenseSrc a_src(const Args& a, int) { return a.x; }
    __device__ static const float* w(const Args& a, int, int) { return a.w; }
    __device__ static int tap_index(int, int t) { return t; }
    __device__ static void kt_range(const Args& a, int, int& k0, int& k1) { k0 = 0; k1 = a.kred / BK; }
    using Epi = Args;   // dense arguments are not instance-indexed: plain kernel-argument reads
    __device__ static const Epi& epi(const Args& a, int, int) { return a; }
    __device__ static float epi_load(const Epi& a, int, int n) { return a.bias[n]; }
    __device__ static void store(const Epi& a, int m, int n, float v, float bias)
    {
        v += bias;
        if (a.relu) v = v > 0.f ? v : 0.f;
        a.out[(size_t)m * a.ldo + n] = v;
    }
};
// forward layer whose input rows are a Hadamard product that is never materialised (igemm.hpp ADenseHad)
struct DenseFwdHad : DenseFwd {
    using A = ADenseHad;
    __device__ static DenseSrcHad a_src(const Args& a, int) { return DenseSrcHad{a.x.p, a.x.ld, a.xhad, a.xhad_ld, a.xhad_group}; }
};

// Several networks of the same architecture in one launch (SAC's critics): instance z = blockIdx.z
struct DenseArgsZ {
    DenseArgs a[4];
    unsigned* sig_flag; unsigned sig_epoch;   // optional: the launch's first workgroup publishes "everything queued before me on my stream is complete" (igemm.hpp start_signal)
};
struct DenseFwdZ : DenseFwd {
    using Args = DenseArgsZ;
    __device__ static bool vrow(const Args& a, int mv, int& mr) { return vrow_flat(a.a[0].M, mv, mr); }
    __device__ static int N(const Args& a) { return a.a[0].ncols; }
    __device__ static int M(const Args& a) { return a.a[0].M; }
    __device__ static DenseSrc a_src(const Args& a, int z) { return a.a[z].x; }
    __device__ static const float* w(const Args& a, int z, int) { return a.a[z].w; }
    __device__ static void kt_range(const Args& a, int, int& k0, int& k1) { k0 = 0; k1 = a.a[0].kred / BK; }
    using Epi = DenseArgs;
    __device__ static const Epi& epi(const Args& a, int z, int) { return a.a[z]; }
};

// dX[M][Kp] = dY[M][Np] * W[Kp][Np]^T, optional ReLU' mask
struct DenseDx {
    using A = ADense;
    using Args = DenseArgs;
    static constexpr int WM = 2, WN = 2, TM = 1, TN = 1;
    static constexpr int RPI = 1, RPIP = 0;
    __device__ static bool vrow(const Args& a, int mv, int& mr) { return vrow_flat(a.M, mv, mr); }
    static constexpr bool B_TR = true;
    __device__ static int N(const Args& a) { return a.ncols; }     // = Kp of the layer
    __device__ static int KP(const Args& a) { return a.w_ld; }     // = Np of the layer (contiguous)
    __device__ static int M(const Args& a) { return a.M; }
    __device__ static DenseSrc a_src(const Args& a, int) { return a.x; }
    __device__ static const float* w(const Args& a, int, int) { return a.w; }
    __device__ static int tap_index(int, int t) { return t; }
    __device__ static void kt_range(const Args& a, int, int& k0, int& k1) { k0 = 0; k1 = a.kred / BK; }
    using Epi = Args;
    __device__ static const Epi& epi(const Args& a, int, int) { return a; }
    __device__ static float epi_load(const Epi& a, int m, int n)
    {
        return a.mask ? a.mask[(size_t)m * a.ldm + n] : 1.f;
    }
    __device__ static void store(const Epi& a, int m, int n, float v, float mask)
    {
        if (!(mask > 0.f)) v = 0.f;
        float* o = a.out + (size_t)m * a.ldo + n;
        *o = a.accum ? *o + v : v;
    }
};

// The same two layers on the bf16 matrix cores with split operands (igemm_b3.hpp: six of the nine exact bf16 partial products):
// the weights come from pre-split bf16 planes [plane][ncols][kred] (k contiguous: W^T for the forward, W itself for dX).
struct DenseB3Args : DenseArgs { const uint16_t* wpl; };
template <class Base, int WM_, int WN_, int TM_, int TN_>
struct DenseB3 : Base {
    using Args = DenseB3Args;
    static constexpr int WM = WM_, WN = WN_, TM = TM_, TN = TN_;
    static constexpr int MAXW = TM_ * TN_ >= 4 ? 1 : 2;
    __device__ static const uint4* b_chunk(const Args& a, int, int, int pl, int kt, int n, int kq)
    {
        return reinterpret_cast<const uint4*>(a.wpl + (size_t)pl * a.ncols * a.kred + (size_t)n * a.kred + kt * 32 + kq * 8);
    }
};
// dX of the layer behind a Hadamard merge m[r] = g[r / group] * e[r] (e = a post-ReLU activation), with the merge's backward in the
// epilogue (igemm_b3.hpp's row-group epilogue; group = the wave's TM * 32 rows):  d e_pre[r][n] = 1{e > 0} dm g[group][n]  and
// d g[group][n] = sum_r dm[r][n] e[r][n]  (masked by g > 0 when g is itself a ReLU output).  mask = e, xhad = g.
struct DenseDxHad : DenseDx {
    static constexpr bool GROUP_EPI = true;
    __device__ static float epi_load(const Epi& a, int m, int n) { return a.mask[(size_t)m * a.ldm + n]; }
    __device__ static float group_scale(const Epi& a, int g, int n) { return a.xhad[(size_t)g * a.xhad_ld + n]; }
    __device__ static void group_elem_store(const Epi& a, int m, int n, float v, float e, float gs) { a.out[(size_t)m * a.ldo + n] = e > 0.f ? v * gs : 0.f; }
    __device__ static void group_store(const Epi& a, int g, int n, float s, float gs) { a.gsum[(size_t)g * a.xhad_ld + n] = (a.gmask && !(gs > 0.f)) ? 0.f : s; }
};
using DenseFwdHadB3 = DenseB3<DenseFwdHad, 2, 2, 2, 2>;   // 128 x 128 tiles (ncols % 128 == 0)
using DenseDxB3 = DenseB3<DenseDx, 2, 2, 2, 2>;           // 128 x 128 tiles (ncols % 64 == 0: the last column tile may be half empty)
using DenseDxHadB3 = DenseB3<DenseDxHad, 2, 2, 2, 2>;     // 128 x 128 tiles (a ragged last column tile when ncols % 128 = 64), row groups of 64:
                                                          // a 128 x 64 tile reads 0.75 LDS fragments per MFMA and is LDS-bound (0.32 MFMA busy), this one 0.5

struct DenseDxZ : DenseDx {
    using Args = DenseArgsZ;
    __device__ static bool vrow(const Args& a, int mv, int& mr) { return vrow_flat(a.a[0].M, mv, mr); }
    __device__ static int N(const Args& a) { return a.a[0].ncols; }
    __device__ static int KP(const Args& a) { return a.a[0].w_ld; }
    __device__ static int M(const Args& a) { return a.a[0].M; }
    __device__ static DenseSrc a_src(const Args& a, int z) { return a.a[z].x; }
    __device__ static const float* w(const Args& a, int z, int) { return a.a[z].w; }
    __device__ static void kt_range(const Args& a, int, int& k0, int& k1) { k0 = 0; k1 = a.a[0].kred / BK; }
    using Epi = DenseArgs;
    __device__ static const Epi& epi(const Args& a, int z, int) { return a.a[z]; }
};

struct DenseDwArgs {
    DenseSrc x;          // layer input rows [M][Kp]
    const float* dy;     // [M][Np]
    float* part; size_t part_stride;
    int M, Kp, Np;
    const float* xhad; int xhad_ld, xhad_group;   // DenseDwHad: x[m][k] * xhad[m / xhad_group][k]
};
struct DenseDw {
    using A = ADense;
    using Args = DenseDwArgs;
    static constexpr int WM = 2, WN = 2, TM = 1, TN = 1;
    __device__ static int K(const Args& a) { return a.Kp; }
    __device__ static int N(const Args& a) { return a.Np; }
    __device__ static int M(const Args& a) { return a.M; }
    __device__ static DenseSrc a_src(const Args& a) { return a.x; }
    __device__ static const float* y_src(const Args& a) { return a.dy; }
    __device__ static float* part(const Args& a, int chunk) { return a.part + (size_t)chunk * a.part_stride; }
};

struct DenseDwHad : DenseDw {
    using A = ADenseHad;
    __device__ static DenseSrcHad a_src(const Args& a) { return DenseSrcHad{a.x.p, a.x.ld, a.xhad, a.xhad_ld, a.xhad_group}; }
};

// g[i] = sum_c part[c][i]  (same deterministic scheme as the conv path)
__global__ __launch_bounds__(256) void k_dense_reduce(const float* __restrict__ part, size_t stride, int chunks,
                                                       float* __restrict__ g, int n)
{
    __shared__ float red[4][64];
    const int o = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + o;
    float s = 0.f;
    if (i < n) for (int c = grp; c < chunks; c += 4) s += part[(size_t)c * stride + i];
    red[grp][o] = s;
    __syncthreads();
    if (grp == 0 && i < n) g[i] = ((red[0][o] + red[1][o]) + red[2][o]) + red[3][o];
}

// copy the logical columns of row-major host-layout rows into a zero-padded [B][ld] matrix:
// dst[b][col0 + c] = src[b*src_ld + c], c < cols
__global__ void k_pack_rows(const float* __restrict__ src, int src_ld, int cols, float* __restrict__ dst, int ld, int col0, int B)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * cols) return;
    const int b = i / cols, c = i % cols;
    dst[(size_t)b * ld + col0 + c] = src[(size_t)b * src_ld + c];
}

// ---- small layers: one 32x32 output tile per workgroup, the reduction split over its four waves ---------
// A 64x64 tile of k_igemm keeps one CU busy for kred/2 * 64 cycles however its waves are arranged (the CU's four matrix pipes
// are the bound: 3.4 us for kred = 256, measured 5.5 us with staging; tools/probes/dense_trace.hip), and a 1024 x 256 layer has
// only 64 such tiles for 256 CUs.  Launch-bound agents (SAC: ~20 such GEMMs per step, one after the other) want latency, not
// reuse: here a workgroup owns a 32x32 tile, wave w multiplies the k-slice [w*kred/4, (w+1)*kred/4) straight from global memory
// (every operand load of the slice is issued before the first MFMA: no LDS staging, no barrier in the loop) and the four
// partial tiles are added through LDS in a fixed order.  4x the workgroups, 1/4 of the MFMA chain per wave.
// DX = false: out[m][n] = act(sum_k x[m][k] W[k][n] + bias[n]);  DX = true: out[m][kc] (+)= mask * sum_n dy[m][n] W[kc][n].
// The tile itself, shared with the fused row-block kernels of the SAC step (sac.hip): wave `wave` of four accumulates its k-slice
// of the 32x32 tile (m0, n0) and leaves it in red[wave]; after a barrier dense_small_sum() adds the four slices in a fixed order.
// Every kernel that forms a layer's tile through these two functions produces the same bits.
//   loadA(k) -> the lane's four A values a[row i][k .. k+3]   (global rows, or rows a fused kernel keeps in LDS)
//   DX = false: B(k, n) = w[k * w_ld + n0 + i];  DX = true: B(k, n) = w[(n0 + i) * w_ld + k]  (transposed weights)
template <bool DX, class LoadA>
__device__ __forceinline__ void dense_small_tile(LoadA&& loadA, const float* __restrict__ w, int w_ld, int n0, int kred, int wave, int lane,
                                                 float (*red)[32][33])
{
    const int i = lane & 31, h = lane >> 5;
    const int Kw = kred / 4;                   // kred % 64 == 0: a multiple of 16
    const int kbeg = wave * Kw;
    const float* wcol = DX ? w + (size_t)(n0 + i) * w_ld : w + n0 + i;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int kb = 0; kb < Kw; kb += 64) {      // up to four 16-wide chunks in flight (all of the slice for kred <= 256)
        f32x4 av[8], bv[8];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (kb + 16 * c < Kw) {            // wave-uniform
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int k = kbeg + kb + 16 * c + 8 * u + 4 * h;   // this lane half's quad of the MFMA group (igemm.hpp: k-slots are free)
                    av[c * 2 + u] = loadA(k);
                    if constexpr (DX) bv[c * 2 + u] = *reinterpret_cast<const f32x4*>(wcol + k);
                    else { const float* p = wcol + (size_t)k * w_ld; bv[c * 2 + u] = f32x4{p[0], p[w_ld], p[2 * (size_t)w_ld], p[3 * (size_t)w_ld]}; }
                }
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (kb + 16 * c < Kw) {
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c * 2 + u][q], bv[c * 2 + u][q], acc, 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][(r & 3) + 8 * (r >> 2) + 4 * h][i] = acc[r];
}
// The same tile with the B operand fetched ahead of time (kred <= 256: one pass of the loop above): a row-block kernel whose second
// tile takes its A operand from LDS loads the weights at kernel start instead of as a dependent round trip in the middle.
// Same k-slots, same MFMA order as dense_small_tile: same bits.
template <bool DX>
__device__ __forceinline__ void dense_small_load_b(const float* __restrict__ w, int w_ld, int n0, int kred, int wave, int lane, f32x4 (&bv)[8])
{
    const int i = lane & 31, h = lane >> 5;
    const int Kw = kred / 4, kbeg = wave * Kw;
    const float* wcol = DX ? w + (size_t)(n0 + i) * w_ld : w + n0 + i;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (16 * c < Kw) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int k = kbeg + 16 * c + 8 * u + 4 * h;
                if constexpr (DX) bv[c * 2 + u] = *reinterpret_cast<const f32x4*>(wcol + k);
                else { const float* p = wcol + (size_t)k * w_ld; bv[c * 2 + u] = f32x4{p[0], p[w_ld], p[2 * (size_t)w_ld], p[3 * (size_t)w_ld]}; }
            }
        }
    }
}
template <class LoadA>
__device__ __forceinline__ void dense_small_tile_pre(LoadA&& loadA, const f32x4 (&bv)[8], int kred, int wave, int lane, float (*red)[32][33])
{
    const int i = lane & 31, h = lane >> 5;
    const int Kw = kred / 4, kbeg = wave * Kw;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    f32x4 av[8];
#pragma unroll
    for (int c = 0; c < 4; ++c)
        if (16 * c < Kw) {
#pragma unroll
            for (int u = 0; u < 2; ++u) av[c * 2 + u] = loadA(kbeg + 16 * c + 8 * u + 4 * h);
        }
#pragma unroll
    for (int c = 0; c < 4; ++c)
        if (16 * c < Kw) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c * 2 + u][q], bv[c * 2 + u][q], acc, 0, 0, 0);
        }
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][(r & 3) + 8 * (r >> 2) + 4 * h][i] = acc[r];
}
// element (r, c) of the tile: the four wave slices in their fixed order (call after the barrier that follows dense_small_tile)
__device__ __forceinline__ float dense_small_sum(const float (*red)[32][33], int r, int c)
{
    return ((red[0][r][c] + red[1][r][c]) + red[2][r][c]) + red[3][r][c];
}

// Values one workgroup hands to the LAST workgroup of the same launch go through agent-scope accesses: the XCDs' L2 caches are
// not coherent with each other, and a full release fence (__threadfence) would write back every dirty line of the XCD's L2 - tens
// of microseconds per workgroup (measured: 33 us for a kernel that otherwise takes 8).  A relaxed agent-scope store / load goes to
// the coherent level directly; an explicit s_waitcnt vmcnt(0) + the workgroup barrier order them before the ticket (last_workgroup).
__device__ __forceinline__ void st_agent(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_agent(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// "am I the last workgroup of this launch?" in two halves, so that work the answer does not depend on (stores of results nobody in
// this launch reads) can sit between them and hide the ticket's round trip:
//   ticket_take   after the workgroup's LAST agent-scope store to data the last workgroup will read; contains a workgroup barrier
//   ticket_last   the answer (contains a workgroup barrier)
__device__ __forceinline__ void ticket_take(unsigned* ticket, unsigned n_wg, unsigned* s_flag)
{
    // every thread's agent-scope stores acknowledged by the coherent level before the ticket is taken: the workgroup barrier alone
    // does not wait for them (a workgroup-scope release is lgkmcnt only outside threadgroup-split mode)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *s_flag = t == n_wg - 1 ? 1u : 0u;
        if (t == n_wg - 1) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-armed for the next launch
    }
}
__device__ __forceinline__ bool ticket_last(const unsigned* s_flag)
{
    __syncthreads();
    return *s_flag != 0u;
}
__device__ __forceinline__ bool last_workgroup(unsigned* ticket, unsigned n_wg, unsigned* s_flag)
{
    ticket_take(ticket, n_wg, s_flag);
    return ticket_last(s_flag);
}

struct HeadRef { const float* w; const float* bias; int relu; };   // a narrow layer handed to a row-block kernel

// (the body as a function: sac.hip's k_dense_small_dx_tail runs it beside one more workgroup)
template <bool DX>
__device__ __forceinline__ void dense_small_body(const DenseArgs& a, float (*red)[32][33])
{
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NT = a.ncols / 32;
    const int m0 = ((int)blockIdx.x / NT) * 32, n0 = ((int)blockIdx.x % NT) * 32;
    const float* arow = a.x.p + (size_t)min(m0 + (lane & 31), a.M - 1) * a.x.ld;   // rows >= M alias the last row (never stored)
    // the epilogue's operands (bias / ReLU mask / the value accumulated onto) depend on nothing this kernel computes: their loads are
    // issued here, beside the tile operands, instead of as one more memory round trip behind the barrier (the step is a chain of
    // these launches, each a handful of dependent round trips long)
    const int r = tid >> 3, c4 = (tid & 7) * 4, m = m0 + r, n = n0 + c4, mc = min(m, a.M - 1);
    float* o = a.out + (size_t)mc * a.ldo + n;
    f32x4 e0 = {0.f, 0.f, 0.f, 0.f}, e1 = {0.f, 0.f, 0.f, 0.f};
    if constexpr (!DX) e0 = *reinterpret_cast<const f32x4*>(a.bias + n);
    else {
        if (a.mask) e0 = *reinterpret_cast<const f32x4*>(a.mask + (size_t)mc * a.ldm + n);
        if (a.accum) e1 = *reinterpret_cast<const f32x4*>(o);
    }
    dense_small_tile<DX>([&](int k) { return *reinterpret_cast<const f32x4*>(arow + k); }, a.w, a.w_ld, n0, a.kred, wave, lane, red);
    __syncthreads();
    if (m >= a.M) return;
    f32x4 v;
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = dense_small_sum(red, r, c4 + q);
    if constexpr (!DX) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { v[q] += e0[q]; if (a.relu) v[q] = v[q] > 0.f ? v[q] : 0.f; }
    } else {
        if (a.mask) {
#pragma unroll
            for (int q = 0; q < 4; ++q) if (!(e0[q] > 0.f)) v[q] = 0.f;
        }
        if (a.accum) v += e1;
    }
    *reinterpret_cast<f32x4*>(o) = v;
}
template <bool DX>
__global__ __launch_bounds__(256) void k_dense_small(DenseArgsZ dz)
{
    start_signal(dz.sig_flag, dz.sig_epoch);
    __shared__ float red[4][32][33];
    dense_small_body<DX>(dz.a[blockIdx.z], red);
}
template <bool DX>
inline hipError_t launch_dense_small(hipStream_t st, const DenseArgsZ& dz, int nz)
{
    const DenseArgs& d = dz.a[0];
    return step_launch(st, false, k_dense_small<DX>, dim3(((d.M + 31) / 32) * (d.ncols / 32), 1, nz), dim3(256), dz);
}

// ---- host-side layer launches ------------------------------------------------------------------------
// (two teams of four waves per tile taking alternate k-tiles - k_igemm's TEAMS = 2 - were measured on SAC's 1024 x 256 x 256
// layers: no gain, these launches are bound by kernel start-up and first-touch latency, not by the length of the k loop)
template <class P>
inline hipError_t launch_dense(hipStream_t st, dim3 grid, const typename P::Args& d)
{
    return step_launch(st, false, k_igemm<P, 1>, grid, dim3(256), d);
}
// sig_flag (small only): the kernel's start publishes sig_epoch (DenseArgsZ)
inline int32_t dense_forward(bdr_agent* a, hipStream_t st, const DenseLayer& l, const float* params_base, DenseSrc x, float* out, int M,
                             bool small = false, unsigned* sig_flag = nullptr, unsigned sig_epoch = 0)
{
    DenseArgs d{};
    d.x = x; d.w = params_base + l.w; d.bias = params_base + l.b; d.out = out; d.ldo = l.Np;
    d.M = M; d.ncols = l.Np; d.kred = l.Kp; d.relu = l.relu; d.w_ld = l.Np;
    if (small) { DenseArgsZ dz{}; dz.a[0] = d; dz.sig_flag = sig_flag; dz.sig_epoch = sig_epoch; BDR_HIP(launch_dense_small<false>(st, dz, 1)); return BDR_OK; }
    BDR_HIP(launch_dense<DenseFwd>(st, dim3(((M + 63) / 64) * (l.Np / 64), 1, 1), d));
    return BDR_OK;
}
// the same layer on input rows x[m][k] * had[m / had_group][k] (the product exists only inside the kernel)
inline int32_t dense_forward_had(hipStream_t st, const DenseLayer& l, const float* params_base, DenseSrc x, const float* had, int had_ld,
                                 int had_group, float* out, int M)
{
    DenseArgs d{};
    d.x = x; d.xhad = had; d.xhad_ld = had_ld; d.xhad_group = had_group;
    d.w = params_base + l.w; d.bias = params_base + l.b; d.out = out; d.ldo = l.Np;
    d.M = M; d.ncols = l.Np; d.kred = l.Kp; d.relu = l.relu; d.w_ld = l.Np;
    BDR_HIP(launch_dense<DenseFwdHad>(st, dim3(((M + 63) / 64) * (l.Np / 64), 1, 1), d));
    return BDR_OK;
}

// Split-operand forms of the two calls above (igemm_b3.hpp).  planes_tr / planes_nat: the layer's weights as three bf16 planes,
// [plane][Np][Kp] and [plane][Kp][Np] (dense_split_planes).  Shapes: Np % 128 == 0 for the forward, Kp % 64 == 0 for dX.
inline int32_t dense_split_planes(hipStream_t st, const DenseLayer& l, const float* params_base, uint16_t* planes_nat, uint16_t* planes_tr)
{
    hipLaunchKernelGGL(k_split_planes2, dim3((l.Np + 31) / 32, (l.Kp + 31) / 32), dim3(256), 0, st, params_base + l.w, l.Np, planes_nat, planes_tr, l.Kp, l.Np);
    BDR_HIP(hipGetLastError());
    return BDR_OK;
}
// a layer with Kp = 64 over many rows (dense_k64_b3.hpp): planes_tr = [3][Np][64]
inline int32_t dense_forward_k64_b3(hipStream_t st, const DenseLayer& l, const float* params_base, const uint16_t* planes_tr, DenseSrc x, float* out, int M)
{
    DenseK64Args d{x.p, x.ld, planes_tr, params_base + l.b, out, l.Np, M, l.Np, l.relu};
    hipLaunchKernelGGL(k_dense_k64_b3, dim3((M + 127) / 128), dim3(256), 0, st, d);
    BDR_HIP(hipGetLastError());
    return BDR_OK;
}
inline int32_t dense_forward_had_b3(hipStream_t st, const DenseLayer& l, const float* params_base, const uint16_t* planes_tr, DenseSrc x, const float* had,
                                    int had_ld, int had_group, float* out, int M)
{
    DenseB3Args d{};
    d.x = x; d.xhad = had; d.xhad_ld = had_ld; d.xhad_group = had_group;
    d.w = params_base + l.w; d.bias = params_base + l.b; d.out = out; d.ldo = l.Np;
    d.M = M; d.ncols = l.Np; d.kred = l.Kp; d.relu = l.relu; d.w_ld = l.Np; d.wpl = planes_tr;
    BDR_HIP((launch_igemm_b3<DenseFwdHadB3, 6>(st, dim3(((M + 127) / 128) * (l.Np / 128), 1, 1), d)));
    return BDR_OK;
}
inline int32_t dense_dx_b3(hipStream_t st, const DenseLayer& l, const float* params_base, const uint16_t* planes_nat, const float* dy, float* dx,
                           const float* mask, int M)
{
    DenseB3Args d{};
    d.x = DenseSrc{dy, l.Np}; d.w = params_base + l.w; d.out = dx; d.ldo = l.Kp; d.mask = mask; d.ldm = l.Kp;
    d.M = M; d.ncols = l.Kp; d.kred = l.Np; d.w_ld = l.Np; d.wpl = planes_nat;
    BDR_HIP((launch_igemm_b3<DenseDxB3, 6>(st, dim3(((M + 127) / 128) * ((l.Kp + 127) / 128), 1, 1), d)));
    return BDR_OK;
}

// ... with the Hadamard merge's backward in its epilogue (DenseDxHad): dx <- 1{e > 0} dm g[group], gsum[group] <- sum_rows dm e.
// Rows come in groups of 64 (= the wave's rows) and M is a multiple of the 128-row tile: the caller checks both.
inline int32_t dense_dx_had_b3(hipStream_t st, const DenseLayer& l, const float* params_base, const uint16_t* planes_nat, const float* dy, float* dx,
                               const float* e, const float* g, int ldg, float* gsum, int gmask, int M)
{
    DenseB3Args d{};
    d.x = DenseSrc{dy, l.Np}; d.w = params_base + l.w; d.out = dx; d.ldo = l.Kp; d.mask = e; d.ldm = l.Kp;
    d.M = M; d.ncols = l.Kp; d.kred = l.Np; d.w_ld = l.Np; d.wpl = planes_nat;
    d.xhad = g; d.xhad_ld = ldg; d.xhad_group = 64; d.gsum = gsum; d.gmask = gmask;
    BDR_HIP((launch_igemm_b3<DenseDxHadB3, 6>(st, dim3((M / 128) * ((l.Kp + 127) / 128), 1, 1), d)));
    return BDR_OK;
}

// the same layer of nz (<= 4) networks of one architecture in one launch: params_base[z], x[z], out[z]
inline int32_t dense_forward_z(hipStream_t st, const DenseLayer& l, int nz, const float* const* params_base, const DenseSrc* x, float* const* out, int M,
                               bool small = false)
{
    DenseArgsZ dz{};
    for (int z = 0; z < nz; ++z) {
        DenseArgs& d = dz.a[z];
        d.x = x[z]; d.w = params_base[z] + l.w; d.bias = params_base[z] + l.b; d.out = out[z]; d.ldo = l.Np;
        d.M = M; d.ncols = l.Np; d.kred = l.Kp; d.relu = l.relu; d.w_ld = l.Np;
    }
    if (small) { BDR_HIP(launch_dense_small<false>(st, dz, nz)); return BDR_OK; }
    BDR_HIP(launch_dense<DenseFwdZ>(st, dim3(((M + 63) / 64) * (l.Np / 64), 1, nz), dz));
    return BDR_OK;
}
inline int32_t dense_dx_z(hipStream_t st, const DenseLayer& l, int nz, const float* const* params_base, const float* const* dy, float* const* dx,
                          const float* const* mask, int M, bool small = false)
{
    DenseArgsZ dz{};
    for (int z = 0; z < nz; ++z) {
        DenseArgs& d = dz.a[z];
        d.x = DenseSrc{dy[z], l.Np}; d.w = params_base[z] + l.w; d.out = dx[z]; d.ldo = l.Kp; d.mask = mask ? mask[z] : nullptr; d.ldm = l.Kp;
        d.M = M; d.ncols = l.Kp; d.kred = l.Np; d.w_ld = l.Np;
    }
    if (small) { BDR_HIP(launch_dense_small<true>(st, dz, nz)); return BDR_OK; }
    BDR_HIP(launch_dense<DenseDxZ>(st, dim3(((M + 63) / 64) * (l.Kp / 64), 1, nz), dz));
    return BDR_OK;
}

// dX (masked by the ReLU of the producing layer's post-activation `mask`, may be null)
inline int32_t dense_dx(hipStream_t st, const DenseLayer& l, const float* params_base, const float* dy, float* dx,
                        const float* mask, int M, bool accum = false, bool small = false)
{
    DenseArgs d{};
    d.accum = accum ? 1 : 0;
    d.x = DenseSrc{dy, l.Np}; d.w = params_base + l.w; d.out = dx; d.ldo = l.Kp; d.mask = mask; d.ldm = l.Kp;
    d.M = M; d.ncols = l.Kp; d.kred = l.Np; d.w_ld = l.Np;
    if (small) { DenseArgsZ dz{}; dz.a[0] = d; BDR_HIP(launch_dense_small<true>(st, dz, 1)); return BDR_OK; }
    BDR_HIP(launch_dense<DenseDx>(st, dim3(((M + 63) / 64) * (l.Kp / 64), 1, 1), d));
    return BDR_OK;
}

// dW, db straight into the gradient arena (single row chunk), or through `part` (chunks > 1)
inline int32_t dense_dw(hipStream_t st, const DenseLayer& l, float* grad_base, DenseSrc x, const float* dy, int M,
                        float* part = nullptr, int chunks = 1, const float* had = nullptr, int had_ld = 0, int had_group = 1)
{
    const int tiles = (l.Kp / 64) * (l.Np / 64);
    const int n = l.Kp * l.Np + l.Np;
    DenseDwArgs d{x, dy, chunks > 1 ? part : grad_base + l.w, chunks > 1 ? (size_t)n : 0, M, l.Kp, l.Np, had, had_ld, had_group};
    if (had) BDR_HIP(step_launch(st, false, k_igemm_red<DenseDwHad>, dim3(tiles * chunks), dim3(256), d));
    else BDR_HIP(step_launch(st, false, k_igemm_red<DenseDw>, dim3(tiles * chunks), dim3(256), d));
    if (chunks > 1) {
        BDR_HIP(step_launch(st, false, k_dense_reduce, dim3((n + 63) / 64), dim3(256), part, (size_t)n, chunks, grad_base + l.w, n));
    }
    return BDR_OK;
}

// The weight gradient on the bf16 matrix cores with split operands (igemm_red_b3.hpp): dy is split ONCE into transposed planes
// (ytr: [3][Np][M] u16, caller-owned), X (x had[m / had_group]) in the kernel.  M % 64 == 0, Np % 128 == 0, Kp even, had_group % 8 == 0.
inline int32_t dense_dw_b3(hipStream_t st, const DenseLayer& l, float* grad_base, DenseSrc x, const float* dy, uint16_t* ytr, int M, float* part, int chunks,
                           const float* had = nullptr, int had_ld = 0, int had_group = 8)
{
    const int n = l.Kp * l.Np + l.Np;
    hipLaunchKernelGGL(k_split_rows_tr, dim3(l.Np / 64, M / 64), dim3(256), 0, st, dy, l.Np, ytr, M, l.Np);
    BDR_HIP(hipGetLastError());
    RedB3Args a{x.p, x.ld, had, had_ld, had_group, ytr, part, (size_t)n, M, l.Kp, l.Np, chunks};
    const int tiles = ((l.Kp + 127) / 128) * (l.Np / 128);
    hipLaunchKernelGGL(k_igemm_red_b3<6>, dim3(tiles * chunks), dim3(256), 0, st, a);
    BDR_HIP(hipGetLastError());
    BDR_HIP(step_launch(st, false, k_dense_reduce, dim3((n + 63) / 64), dim3(256), part, (size_t)n, chunks, grad_base + l.w, n));
    return BDR_OK;
}

// Row chunks for a dW launch: the kernel has (Kp/64)*(Np/64) output tiles; a 256x256 layer over 1 024 rows would run on 16
// workgroups (32 us on a 256-CU chip).  Split the row reduction so that about one workgroup per CU is in flight (at least
// 64 rows per chunk, at most 16 chunks; the partials are summed in a fixed order by k_dense_reduce).
inline int dense_dw_chunks(const DenseLayer& l, int M)
{
    const int tiles = (l.Kp / 64) * (l.Np / 64);
    return std::max(1, std::min(std::min(16, M / 64), (256 + tiles - 1) / tiles));
}
inline size_t dense_dw_part_floats(const DenseLayer& l) { return 16 * ((size_t)l.Kp * l.Np + l.Np); }

// ---- grouped weight gradients + fused reduce / Adam / track (launch-bound agents) -----------------------
// All dW GEMMs of a backward pass only meet again in the optimizer, so they run as ONE grouped launch (k_igemm_red_group) into
// per-GEMM row-chunk partials, and ONE kernel sums the partials in k_dense_reduce's order, stores the gradient arena and applies
// Adam (and the target tracking) - 2 launches where the layer-by-layer path takes 3 per layer + 2 per network.
constexpr int DW_GROUP = 16;
struct DenseDwJob { const DenseLayer* l; DenseSrc x; const float* dy; float* part; int chunks; };
inline int32_t dense_dw_group(hipStream_t st, const DenseDwJob* jobs, int n, int M)
{
    for (int j0 = 0; j0 < n; j0 += DW_GROUP) {
        IgemmRedGroup<DenseDwArgs, DW_GROUP> g{};
        const int nj = std::min(DW_GROUP, n - j0);
        int first = 0;
        for (int j = 0; j < nj; ++j) {
            const DenseDwJob& q = jobs[j0 + j];
            const int tiles = (q.l->Kp / 64) * (q.l->Np / 64);
            g.a[j] = DenseDwArgs{q.x, q.dy, q.part, (size_t)q.l->Kp * q.l->Np + q.l->Np, M, q.l->Kp, q.l->Np, nullptr, 0, 1};
            g.first[j] = first;
            first += tiles * q.chunks;
        }
        for (int j = nj; j <= DW_GROUP; ++j) g.first[j] = first;
        g.n = nj;
        BDR_HIP(step_launch(st, false, k_igemm_red_group<DenseDw, DW_GROUP>, dim3(first), dim3(256), g));
    }
    return BDR_OK;
}

// The same grouped launch with the latency-oriented tile of k_dense_small: a workgroup owns a 32x32 tile of dW (k rows x n
// columns) for one row chunk, its four waves reduce a quarter of the chunk's batch rows each, straight from global memory (x and
// dY rows are read along their contiguous dimension, 128 B per row and operand), partial tiles are added through LDS in a fixed
// order; the workgroups of k-tile 0 also produce the bias partials (column sums of the dY values they hold).  A 64x64 tile of
// k_igemm_red_group walks 2-6 row blocks with a barrier each and there are few of them; here 450-770 workgroups run 16-32 MFMAs
// per wave.  Partials have k_igemm_red's layout ([Kp*Np] weights then [Np] bias per chunk): k_dense_reduce_adam is unchanged.
struct DenseDwSmallGroup { DenseDwArgs a[DW_GROUP]; int first[DW_GROUP + 1]; int chunks[DW_GROUP]; int n; };
__global__ __launch_bounds__(256) void k_dense_dw_small_group(DenseDwSmallGroup g)
{
    __shared__ float red[4][32][33];
    __shared__ float bred[4][32];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = (int)blockIdx.x;
    int z = 0;
#pragma unroll
    for (int k = 1; k < DW_GROUP; ++k) z += (k < g.n && b >= g.first[k]) ? 1 : 0;
    const DenseDwArgs& a = g.a[z];
    const int nch = g.chunks[z], local = b - g.first[z];
    const int tile = local / nch, chunk = local % nch;
    const int NT = a.Np / 32, kt = tile / NT, nt = tile % NT;
    const int i = lane & 31, h = lane >> 5;
    // rows of this chunk, then of this wave: multiples of 8 (an MFMA group takes 8 batch rows)
    const int per_chunk = ((a.M + nch - 1) / nch + 31) / 32 * 32;
    const int r0 = chunk * per_chunk, r1 = min(a.M, r0 + per_chunk);
    const int per_wave = per_chunk / 4;                       // multiple of 8
    const int w0 = r0 + wave * per_wave, w1 = min(r1, w0 + per_wave);
    const float* xcol = a.x.p + kt * 32 + i;
    const float* ycol = a.dy + nt * 32 + i;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float bsum = 0.f;
    // 32 rows = 4 MFMA groups = 32 loads per lane; the next 32 rows are requested before this batch's MFMAs (two batches in flight)
    auto load32 = [&](int rb, float (&av)[4][4], float (&bv)[4][4]) {
        const float* px = xcol + (size_t)(rb + 4 * h) * a.x.ld;
        const float* py = ycol + (size_t)(rb + 4 * h) * a.Np;
        if (rb + 32 <= w1) {                                  // full batch (wave-uniform): straight-line loads
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int s = 0; s < 4; ++s) { av[u][s] = px[(8 * u + s) * a.x.ld]; bv[u][s] = py[(8 * u + s) * a.Np]; }
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const bool ok = rb + 8 * u + 4 * h + s < w1;
                    const float xa = ok ? px[(8 * u + s) * a.x.ld] : 0.f, ya = ok ? py[(8 * u + s) * a.Np] : 0.f;
                    av[u][s] = xa; bv[u][s] = ya;
                }
        }
    };
    auto mma32 = [&](const float (&av)[4][4], const float (&bv)[4][4]) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][s], bv[u][s], acc, 0, 0, 0);
                bsum += bv[u][s];
            }
    };
    {
        float a0[4][4], b0[4][4], a1[4][4], b1[4][4];
        if (w0 < w1) load32(w0, a0, b0);
        for (int rb = w0; rb < w1; rb += 64) {
            if (rb + 32 < w1) load32(rb + 32, a1, b1);
            mma32(a0, b0);
            if (rb + 64 < w1) load32(rb + 64, a0, b0);
            if (rb + 32 < w1) mma32(a1, b1);
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][(r & 3) + 8 * (r >> 2) + 4 * h][i] = acc[r];
    if (kt == 0) {
        bsum += __shfl_xor(bsum, 32);                         // the two row parities of the lane's column
        if (h == 0) bred[wave][i] = bsum;
    }
    __syncthreads();
    float* part = a.part + (size_t)chunk * a.part_stride;
    const int r = tid >> 3, c4 = (tid & 7) * 4;
    f32x4 v;
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = ((red[0][r][c4 + q] + red[1][r][c4 + q]) + red[2][r][c4 + q]) + red[3][r][c4 + q];
    *reinterpret_cast<f32x4*>(part + (size_t)(kt * 32 + r) * a.Np + nt * 32 + c4) = v;
    if (kt == 0 && tid < 32) part[(size_t)a.Kp * a.Np + nt * 32 + tid] = ((bred[0][tid] + bred[1][tid]) + bred[2][tid]) + bred[3][tid];
}
inline int32_t dense_dw_small_group(hipStream_t st, const DenseDwJob* jobs, int n, int M)
{
    for (int j0 = 0; j0 < n; j0 += DW_GROUP) {
        DenseDwSmallGroup g{};
        const int nj = std::min(DW_GROUP, n - j0);
        int first = 0;
        for (int j = 0; j < nj; ++j) {
            const DenseDwJob& q = jobs[j0 + j];
            g.a[j] = DenseDwArgs{q.x, q.dy, q.part, (size_t)q.l->Kp * q.l->Np + q.l->Np, M, q.l->Kp, q.l->Np, nullptr, 0, 1};
            g.first[j] = first; g.chunks[j] = q.chunks;
            first += (q.l->Kp / 32) * (q.l->Np / 32) * q.chunks;
        }
        for (int j = nj; j <= DW_GROUP; ++j) g.first[j] = first;
        g.n = nj;
        BDR_HIP(step_launch(st, false, k_dense_dw_small_group, dim3(first), dim3(256), g));
    }
    return BDR_OK;
}

constexpr int RA_SEGS = 12, RA_INST = 4;
struct DenseReduceSeg { const float* part; size_t stride; int chunks; unsigned off4, n4; };   // arena float4s [off4, off4 + n4)
struct ReduceAdamArgs {
    DenseReduceSeg seg[RA_SEGS]; int nseg; size_t inst_part_stride;   // instance z reads seg.part + z * inst_part_stride
    float* p[RA_INST]; float* g[RA_INST]; float* m[RA_INST]; float* v[RA_INST]; float* tgt[RA_INST];
    float* vmax[RA_INST];   // AdamW{amsgrad: true}: max_exp_avg_sq of instance z (nullptr: plain Adam / AdamW)
    AdamScalars s[RA_INST]; unsigned n4; float tau, omt; int track;
    int grads_only;   // 1: sum the partials into the gradient arena and stop (synchronous-DP mode: the all-reduce comes before Adam)
    const unsigned* poison;   // optional: a cross-queue wait of the step timed out (queue_flags.hpp) - the inputs may be incomplete, leave everything alone
    unsigned long long* applied; unsigned long long step;   // optional: *applied = step by a pass that was not skipped (the host rolls its step counter back to it)
};
__global__ __launch_bounds__(256) void k_dense_reduce_adam(ReduceAdamArgs a)
{
    const unsigned i = blockIdx.x * 256 + threadIdx.x;
    const int z = blockIdx.y;
    if (i >= a.n4) return;
    if (a.poison && *a.poison) return;
    if (a.applied && i == 0 && z == 0 && !a.grads_only) *a.applied = a.step;
    int k = 0;
    for (int q = 1; q < a.nseg; ++q) k += i >= a.seg[q].off4 ? 1 : 0;
    const DenseReduceSeg sg = a.seg[k];
    f32x4 gg = f32x4{0.f, 0.f, 0.f, 0.f};
    if (i < sg.off4 + sg.n4) {   // (arena slack past the last layer keeps a zero gradient)
        const float* part = sg.part + (size_t)z * a.inst_part_stride + (size_t)(i - sg.off4) * 4;
        f32x4 s4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) s4[q] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < sg.chunks; c += 4) {   // k_dense_reduce's order: four interleaved partial sums
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (c + q < sg.chunks) s4[q] += *reinterpret_cast<const f32x4*>(part + (size_t)(c + q) * sg.stride);
        }
        gg = ((s4[0] + s4[1]) + s4[2]) + s4[3];
    }
    reinterpret_cast<f32x4*>(a.g[z])[i] = gg;
    if (a.grads_only) return;
    f32x4 pp = reinterpret_cast<f32x4*>(a.p[z])[i], mm = reinterpret_cast<f32x4*>(a.m[z])[i], vv = reinterpret_cast<f32x4*>(a.v[z])[i];
    const AdamScalars s = a.s[z];
    if (a.vmax[z]) {   // (uniform per launch row z)
        f32x4 xx = reinterpret_cast<f32x4*>(a.vmax[z])[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float pe = pp[j], me = mm[j], ve = vv[j], xe = xx[j];
            adam_element_amsgrad(pe, gg[j], me, ve, xe, s);
            pp[j] = pe; mm[j] = me; vv[j] = ve; xx[j] = xe;
        }
        reinterpret_cast<f32x4*>(a.vmax[z])[i] = xx;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float pe = pp[j], me = mm[j], ve = vv[j];
            adam_element(pe, gg[j], me, ve, s);
            pp[j] = pe; mm[j] = me; vv[j] = ve;
        }
    }
    reinterpret_cast<f32x4*>(a.p[z])[i] = pp;
    reinterpret_cast<f32x4*>(a.m[z])[i] = mm;
    reinterpret_cast<f32x4*>(a.v[z])[i] = vv;
    if (a.track) {
        f32x4 d = reinterpret_cast<f32x4*>(a.tgt[z])[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) d[j] = track_element(pp[j], d[j], a.tau, a.omt);
        reinterpret_cast<f32x4*>(a.tgt[z])[i] = d;
    }
}

inline int32_t pack_rows(hipStream_t st, const float* src, int src_ld, int cols, float* dst, int ld, int col0, int B)
{
    const int n = B * cols;
    BDR_HIP(step_launch(st, false, k_pack_rows, dim3((n + 255) / 256), dim3(256), src, src_ld, cols, dst, ld, col0, B));
    return BDR_OK;
}

}  // namespace
