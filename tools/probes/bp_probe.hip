// Pre-split bf16 planes implicit GEMM (tools/probes/igemm_bp.hpp) vs the FP32-MFMA kernel: accuracy and time (B=256 Atari shapes).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Iborder_amd/csrc -Itools/probes tools/probes/bp_probe.hip -o tools/probes/bp_probe.bin
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "cnn_layers.hpp"
#include "igemm_bp.hpp"

using namespace bdr;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

static float* dev_rand(size_t n, float lo, float hi, unsigned seed)
{
    std::vector<float> h(n);
    unsigned s = seed * 2654435761u + 12345u;
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = lo + (hi - lo) * ((s >> 8) * (1.0f / 16777216.0f)); }
    float* d; CK(hipMalloc(&d, n * 4)); CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
    return d;
}

template <class G>
struct AFwdBp {
    static constexpr int NKT = G::K / BK;
    struct Row { size_t base; };
    __device__ static Row row(const void*, int m, int M)
    {
        const int mm = m < M ? m : 0;
        const int b = mm / (G::OH * G::OW), rem = mm % (G::OH * G::OW);
        const int oh = rem / G::OW, ow = rem % G::OW;
        return Row{((size_t)(b * G::IH + oh * G::S) * G::IW + ow * G::S) * G::CIN};
    }
    __device__ static size_t aoff(const Row& r, int kt, int q, bool& ok)
    {
        constexpr int TPT = G::CIN / BK;
        const int tap = kt / TPT, c0 = (kt % TPT) * BK;
        const int kh = tap / G::KW, kw = tap % G::KW;
        ok = true;
        return r.base + (size_t)(kh * G::IW + kw) * G::CIN + c0 + q * 8;
    }
};
struct FwdBpArgs : FwdArgs { const uint16_t* xpl[MAXZ]; size_t xps; const uint16_t* wpl[MAXZ]; uint16_t* opl[MAXZ]; size_t ops; };
template <class G, int WM_, int WN_, int TM_, int TN_, bool PLANES_OUT>
struct FwdBp {
    using A = AFwdBp<G>;
    using Args = FwdBpArgs;
    static constexpr int WM = WM_, WN = WN_, TM = TM_, TN = TN_;
    static constexpr int NC = G::COUT;
    __device__ static bool vrow(const Args& a, int mv, int& mr) { return vrow_flat(a.M, mv, mr); }
    __device__ static constexpr int N(const Args&) { return NC; }
    __device__ static int M(const Args& a) { return a.M; }
    __device__ static const uint16_t* a_planes(const Args& a, int z) { return a.xpl[z]; }
    __device__ static size_t a_plane_stride(const Args& a) { return a.xps; }
    __device__ static const uint4* b_chunk(const Args& a, int z, int, int pl, int kt, int n, int kq)
    {
        return reinterpret_cast<const uint4*>(a.wpl[z] + (size_t)pl * G::COUT * G::K + (size_t)n * G::K + kt * 32 + kq * 8);
    }
    __device__ static void kt_range(const Args&, int, int& k0, int& k1) { k0 = 0; k1 = A::NKT; }
    struct Epi { gptr<const float> bias; gptr<float> out; gptr<uint16_t> opl; size_t ops; };
    __device__ static Epi epi(const Args& a, int z, int) { return Epi{pin_sgpr(a.bias[z]), pin_sgpr(a.out[z]), pin_sgpr(a.opl[z]), a.ops}; }
    __device__ static float epi_load(const Epi& e, int, int n) { return e.bias[n]; }
    __device__ static void store(const Epi& e, int m, int n, float v, float bias)
    {
        v += bias;
        v = v > 0.f ? v : 0.f;
        e.out[(size_t)m * NC + n] = v;
        if constexpr (PLANES_OUT) store_split3(e.opl, e.ops, (size_t)m * NC + n, v);
    }
};

template <class F>
static double time_us(F f)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < 50; ++i) f();
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1000.0 / 50;
}
static void compare(const char* name, const float* a, const float* b, size_t n)
{
    std::vector<float> ha(n), hb(n);
    CK(hipMemcpy(ha.data(), a, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), b, n * 4, hipMemcpyDeviceToHost));
    double md = 0, mx = 0;
    for (size_t i = 0; i < n; ++i) { md = std::max(md, (double)std::fabs(ha[i] - hb[i])); mx = std::max(mx, (double)std::fabs(hb[i])); }
    printf("   %-28s max |diff| %.3e, max |ref| %.3e, rel %.2e\n", name, md, mx, md / mx);
}

template <class G, int KDIM>
static void run_fwd(const char* name, int B, int rows_per_img, unsigned seed)
{
    const int NZ = 2;
    const size_t nin = (size_t)B * G::IH * G::IW * G::CIN, nout = (size_t)B * rows_per_img * G::COUT;
    float* x = dev_rand(nin, 0.f, 1.f, seed);
    float* w = dev_rand((size_t)G::K * G::COUT, -0.05f, 0.05f, seed + 1);
    float* bias = dev_rand(G::COUT, -0.1f, 0.1f, seed + 2);
    float *o[2], *ob[2];
    uint16_t *xpl, *wpl, *opl[2];
    for (int z = 0; z < 2; ++z) { CK(hipMalloc(&o[z], nout * 4)); CK(hipMalloc(&ob[z], nout * 4)); CK(hipMalloc(&opl[z], nout * 6)); }
    CK(hipMalloc(&xpl, nin * 6)); CK(hipMalloc(&wpl, (size_t)G::K * G::COUT * 6));
    hipLaunchKernelGGL(k_split_planes, dim3((unsigned)((nin + 255) / 256)), dim3(256), 0, 0, x, xpl, (int)(nin / G::CIN), G::CIN, 0);
    hipLaunchKernelGGL(k_split_planes, dim3((G::K * G::COUT + 255) / 256), dim3(256), 0, 0, w, wpl, G::K, G::COUT, 1);
    CK(hipDeviceSynchronize());
    FwdBpArgs f{}; f.M = B * rows_per_img; f.xps = nin; f.ops = nout;
    for (int z = 0; z < NZ; ++z) { f.x[z] = x; f.w[z] = w; f.bias[z] = bias; f.out[z] = o[z]; f.xpl[z] = xpl; f.wpl[z] = wpl; f.opl[z] = opl[z]; }
    FwdBpArgs fb = f; for (int z = 0; z < NZ; ++z) fb.out[z] = ob[z];
    using PF = FwdP<G, AFwd<G>, 2, 2, false>;
    const dim3 g((f.M + 63) / 64, 1, NZ);
    printf("%s f32 MFMA 64x64            : %7.2f us\n", name, time_us([&] { CK((launch_igemm<PF, 1>(0, g, (const FwdArgs&)f))); }));
    {
        using P = FwdBp<G, 2, 2, 1, 1, false>;
        printf("%s planes 6t 64x64 f32 out    : %7.2f us\n", name, time_us([&] { CK((launch_igemm_bp<P, 6>(0, g, fb))); }));
        compare("6 terms vs f32", ob[1], o[1], nout);
    }
    {
        using P = FwdBp<G, 2, 2, 1, 1, true>;
        printf("%s planes 6t 64x64 f32+planes : %7.2f us\n", name, time_us([&] { CK((launch_igemm_bp<P, 6>(0, g, fb))); }));
        compare("6 terms vs f32", ob[1], o[1], nout);
    }
    {
        using P = FwdBp<G, 2, 2, 1, 1, false>;
        printf("%s planes 9t 64x64 f32 out    : %7.2f us\n", name, time_us([&] { CK((launch_igemm_bp<P, 9>(0, g, fb))); }));
        compare("9 terms vs f32", ob[1], o[1], nout);
    }
    {
        using P = FwdBp<G, 4, 1, 1, 2, true>;   // 4 waves x (32 x 64): 128 x 64 tile
        const dim3 gw((f.M + 127) / 128, 1, NZ);
        printf("%s planes 6t 128x64 f32+planes: %7.2f us\n", name, time_us([&] { CK((launch_igemm_bp<P, 6>(0, gw, fb))); }));
        compare("6 terms 128x64 vs f32", ob[1], o[1], nout);
    }
    {
        using P = FwdBp<G, 2, 1, 1, 2, true>;   // 2 waves x (32 x 64): 64 x 64 tile, 128 threads
        printf("%s planes 6t 2w 64x64 f32+pl  : %7.2f us\n", name, time_us([&] { CK((launch_igemm_bp<P, 6>(0, g, fb))); }));
        compare("6 terms 2 waves vs f32", ob[1], o[1], nout);
    }
    {
        using P = FwdBp<G, 1, 2, 1, 1, true>;   // 2 waves: 32 x 64 tile
        const dim3 gs((f.M + 31) / 32, 1, NZ);
        printf("%s planes 6t 32x64 f32+planes : %7.2f us\n", name, time_us([&] { CK((launch_igemm_bp<P, 6>(0, gs, fb))); }));
        compare("6 terms 32x64 vs f32", ob[1], o[1], nout);
    }
}

int main()
{
    run_fwd<GeomC2, 512>("fwd_c2", 256, 81, 1);
    run_fwd<GeomC3, 576>("fwd_c3", 256, 49, 11);
    return 0;
}
