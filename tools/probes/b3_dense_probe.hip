// Would IQN's big dense GEMMs (config C4: [32768][3136] x [3136][512], 77 % of the FP32-MFMA peak) be faster on the bf16 pipes with
// exactly split operands?  Unlike the B = 256 conv layers (bp_probe: operand-delivery bound, no gain) these are matrix-bound.
// k_igemm_b3 (tools/probes/igemm_b3.hpp): A split in the kernel on its way into LDS, B from pre-split k-major planes.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Iborder_amd/csrc -Itools/probes tools/probes/b3_dense_probe.hip -o tools/probes/b3_dense_probe.bin
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "dense.hpp"
#include "igemm_b3.hpp"

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
template <int WM_, int WN_, int TM_, int TN_>
struct DenseFwdB3P : DenseFwd {
    using Args = DenseB3Args;
    static constexpr int WM = WM_, WN = WN_, TM = TM_, TN = TN_;
    __device__ static const uint4* b_chunk(const Args& a, int, int, int pl, int kt, int n, int kq)
    {
        return reinterpret_cast<const uint4*>(a.wpl + (size_t)pl * a.ncols * a.kred + (size_t)n * a.kred + kt * 32 + kq * 8);
    }
};
template <class F>
static double time_us(F f, int reps = 10)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1000.0 / reps;
}
static void compare(const char* name, const float* a, const float* b, size_t n)
{
    std::vector<float> ha(n), hb(n);
    CK(hipMemcpy(ha.data(), a, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), b, n * 4, hipMemcpyDeviceToHost));
    double md = 0, mx = 0;
    for (size_t i = 0; i < n; ++i) { md = std::max(md, (double)std::fabs(ha[i] - hb[i])); mx = std::max(mx, (double)std::fabs(hb[i])); }
    printf("   %-28s max |diff| %.3e, max |ref| %.3e, rel %.2e\n", name, md, mx, md / mx);
}

int main()
{
    const int M = 32768, K = 3136, N = 512;
    float* x = dev_rand((size_t)M * K, 0.f, 1.f, 1);
    float* w = dev_rand((size_t)K * N, -0.02f, 0.02f, 2);
    float* bias = dev_rand(N, -0.1f, 0.1f, 3);
    float *o, *ob;
    CK(hipMalloc(&o, (size_t)M * N * 4)); CK(hipMalloc(&ob, (size_t)M * N * 4));
    uint16_t* wpl; CK(hipMalloc(&wpl, (size_t)K * N * 6));
    hipLaunchKernelGGL(k_split_planes, dim3((unsigned)(((size_t)K * N + 255) / 256)), dim3(256), 0, 0, w, wpl, K, N, 1);
    CK(hipDeviceSynchronize());
    DenseB3Args d{};
    d.x = DenseSrc{x, K}; d.w = w; d.bias = bias; d.out = o; d.ldo = N; d.M = M; d.ncols = N; d.kred = K; d.relu = 1; d.w_ld = N; d.wpl = wpl;
    DenseB3Args db = d; db.out = ob;
    const double gflop = 2.0 * M * K * N / 1e9;
    {
        const double t = time_us([&] { CK((launch_dense<DenseFwd>(0, dim3((M / 64) * (N / 64), 1, 1), (const DenseArgs&)d))); });
        printf("f32 MFMA 64x64            : %8.1f us  %6.1f TFLOP/s\n", t, gflop / t * 1e-3);
    }
    {
        using P = DenseFwdB3P<2, 2, 1, 1>;
        const double t = time_us([&] { CK((launch_igemm_b3<P, 6>(0, dim3((M / 64) * (N / 64), 1, 1), db))); });
        printf("3xbf16 6 terms 64x64      : %8.1f us  %6.1f TFLOP/s (algorithmic)\n", t, gflop / t * 1e-3);
        compare("6 terms vs f32", ob, o, (size_t)M * N);
    }
    {
        using P = DenseFwdB3P<2, 2, 2, 1>;   // 128 x 64
        const double t = time_us([&] { CK((launch_igemm_b3<P, 6>(0, dim3((M / 128) * (N / 64), 1, 1), db))); });
        printf("3xbf16 6 terms 128x64     : %8.1f us  %6.1f TFLOP/s\n", t, gflop / t * 1e-3);
        compare("6 terms 128x64 vs f32", ob, o, (size_t)M * N);
    }
    {
        using P = DenseFwdB3P<2, 2, 2, 2>;   // 128 x 128
        const double t = time_us([&] { CK((launch_igemm_b3<P, 6>(0, dim3((M / 128) * (N / 128), 1, 1), db))); });
        printf("3xbf16 6 terms 128x128    : %8.1f us  %6.1f TFLOP/s\n", t, gflop / t * 1e-3);
        compare("6 terms 128x128 vs f32", ob, o, (size_t)M * N);
    }
    {
        using P = DenseFwdB3P<2, 2, 1, 2>;   // 64 x 128
        const double t = time_us([&] { CK((launch_igemm_b3<P, 6>(0, dim3((M / 64) * (N / 128), 1, 1), db))); });
        printf("3xbf16 6 terms 64x128     : %8.1f us  %6.1f TFLOP/s\n", t, gflop / t * 1e-3);
        compare("6 terms 64x128 vs f32", ob, o, (size_t)M * N);
    }
    {
        using P = DenseFwdB3P<2, 2, 1, 1>;
        const double t = time_us([&] { CK((launch_igemm_b3<P, 9>(0, dim3((M / 64) * (N / 64), 1, 1), db))); });
        printf("3xbf16 9 terms 64x64      : %8.1f us  %6.1f TFLOP/s\n", t, gflop / t * 1e-3);
    }
    return 0;
}
