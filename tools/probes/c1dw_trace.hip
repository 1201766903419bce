// Phase timeline of k_conv1_dw_bf16 (B = 256): prologue (dY split) / 25 k-steps / epilogue per workgroup.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DC1DW_TRACE -Iinclude -Iborder_amd/csrc tools/probes/c1dw_trace.hip -o tools/probes/c1dw_trace.bin
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "conv1_dw_bf16.hpp"
using namespace bdr;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

int main()
{
    const int B = 256;
    std::vector<uint8_t> hx((size_t)B * 28224);
    std::vector<float> hdy((size_t)B * 400 * 32);
    unsigned s = 1;
    for (auto& v : hx) { s = s * 1664525u + 1013904223u; v = s >> 24; }
    for (auto& v : hdy) { s = s * 1664525u + 1013904223u; v = ((s >> 8) * (1.0f / 16777216.0f) - 0.5f) * 1e-3f; }
    uint8_t* x; float *dy, *part; unsigned long long* tr;
    const size_t stride = 256 * 32 + 32;
    CK(hipMalloc(&x, hx.size())); CK(hipMalloc(&dy, hdy.size() * 4)); CK(hipMalloc(&part, B * stride * 4)); CK(hipMalloc(&tr, B * 4 * 8));
    CK(hipMemcpy(x, hx.data(), hx.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(dy, hdy.data(), hdy.size() * 4, hipMemcpyHostToDevice));
    Conv1DwArgs a{x, dy, part, stride, B};
    unsigned long long* null = nullptr;
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_c1dw_trace), &null, sizeof(null)));
    for (int i = 0; i < 3; ++i) launch_conv1_dw_bf16(4, dim3(B), 0, a);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < 50; ++i) launch_conv1_dw_bf16(4, dim3(B), 0, a);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("k_conv1_dw_bf16 B=%d: %.2f us per launch (back to back)\n", B, ms * 1000 / 50);
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_c1dw_trace), &tr, sizeof(tr)));
    launch_conv1_dw_bf16(4, dim3(B), 0, a);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(B * 4); CK(hipMemcpy(h.data(), tr, B * 4 * 8, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull;
    for (int i = 0; i < B; ++i) t0 = std::min(t0, h[i * 4]);
    auto stat = [&](int a0, int a1, const char* nm) {
        std::vector<double> v(B);
        for (int i = 0; i < B; ++i) v[i] = (a0 < 0 ? (double)(h[i * 4 + a1] - t0) : (double)(h[i * 4 + a1] - h[i * 4 + a0])) * 0.01;
        std::sort(v.begin(), v.end());
        double m = 0; for (double x : v) m += x; m /= B;
        printf("   %-9s min %6.2f  p50 %6.2f  max %6.2f  mean %6.2f us\n", nm, v.front(), v[B / 2], v.back(), m);
    };
    stat(-1, 0, "start"); stat(0, 1, "prologue"); stat(1, 2, "k-steps"); stat(2, 3, "epilogue"); stat(-1, 3, "end");
    return 0;
}
