// Probe: conv1 weight gradient, the (channel, kh half) tiling of rounds 1-4 (conv1_dw_bf16_v1.hpp) against the (channel pair, column phase)
// tiling (csrc/conv1_dw_bf16.hpp): bit comparison of the partials and launch times.  usage: c1dw_probe.bin [B] [n_stack]
#include <algorithm>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "conv1_dw_bf16_v1.hpp"
using namespace bdr;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
int main(int argc, char** argv)
{
    const int B = argc > 1 ? atoi(argv[1]) : 256, NS = argc > 2 ? atoi(argv[2]) : 4;
    std::vector<uint8_t> hx((size_t)B * NS * 7056);
    std::vector<float> hdy((size_t)B * 400 * 32);
    unsigned s = 1;
    for (auto& v : hx) { s = s * 1664525u + 1013904223u; v = s >> 24; }
    for (auto& v : hdy) { s = s * 1664525u + 1013904223u; v = ((s >> 8) * (1.0f / 16777216.0f) - 0.5f) * 1e-3f; }
    uint8_t* x; float *dy, *p1, *p2;
    const size_t stride = 64 * NS * 32 + 32;
    CK(hipMalloc(&x, hx.size())); CK(hipMalloc(&dy, hdy.size() * 4)); CK(hipMalloc(&p1, B * stride * 4)); CK(hipMalloc(&p2, B * stride * 4));
    CK(hipMemcpy(x, hx.data(), hx.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(dy, hdy.data(), hdy.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(p1, 0xff, B * stride * 4)); CK(hipMemset(p2, 0x7f, B * stride * 4));
    const int grid = std::min(B, 256);
    Conv1DwArgs a1{x, dy, p1, stride, B}, a2{x, dy, p2, stride, B};
    CK(launch_conv1_dw_bf16_v1(NS, dim3(grid), 0, a1)); CK(launch_conv1_dw_bf16(NS, dim3(grid), 0, a2)); CK(hipDeviceSynchronize());
    std::vector<uint32_t> h1((size_t)grid * stride), h2((size_t)grid * stride);
    CK(hipMemcpy(h1.data(), p1, h1.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h2.data(), p2, h2.size() * 4, hipMemcpyDeviceToHost));
    size_t diff = 0; for (size_t k = 0; k < h1.size(); ++k) diff += h1[k] != h2[k];
    double maxabs = 0, maxdiff = 0;
    for (size_t k = 0; k < h1.size(); ++k) {
        float f1, f2; memcpy(&f1, &h1[k], 4); memcpy(&f2, &h2[k], 4);
        maxabs = std::max(maxabs, (double)fabsf(f1)); maxdiff = std::max(maxdiff, (double)fabsf(f1 - f2));
    }
    printf("B=%d n_stack=%d: %zu differing words of %zu; max |difference| %.3g against max |value| %.3g (%.2g relative: the two tilings add the same exact products in a different order)\n",
           B, NS, diff, h1.size(), maxdiff, maxabs, maxdiff / maxabs);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int form = 0; form < 2; ++form) {
        for (int k = 0; k < 3; ++k) { if (form) CK(launch_conv1_dw_bf16(NS, dim3(grid), 0, a2)); else CK(launch_conv1_dw_bf16_v1(NS, dim3(grid), 0, a1)); }
        CK(hipEventRecord(e0, 0));
        for (int k = 0; k < 50; ++k) { if (form) CK(launch_conv1_dw_bf16(NS, dim3(grid), 0, a2)); else CK(launch_conv1_dw_bf16_v1(NS, dim3(grid), 0, a1)); }
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("  %s: %.2f us per launch (back to back)\n", form ? "phase tiles      " : "(c, kh half) tiles", ms * 1000 / 50);
    }
    return 0;
}
