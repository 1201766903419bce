// Timing probe for k_conv1_bf16 (not part of the product).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "conv1_bf16_abl.hpp"   // k_conv1_bf16 with the ablation switches (namespace bdr_abl)
using namespace bdr_abl;
__global__ void k_copy(const uint4* s, uint4* d, size_t n) { size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; if (i < n) d[i] = s[i]; }
int main(int argc, char** argv)
{
    const int B = 256, nz = 2, M = B * 400;
    Conv1Args c{};
    c.M = M; c.nz = nz;
    for (int z = 0; z < nz; ++z) {
        uint8_t* x; float *w, *b, *o;
        hipMalloc(&x, (size_t)B * 28224);
        { std::vector<uint8_t> hx((size_t)B * 28224); uint32_t st = 12345u + z; for (auto& v : hx) { st = st * 1664525u + 1013904223u; v = (argc > 1) ? (uint8_t)(st >> 24) : 37; } hipMemcpy(x, hx.data(), hx.size(), hipMemcpyHostToDevice); }
        hipMalloc(&w, 8192 * 4); hipMalloc(&b, 128); hipMalloc(&o, (size_t)M * 32 * 4);
        std::vector<float> hw(8192); for (int i = 0; i < 8192; ++i) hw[i] = (float)((i * 7919) % 1000) * 1e-4f - 0.05f;
        hipMemcpy(w, hw.data(), 8192 * 4, hipMemcpyHostToDevice); hipMemset(b, 0, 128);
        c.x[z] = x; c.w1[z] = w; c.bias[z] = b; c.out[z] = o;
    }
    // cold-input mode: rewrite x from a shadow copy before every launch (like the gather kernel does)
    uint8_t* shadow[2];
    for (int z = 0; z < nz; ++z) { hipMalloc(&shadow[z], (size_t)B * 28224); hipMemcpy(shadow[z], c.x[z], (size_t)B * 28224, hipMemcpyDeviceToDevice); }
    {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        float tot = 0;
        for (int k = 0; k < 20; ++k) {
            for (int z = 0; z < nz; ++z) hipLaunchKernelGGL(k_copy, dim3((B * 28224 / 16 + 255) / 256), dim3(256), 0, 0, (const uint4*)shadow[z], (uint4*)c.x[z], (size_t)B * 28224 / 16);
            hipEventRecord(e0);
            hipLaunchKernelGGL(bdr_abl::k_conv1_bf16, dim3(256 * nz), dim3(512), 0, 0, c);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (k >= 5) tot += ms;
        }
        printf("cold input (rewritten before each launch), g=256: %.2f us per launch\n", tot / 15 * 1000);
    }
    for (int g : {64, 128, 256, 512}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            for (int k = 0; k < 10; ++k) hipLaunchKernelGGL(bdr_abl::k_conv1_bf16, dim3(g * nz), dim3(512), 0, 0, c);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("g=%d per-instance WGs: %.2f us per launch\n", g, ms * 100.0f);
    }
    return 0;
}
