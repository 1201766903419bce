// Probe: conv1 forward, LDS-weights form (conv1_bf16.hpp) against the register-weights form (conv1_bf16_rw.hpp):
// bit comparison of the outputs and launch times at nz = 1 (the product's split schedule) and nz = 2 (the serial profile form).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#define C1_TRACE 1
#include "conv1_bf16_rw.hpp"
#include "../../border_amd/csrc/conv1_bf16_img.hpp"
using namespace bdr;
#define CK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e__), __LINE__); return 1; } } while (0)
// floor probes: the output stores alone (every lane 16 B, a workgroup's stores contiguous), and the input read alone
template <int T> __global__ __launch_bounds__(T) void k_store_only(float* out, size_t n16, float v)
{
    const f32x4 val = {v, v, v, v};
    for (size_t i = (size_t)blockIdx.x * T + threadIdx.x; i < n16; i += (size_t)gridDim.x * T) reinterpret_cast<f32x4*>(out)[i] = val;
}
template <int T> __global__ __launch_bounds__(T) void k_load_only(const uint4* in, size_t n16, unsigned* sink)
{
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * T + threadIdx.x; i < n16; i += (size_t)gridDim.x * T) { const uint4 v = in[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void k_empty() {}
__global__ void k_copy(const uint4* s, uint4* d, size_t n) { size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; if (i < n) d[i] = s[i]; }
int main(int argc, char** argv)
{
    const int B = argc > 1 ? atoi(argv[1]) : 256, NS = 4;
    const int M = B * 400;
    const int form_img = getenv("C1_IMG_WAVES") ? atoi(getenv("C1_IMG_WAVES")) : 8;
    Conv1Args c{}, c2{};
    uint8_t* shadow[2];
    for (int z = 0; z < 2; ++z) {
        uint8_t* x; float *w, *b, *o, *o2;
        CK(hipMalloc(&x, (size_t)B * 28224)); CK(hipMalloc(&shadow[z], (size_t)B * 28224));
        { std::vector<uint8_t> hx((size_t)B * 28224); uint32_t st = 12345u + z; for (auto& v : hx) { st = st * 1664525u + 1013904223u; v = (uint8_t)(st >> 24); }
          CK(hipMemcpy(x, hx.data(), hx.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(shadow[z], hx.data(), hx.size(), hipMemcpyHostToDevice)); }
        CK(hipMalloc(&w, 8192 * 4)); CK(hipMalloc(&b, 128)); CK(hipMalloc(&o, (size_t)M * 32 * 4)); CK(hipMalloc(&o2, (size_t)M * 32 * 4));
        std::vector<float> hw(8192), hb(32);
        uint32_t st = 777u + z;
        for (auto& v : hw) { st = st * 1664525u + 1013904223u; v = ((float)(st >> 8) / 16777216.0f - 0.5f) * 0.125f; }
        for (auto& v : hb) { st = st * 1664525u + 1013904223u; v = ((float)(st >> 8) / 16777216.0f - 0.5f) * 0.1f; }
        CK(hipMemcpy(w, hw.data(), 8192 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(b, hb.data(), 128, hipMemcpyHostToDevice));
        c.x[z] = c2.x[z] = x; c.w1[z] = c2.w1[z] = w; c.bias[z] = c2.bias[z] = b; c.out[z] = o; c2.out[z] = o2;
    }
    c.M = c2.M = M;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int nz = 1; nz <= 2; ++nz) {
        c.nz = c2.nz = nz;
        const int items = (M + 31) / 32;
        const int g_old = std::max(1, std::min(512 / nz, (items + 7) / 8));
        const int g_new = conv1_rw_groups(nz, M);
        for (int z = 0; z < nz; ++z) { CK(hipMemset(c.out[z], 0xff, (size_t)M * 128)); CK(hipMemset(c2.out[z], 0x7f, (size_t)M * 128)); }
        CK(launch_conv1_bf16(NS, dim3(g_old * nz), 0, c));
        const int ipw = B * nz > 256 ? 2 : 1, g_img = std::min(256 / nz, (B + ipw - 1) / ipw);
        if (getenv("C1_FORM_RW")) CK(launch_conv1_bf16_rw(NS, dim3(g_new * nz), 0, c2)); else CK(launch_conv1_bf16_img(NS, ipw, form_img, dim3(g_img * nz), 0, c2));
        CK(hipDeviceSynchronize());
        size_t diff = 0;
        for (int z = 0; z < nz; ++z) {
            std::vector<uint32_t> h1((size_t)M * 32), h2((size_t)M * 32);
            CK(hipMemcpy(h1.data(), c.out[z], h1.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h2.data(), c2.out[z], h2.size() * 4, hipMemcpyDeviceToHost));
            for (size_t k = 0; k < h1.size(); ++k) diff += h1[k] != h2[k];
        }
        printf("nz=%d B=%d: %zu differing output words (grids %d / %d / %d per instance, %d image(s) per pass)\n", nz, B, diff, g_old, g_new, g_img, ipw);
        for (int form = 0; form < 3; ++form) {
            const char* names[3] = {"LDS weights     ", "register weights", "staged images   "};
            float tot = 0, mn = 1e9f;
            for (int k = 0; k < 25; ++k) {   // cold input: rewritten before each launch, as the gather does
                for (int z = 0; z < nz; ++z) hipLaunchKernelGGL(k_copy, dim3((B * 28224 / 16 + 255) / 256), dim3(256), 0, 0, (const uint4*)shadow[z], (uint4*)c.x[z], (size_t)B * 28224 / 16);
                CK(hipEventRecord(e0));
                if (form == 0) CK(launch_conv1_bf16(NS, dim3(g_old * nz), 0, c)); else if (form == 1) CK(launch_conv1_bf16_rw(NS, dim3(g_new * nz), 0, c2)); else CK(launch_conv1_bf16_img(NS, ipw, form_img, dim3(g_img * nz), 0, c2));
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (k >= 5) { tot += ms; mn = std::min(mn, ms); }
            }
            printf("  %s cold input: %.2f us avg, %.2f min\n", names[form], tot / 20 * 1000, mn * 1000);
            CK(hipEventRecord(e0));
            for (int k = 0; k < 20; ++k) { if (form == 0) CK(launch_conv1_bf16(NS, dim3(g_old * nz), 0, c)); else if (form == 1) CK(launch_conv1_bf16_rw(NS, dim3(g_new * nz), 0, c2)); else CK(launch_conv1_bf16_img(NS, ipw, form_img, dim3(g_img * nz), 0, c2)); }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("  %s back to back: %.2f us per launch\n", names[form], ms / 20 * 1000);
        }
    }
    {   // floors: empty kernel, stores alone, loads alone (back to back, 20 launches each)
        unsigned* sink; CK(hipMalloc(&sink, 4));
        auto timeit = [&](const char* name, auto&& launch) {
            for (int k = 0; k < 3; ++k) launch();
            hipEventRecord(e0);
            for (int k = 0; k < 20; ++k) launch();
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("  floor %-58s %.2f us per launch\n", name, ms / 20 * 1000);
        };
        const size_t o16 = (size_t)M * 8, i16 = (size_t)B * 28224 / 16;
        timeit("empty kernel", [&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, 0); });
        timeit("stores of one instance's output, 256 x 256 threads", [&] { hipLaunchKernelGGL(k_store_only<256>, dim3(256), dim3(256), 0, 0, c.out[0], o16, 1.0f); });
        timeit("stores of one instance's output, 512 x 512 threads", [&] { hipLaunchKernelGGL(k_store_only<512>, dim3(512), dim3(512), 0, 0, c.out[0], o16, 1.0f); });
        timeit("stores of one instance's output, 2048 x 256 threads", [&] { hipLaunchKernelGGL(k_store_only<256>, dim3(2048), dim3(256), 0, 0, c.out[0], o16, 1.0f); });
        timeit("loads of one instance's input, 256 x 256 threads", [&] { hipLaunchKernelGGL(k_load_only<256>, dim3(256), dim3(256), 0, 0, (const uint4*)c.x[0], i16, sink); });
        timeit("loads of one instance's input, 1024 x 256 threads", [&] { hipLaunchKernelGGL(k_load_only<256>, dim3(1024), dim3(256), 0, 0, (const uint4*)c.x[0], i16, sink); });
    }
    {   // phase trace of the staged-images form, nz = 1 (stamps of wave 0 / wave 3 of every workgroup, 100 MHz wall clock)
        c.nz = c2.nz = 1;
        const int ipw = B > 256 ? 2 : 1, g_img = std::min(256, (B + ipw - 1) / ipw);
        unsigned long long* tr; CK(hipMalloc(&tr, (size_t)g_img * 2 * 8 * 8)); CK(hipMemset(tr, 0, (size_t)g_img * 2 * 8 * 8));
        CK(hipMemcpyToSymbol(HIP_SYMBOL(g_c1_trace), &tr, sizeof tr));
        for (int rep = 0; rep < 3; ++rep) { CK(launch_conv1_bf16_img(NS, ipw, form_img, dim3(g_img), 0, c2)); CK(hipDeviceSynchronize()); }
        std::vector<unsigned long long> h((size_t)g_img * 16); CK(hipMemcpy(h.data(), tr, h.size() * 8, hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull, t1 = 0; for (int w = 0; w < g_img * 2; ++w) { t0 = std::min(t0, h[w * 8]); for (int k : {0, 1, 2, 3, 4, 6}) t1 = std::max(t1, h[w * 8 + k]); }
        for (int wv = 0; wv < 2; ++wv) {
            double s[7] = {0}; double mx[7] = {0};
            for (int g = 0; g < g_img; ++g) for (int k : {0, 1, 2, 3, 4, 6}) { const double v = (double)(h[(g * 2 + wv) * 8 + k] - t0) * 0.01; s[k] += v; mx[k] = std::max(mx[k], v); }
            printf("trace wave %d (us since the first workgroup started; mean / max over %d workgroups): start %.2f/%.2f  weights split %.2f/%.2f  images committed %.2f/%.2f  barrier %.2f/%.2f  first fragment %.2f/%.2f  end %.2f/%.2f\n",
                   wv ? (int)(8 * 1) - 1 : 0, g_img, s[0] / g_img, mx[0], s[1] / g_img, mx[1], s[2] / g_img, mx[2], s[3] / g_img, mx[3], s[4] / g_img, mx[4], s[6] / g_img, mx[6]);
        }
        { double cyc = 0, us = 0; for (int g = 0; g < g_img; ++g) { cyc += (double)(h[(g * 2) * 8 + 7] - h[(g * 2) * 8 + 5]); us += (double)(h[(g * 2) * 8 + 6] - h[(g * 2) * 8 + 4]) * 0.01; }
          printf("wave 0, first fragment -> end of its units: %.0f shader cycles in %.2f us = %.2f GHz (8 waves per workgroup: 2 units = 96 MFMAs of wave 0, a SIMD carries two waves)\n", cyc / g_img, us / g_img, cyc / us * 1e-3); }
        printf("kernel span first start -> last stamp: %.2f us\n", (double)(t1 - t0) * 0.01);
        unsigned long long* z0 = nullptr; CK(hipMemcpyToSymbol(HIP_SYMBOL(g_c1_trace), &z0, sizeof z0));
    }
    return 0;
}
