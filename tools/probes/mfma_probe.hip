// Ablation probe for the FP32-MFMA inner loop (not part of the product).
// hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_probe tools/probes/mfma_probe.hip && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int NACC>
__global__ __launch_bounds__(256) void probe(const float* __restrict__ g, float* out, int iters, unsigned mask)
{
    __shared__ __attribute__((aligned(16))) float lds[2 * (32 * 68 + 32 * 64)];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2 * (32 * 68 + 32 * 64); i += 256) lds[i] = (float)(i % 7) * 0.01f;
    __syncthreads();
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    const int i = lane & 31, h = lane >> 5;
    float av = 1.0f + lane * 0.001f, bv = 0.5f;
    f32x4 pre = {0, 0, 0, 0};
    int cur = 0;
    for (int it = 0; it < iters; ++it) {
        const float* As = lds + cur * (32 * 68 + 32 * 64);
        const float* Ys = As + 32 * 68;
        if (MODE >= 3) {   // global prefetch + commit to the other stage
            f32x4 v = *reinterpret_cast<const f32x4*>(g + ((size_t)(((blockIdx.x * 64 + it) & mask)) * 256 + tid) * 4);
            float* W = lds + (cur ^ 1) * (32 * 68 + 32 * 64);
            *reinterpret_cast<f32x4*>(&W[(tid >> 3) * 68 + (tid & 7) * 4]) = pre;
            *reinterpret_cast<f32x4*>(&W[32 * 68 + (tid >> 4) * 64 + (tid & 15) * 4]) = pre;
            pre = v;
        }
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            float a = av, b = bv;
            if (MODE >= 1) {
                const int red = 2 * t + h;
                a = As[red * 68 + (wave >> 1) * 32 + i];
                b = Ys[red * 64 + (wave & 1) * 32 + i];
            }
            acc[t % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t % NACC], 0, 0, 0);
        }
        if (MODE >= 2) __syncthreads();
        if (MODE >= 3) cur ^= 1;
    }
    float s = 0;
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 256 + tid] = s;
}

template <int MODE, int NACC>
void run(const char* name, int wgs, int iters, const float* g, float* out, unsigned mask = 0xffffffffu)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<MODE, NACC>), dim3(wgs), dim3(256), 0, 0, g, out, iters, mask);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<MODE, NACC>), dim3(wgs), dim3(256), 0, 0, g, out, iters, mask);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)wgs * 4 * iters * 16 * 2.0 * 32 * 32 * 2;
    printf("%-38s wgs=%5d iters=%4d  %8.2f us  %7.1f TFLOP/s  (%.0f cyc/MFMA/SIMD @2.4GHz, waves/SIMD=%.2f)\n", name, wgs, iters, ms * 1e3,
           flops / (ms * 1e-3) / 1e12, ms * 1e-3 * 2.4e9 / ((double)wgs * 4 / 1024.0 * iters * 16), wgs * 4 / 1024.0);
}

int main()
{
    float *g, *out;
    hipMalloc(&g, (size_t)4096 * 64 * 256 * 16 + 1024);
    hipMemset(g, 0, (size_t)4096 * 64 * 256 * 16);
    hipMalloc(&out, 4096 * 256 * 4);
    for (int wgs : {256, 512, 768, 1024}) {
        run<0, 1>("regs only, 1 acc (dependent chain)", wgs, 64, g, out);
        run<0, 2>("regs only, 2 acc", wgs, 64, g, out);
        run<1, 1>("+ds_read_b32 x2 per MFMA, 1 acc", wgs, 64, g, out);
        run<1, 2>("+ds_read x2, 2 acc", wgs, 64, g, out);
        run<2, 1>("+barrier per 16 MFMA, 1 acc", wgs, 64, g, out);
        run<3, 1>("+global load + ds_write (dbuf), HBM", wgs, 64, g, out);
        run<3, 1>("+global load + ds_write (dbuf), 16MB set", wgs, 64, g, out, 4095);
        run<3, 1>("+global load + ds_write (dbuf), 1MB set", wgs, 64, g, out, 255);
        printf("\n");
    }
    return 0;
}
