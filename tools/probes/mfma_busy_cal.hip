// Calibration for the MFMA-busy counter pass (tools/gpu_round4_mfma.sh): kernels that keep every SIMD's matrix pipe issuing back to
// back, so that SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE of these launches is the counter ratio that means "100 % busy" whatever
// units and per-XCD row layout rocprofv3 reports the two counters in.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/mfma_busy_cal.hip -o tools/probes/mfma_busy_cal.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_mfma_cal_f32(float* out, int iters, float a0)
{
    f32x16 acc0 = {0}, acc1 = {0}, acc2 = {0}, acc3 = {0};
    const float a = a0 + threadIdx.x, b = a0 * 0.5f;
    for (int i = 0; i < iters; ++i) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc2, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc3, 0, 0, 0);
    }
    acc0 += acc1; acc2 += acc3; acc0 += acc2;
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc0[r];
    if (s == 12345.678f) out[0] = s;
}
__global__ __launch_bounds__(256) void k_mfma_cal_bf16(float* out, int iters, float a0)
{
    f32x16 acc0 = {0}, acc1 = {0}, acc2 = {0}, acc3 = {0};
    bf16x8 a, b;
    for (int q = 0; q < 8; ++q) { a[q] = (__bf16)(a0 + q + threadIdx.x); b[q] = (__bf16)(a0 * 0.5f); }
    for (int i = 0; i < iters; ++i) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc2, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc3, 0, 0, 0);
    }
    acc0 += acc1; acc2 += acc3; acc0 += acc2;
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc0[r];
    if (s == 12345.678f) out[0] = s;
}

int main()
{
    float* d; CK(hipMalloc(&d, 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = 256 * 2;   // two 4-wave workgroups per CU: two waves per SIMD take turns on its matrix pipe
    for (int rep = 0; rep < 12; ++rep) {
        float ms;
        const int it32 = 4000, it16 = 8000;
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_mfma_cal_f32, dim3(grid), dim3(256), 0, 0, d, it32, 1.0f);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        const double tf32 = (double)grid * 4 * it32 * 4 * (32.0 * 32 * 2 * 2) / (ms * 1e-3) * 1e-12;
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_mfma_cal_bf16, dim3(grid), dim3(256), 0, 0, d, it16, 1.0f);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); float ms2; CK(hipEventElapsedTime(&ms2, e0, e1));
        const double tf16 = (double)grid * 4 * it16 * 4 * (32.0 * 32 * 16 * 2) / (ms2 * 1e-3) * 1e-12;
        if (rep >= 10) printf("f32 32x32x2: %.3f ms, %.1f TFLOP/s   bf16 32x32x16: %.3f ms, %.1f TFLOP/s\n", ms, tf32, ms2, tf16);
    }
    return 0;
}
