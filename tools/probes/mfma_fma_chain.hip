// Is v_mfma_f32_32x32x2_f32 bitwise a k-ordered fmaf chain (k-slot 0, then k-slot 1, onto the accumulator)?  (not part of the product)
// hipcc --offload-arch=gfx950 -O3 -o tools/probes/mfma_fma_chain.bin tools/probes/mfma_fma_chain.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int K = 256;
__global__ void k_mfma(const float* A, const float* B, float* D)   // A [32][K], B [K][32], D [32][32]
{
    const int lane = threadIdx.x, i = lane & 31, h = lane >> 5;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int k = 0; k < K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[i * K + k + h], B[(k + h) * 32 + i], acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + i] = acc[r];
}
__global__ void k_chain(const float* A, const float* B, float* D, int mode)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x, r = t >> 5, c = t & 31;
    float acc = 0.f;
    if (mode == 0) for (int k = 0; k < K; ++k) acc = fmaf(A[r * K + k], B[k * 32 + c], acc);                       // plain k order
    else for (int k = 0; k < K; k += 2) acc = (A[r * K + k] * B[k * 32 + c] + A[r * K + k + 1] * B[(k + 1) * 32 + c]) + acc;   // pair first (unfused)
    D[t] = acc;
}
int main()
{
    std::mt19937 g(3);
    std::normal_distribution<float> n(0.f, 1.f);
    std::vector<float> A(32 * K), B(K * 32), d0(1024), d1(1024), d2(1024);
    for (auto& v : A) v = n(g);
    for (auto& v : B) v = n(g) * 0.07f;
    float *dA, *dB, *dD;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, 4096);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, 0, dA, dB, dD); hipMemcpy(d0.data(), dD, 4096, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(k_chain, dim3(4), dim3(256), 0, 0, dA, dB, dD, 0); hipMemcpy(d1.data(), dD, 4096, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(k_chain, dim3(4), dim3(256), 0, 0, dA, dB, dD, 1); hipMemcpy(d2.data(), dD, 4096, hipMemcpyDeviceToHost);
    int same1 = 0, same2 = 0;
    for (int i = 0; i < 1024; ++i) { same1 += !memcmp(&d0[i], &d1[i], 4); same2 += !memcmp(&d0[i], &d2[i], 4); }
    printf("mfma == fmaf chain in k order: %d / 1024 bitwise;  mfma == (pair product sum) + acc: %d / 1024\n", same1, same2);
    return 0;
}
